for v in C4P1 C2P1 C4P2 C1P1; do
echo "== $v"
LBAD_LIB=lbaudiodetective_amd/lib/exp/lib_$v.so timeout 300 python tools/exp/sliding_time.py --nq 21 48 --reps 30 2>&1 | grep scan_ms | cut -c1-100
done
echo "== base"
timeout 300 python tools/exp/sliding_time.py --nq 21 48 --reps 30 2>&1 | grep scan_ms | cut -c1-100
