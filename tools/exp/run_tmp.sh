mkdir -p gpurun_out/r5fuzz5
timeout 2400 python tools/fuzz_parity.py 150000 4102 > gpurun_out/r5fuzz5/parity.log 2>&1
timeout 1200 python tools/fuzz_ragged.py 20000 4103 > gpurun_out/r5fuzz5/ragged.log 2>&1
for f in parity ragged; do echo "$f: $(grep -E 'trials|MISMATCH' gpurun_out/r5fuzz5/$f.log | tail -2 | tr '\n' ' ')"; done
