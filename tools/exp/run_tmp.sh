mkdir -p gpurun_out/r5fuzz4
timeout 1200 python tools/fuzz_parity.py 40000 3102 > gpurun_out/r5fuzz4/parity.log 2>&1
timeout 900 python tools/fuzz_ragged.py 6000 3103 > gpurun_out/r5fuzz4/ragged.log 2>&1
timeout 300 python tools/fuzz_compare.py 3000 3104 > gpurun_out/r5fuzz4/compare.log 2>&1
timeout 300 python tools/fuzz_stage2.py 20000 3101 > gpurun_out/r5fuzz4/stage2.log 2>&1
timeout 300 python tools/fuzz_frame.py 20000 3105 > gpurun_out/r5fuzz4/frame.log 2>&1
timeout 300 python tools/fuzz_stream.py 3000 3106 > gpurun_out/r5fuzz4/stream.log 2>&1
timeout 900 python tools/fuzz_files.py 3000 3107 > gpurun_out/r5fuzz4/files.log 2>&1
for f in parity ragged compare stage2 frame stream files; do echo "$f: $(grep -E 'trials|frames per shape|MISMATCH' gpurun_out/r5fuzz4/$f.log | tail -2 | tr '\n' ' ')"; done
