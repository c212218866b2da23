timeout 900 python -m pytest tests/test_gpu_ragged.py -x -q -k "not nccl_at_one_rank" --tb=short 2>&1 | tail -15
timeout 300 python tools/exp/sliding_time.py --nq 21 48 30 8 --reps 30 2>&1 | grep scan_ms | cut -c1-130
