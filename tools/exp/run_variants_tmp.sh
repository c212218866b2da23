echo "== round 4 tree"
(cd _prev && timeout 300 python tools/exp/sliding_time.py --nq 21 8 5 --reps 40 2>&1 | grep scan_ms | cut -c1-90)
echo "== now"
timeout 300 python tools/exp/sliding_time.py --nq 21 8 5 --reps 40 2>&1 | grep scan_ms | cut -c1-90
timeout 600 python tools/exp/sliding_batch_time.py --nq 5 8 2>&1 | grep n_query
timeout 900 python -m pytest tests/test_gpu_ragged.py -x -q -k "batches or random_shapes or adversarial" --tb=short 2>&1 | tail -3
