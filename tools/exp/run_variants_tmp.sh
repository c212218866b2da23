timeout 2400 python -m pytest tests -x -q -m gpu --tb=short 2>&1 | tail -15
