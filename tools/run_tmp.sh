python -m pytest tests/test_adapter_gpu.py -x -q -m gpu 2>&1 | tail -5 | cut -c1-600
