python -m pytest tests/test_parallel_gpu.py tests/test_nnet_gpu.py -x -q -m gpu 2>&1 | tail -15 | cut -c1-800
