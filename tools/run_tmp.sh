python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py > gpurun_out/bench37.json 2> gpurun_out/bench37.err; tail -c 600 gpurun_out/bench37.err
