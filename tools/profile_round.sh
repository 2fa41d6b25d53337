#!/bin/bash
# Run ON THE GPU BOX (through gpurun): collects the rocprofv3 evidence for bench.py's numbers into gpurun_out/prof_<tag>/.
#   tools/profile_round.sh <tag>
# 1. kernel trace + stats (csv) of `bench.py --steps 3 --warmup 1`; 2./3. PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace
# domains) of one step.  tools/profile_summary.py turns the raw output into the files committed under profiles/.
set -u
TAG=${1:-rXX}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-pipeline --no-extras"      # one stream: every kernel has the GPU to itself (what bench.py's roofline objects are computed from, too)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B --steps 3 --warmup 1 > $OUT/bench_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $B --no-two-pass --steps 1 --warmup 0 > $OUT/bench_pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_MFMA -o pmc -- $B --no-two-pass --steps 1 --warmup 0 > $OUT/bench_pmc_MFMA.log 2>&1
cd $ROOT
timeout 900 python bench.py --steps 20 --warmup 3 --measure-traffic > $OUT/bench_line.json 2> $OUT/bench_line.err
# the decoder's frames by path (shipped library) and, when tools/build_variant.sh fp -DK3_FAST_PROF was run, the phases of the LDS-resident path
python tools/prof_literal.py 512 > $OUT/literal_frames_by_path.txt 2>&1
[ -f build/libk3hip_fp.so ] && K3HIP_LIB=build/libk3hip_fp.so python tools/prof_fast.py 512 >> $OUT/literal_frames_by_path.txt 2>&1
# chain training over the adapter: kernel stats of 24 iterations on the benchmark model + the per-iteration times without the profiler
export K3_TRAIN_BIG=1 RUN_REF=0
python tools/debug_chain_train.py /tmp/ctb_prof 24 2>&1 | tail -1 > $OUT/chain_train_iterations.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- python $ROOT/tools/debug_chain_train.py /tmp/ctb_prof 24 > /dev/null 2>&1)
python tools/profile_summary.py $TAG $OUT
ls -la $OUT
