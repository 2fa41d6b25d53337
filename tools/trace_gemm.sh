#!/bin/bash
# Run ON THE GPU BOX: per-launch durations of one nnet3 forward (kernel trace), printed with the GEMM shapes
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_gemm; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $ROOT/bench.py --no-cpu-baseline --no-decode --steps 2 --warmup 1 > $OUT/log.txt 2>&1
cd $ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_gemm/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
last = rows[-37:]; tot = 0
for i, r in enumerate(last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    print(i, "BN128" if "128, 64" in r["Kernel_Name"] else "BN96 ", "aligned" if "true" in r["Kernel_Name"] or "1>" in r["Kernel_Name"] else "general", r["Grid_Size_X"], "%.0f us" % d)
print("total %.0f us" % tot)
PY
