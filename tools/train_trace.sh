#!/bin/bash
# tools/train_trace.sh <tag> [iters]: per-kernel trace and timed per-product GEMM trace (K3_GEMM_TRACE=2) of kaldi_amd/adapter nnet3-chain-train on the benchmark model
# (K3_TRAIN_BIG=1 tools/debug_chain_train.py); outputs gpurun_out/<tag>/{gemm_trace.log,kernel_trace_compact.txt}.  Run on the GPU box; analyse with tools/train_trace_report.py <tag>.
tag=${1:-tr}; it=${2:-14}; R=$PWD; mkdir -p gpurun_out/$tag; export K3_TRAIN_BIG=1 RUN_REF=0
python tools/debug_chain_train.py /tmp/ctb 1 >/dev/null 2>&1
K3_GEMM_TRACE=2 python tools/debug_chain_train.py /tmp/ctb $it > /dev/null 2>&1; cp /tmp/ctb/gpu.log gpurun_out/$tag/gemm_trace.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/tools/debug_chain_train.py /tmp/ctb $it > /dev/null 2>&1
f=$(ls /tmp/kt/*/*kernel_trace.csv | head -1)
python - "$f" "$R/gpurun_out/$tag/kernel_trace_compact.txt" <<PY
import sys,csv
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
out=open(sys.argv[2],"w"); t0=int(rows[0]["Start_Timestamp"])
for r in rows: out.write("%d %d %s\n"%(int(r["Start_Timestamp"])-t0,int(r["End_Timestamp"])-int(r["Start_Timestamp"]),r["Kernel_Name"][:70].replace(" ","_")))
PY
