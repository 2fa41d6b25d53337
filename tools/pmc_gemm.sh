#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the front end (fbank + TDNN-F only), one rocprofv3 --pmc pass per counter group.
#   tools/pmc_gemm.sh [kernel-name filter, default gemm; `feat` = k3_feat_kernel]
set -u
export K3_PMC_FILTER=${1:-gemm}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_gemm; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-decode --steps 1 --warmup 0"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o pmc -- $B > $OUT/g$i.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, glob, os, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_gemm/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]); k = re.match(r"([A-Za-z0-9_:]+(?:<[^>]*>)?)", k).group(1)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
with open("gpurun_out/pmc_gemm/summary.txt", "w") as o:
    for k in acc:
        if os.environ.get("K3_PMC_FILTER", "gemm") not in k: continue
        o.write(k + "\n")
        for c, v in sorted(acc[k].items()): o.write("  %-32s %.4g\n" % (c, v))
        w = acc[k].get("SQ_WAVE_CYCLES", 0)
        if w: o.write("  ratios to SQ_WAVE_CYCLES: " + ", ".join("%s %.3f" % (c, acc[k][c] / w) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if c in acc[k]) + "\n")
print(open("gpurun_out/pmc_gemm/summary.txt").read())
PY
