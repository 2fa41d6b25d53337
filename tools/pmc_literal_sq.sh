#!/bin/bash
# GPU box: SQ counters of the literal_order token-passing kernel on the bench-configuration decode of tools/prof_literal.py (one rocprofv3 --pmc pass per group, no trace domains).
# Prints per counter the sum over the kernel's dispatches and the ratios DESIGN.md 4 quotes (issue utilisation by instruction class, wait share).
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_literal_sq; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_FLAT SQ_WAVES"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o pmc -- python $ROOT/tools/prof_literal.py 512 > $OUT/g$i.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); nd = set()
for f in glob.glob("gpurun_out/pmc_literal_sq/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "forward_literal" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); nd.add((f, r["Dispatch_Id"]))
with open("gpurun_out/pmc_literal_sq/summary.txt", "w") as o:
    for c, v in sorted(acc.items()): o.write("%-28s %.5g\n" % (c, v))
    w = acc.get("SQ_WAVE_CYCLES", 0) or 1
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS"):
        if c in acc: o.write("%-28s / SQ_WAVE_CYCLES = %.4f\n" % (c, acc[c] / w))
    b = acc.get("SQ_BUSY_CYCLES", 0)
    if b: o.write("SQ_WAVE_CYCLES / SQ_BUSY_CYCLES = %.3f (waves resident per busy SQ cycle)\n" % (w / b))
print(open("gpurun_out/pmc_literal_sq/summary.txt").read())
PY
