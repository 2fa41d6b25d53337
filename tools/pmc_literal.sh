#!/bin/bash
# GPU box: HBM-side traffic (rocprofv3 FETCH_SIZE / WRITE_SIZE, one pass per counter) of the decoder kernels on the bench-configuration decode of
# tools/prof_literal.py.  Prints KB per dispatch.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_literal; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o pmc -- python $ROOT/tools/prof_literal.py 512 > $OUT/$c.log 2>&1
  python - "$OUT/$c" $c <<'PY'
import csv, glob, re, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[0]; acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != sys.argv[2] or "decode" not in r["Kernel_Name"]: continue
    k = re.search(r"k3_\w+", r["Kernel_Name"]).group(0); acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in acc.items(): print(sys.argv[2], k, "dispatches", n, "GB per dispatch %.1f" % (v / n * 1024 / 1e9))
PY
done
