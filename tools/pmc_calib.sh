#!/bin/bash
# GPU box: what FETCH_SIZE / WRITE_SIZE report for small random accesses (tools/microbench/hbm_random_access.hip), one rocprofv3 --pmc pass per counter (no trace domains).
# Writes gpurun_out/hbm_counter_calibration.json: per kernel the bytes it asked for, the counter bytes (KB x 1024) and their ratio.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_calib; mkdir -p $OUT; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/hbm_random_access $ROOT/tools/microbench/hbm_random_access.hip || exit 1
cd /tmp
$OUT/hbm_random_access > $OUT/plain.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o pmc -- $OUT/hbm_random_access > $OUT/$c.log 2>&1; done
python - $OUT <<'PY'
import csv, glob, json, re, sys
out = sys.argv[1]; req = {}
for l in open(out + "/plain.log"):
    m = re.match(r"(calib_\w+)\s+requested_bytes (\d+)\s+ms ([\d.]+)", l)
    if m: req[m.group(1)] = {"requested_bytes": float(m.group(2)), "ms": float(m.group(3))}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True))
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        m = re.search(r"calib_\w+", r["Kernel_Name"])
        if m and m.group(0) in req: req[m.group(0)][c + "_bytes"] = req[m.group(0)].get(c + "_bytes", 0.0) + float(r["Counter_Value"]) * 1024.0
for k, v in req.items():
    v["counter_bytes_per_requested_byte"] = (v.get("FETCH_SIZE_bytes", 0.0) + v.get("WRITE_SIZE_bytes", 0.0)) / v["requested_bytes"]
    v["counter_GBps"] = (v.get("FETCH_SIZE_bytes", 0.0) + v.get("WRITE_SIZE_bytes", 0.0)) / v["ms"] / 1e6
json.dump({"source": "tools/pmc_calib.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, Counter_Value x 1024) over tools/microbench/hbm_random_access.hip (8 GiB footprint, 2^28 accesses per kernel)", "kernels": req},
          open(out + "/../hbm_counter_calibration.json", "w"), indent=1)
print(json.dumps(req, indent=1))
PY
