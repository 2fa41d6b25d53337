#!/usr/bin/env python3
"""Reduce the raw rocprofv3 output of tools/profile_round.sh to the small summaries committed under profiles/:
  <tag>_bench_full_pipeline_kernel_stats.csv   the --stats kernel table of the traced bench run
  <tag>_bench_full_pipeline_pmc_hbm.csv        FETCH_SIZE / WRITE_SIZE summed per kernel (KB, as rocprof reports them)
  <tag>_bench_line.json                        the bench.py line of the same build
  hbm_traffic.json                             per-launch HBM traffic of the dominant kernel (read by bench.py's roofline.traffic)
usage: profile_summary.py <tag> <dir with trace/ pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ bench_line.json>"""
import csv, glob, json, os, re, sys
tag, src = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); dst = os.path.join(ROOT, "gpurun_out", "profiles_" + tag); os.makedirs(dst, exist_ok=True)
def find(sub, pat):
    f = sorted(glob.glob(os.path.join(src, sub, "**", pat), recursive=True)); return f[0] if f else None
st = find("trace", "*kernel_stats.csv")
if st:
    txt = open(st).read()
    # rocprofv3's --stats table counts every launch of a kernel; the token-passing kernel is also launched once or twice per decoder object at creation (one lane, zero / one
    # frame: the templates): a line per kernel with the launches of at least 1 % of its longest, from the kernel trace of the same run, is appended
    kt = find("trace", "*kernel_trace.csv"); extra = []
    if kt:
        dur = {}
        for r in csv.DictReader(open(kt)):
            dur.setdefault(r["Kernel_Name"], []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            keep = [x for x in v if x >= 0.01 * max(v)]
            if len(keep) != len(v) and "k3_decode_forward_literal_kernel" in k: extra.append('# workload launches only (>= 1 %% of the longest): "%s",%d launches, average %.1f ns, min %.0f, max %.0f (the other %d: template builds at decoder creation)' % (k, len(keep), sum(keep) / len(keep), min(keep), max(keep), len(v) - len(keep)))
    open(os.path.join(dst, f"{tag}_bench_full_pipeline_kernel_stats.csv"), "w").write(txt + ("\n".join(extra) + "\n" if extra else ""))
rows = []
def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^>]*>)?)", name); return m.group(1) if m else name
traffic = {}; traffic2 = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = find("pmc_" + c, "*counter_collection.csv")
    if not f: continue
    per = {}      # kernel -> dispatch -> counter value
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != c: continue
        k = short(r["Kernel_Name"]); dd = per.setdefault(k, {}); dd[r["Dispatch_Id"]] = dd.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    # a decoder object decodes "no frames" / one frame on one lane when it is created (the InitDecoding / first-frame templates): those launches of the token-passing kernel
    # move almost nothing and are not launches of the workload -- a dispatch below 1 % of the kernel's largest is left out of the per-launch figures
    acc = {}
    for k, dd in per.items():
        big = max(dd.values()) if dd else 0.0
        keep = {i: v for i, v in dd.items() if v >= 0.01 * big}
        acc[k] = [set(keep), sum(keep.values())]
    for k, (d, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        rows.append((k, c, len(d), v, v / max(1, len(d))))
        if k.startswith("k3_decode_forward_literal_kernel"): traffic[c] = v / max(1, len(d)) * 1024.0
        if k.startswith("k3_decode_forward_kernel"): traffic2[c] = v / max(1, len(d)) * 1024.0
with open(os.path.join(dst, f"{tag}_bench_full_pipeline_pmc_hbm.csv"), "w") as f:
    f.write(f"# rocprofv3 --pmc <counter> (separate passes, no trace domains) over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline` (512 x 10 s utts), {tag}\n"
            "# Counter_Value summed over the dispatches of each kernel; FETCH_SIZE / WRITE_SIZE are in KB (rocprof definition).\n"
            "# gfx950 caveat (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced loads (x2 correction); narrow random\n"
            "# calibration (profiles/hbm_counter_calibration_r04.json, tools/pmc_calib.sh): a random 4 - 16 B read counts one 64 B line, a random 4 - 16 B write or atomic 32 B, streaming 16 B reads count 0.5x.\n"
            "kernel,counter,dispatches,sum_KB,per_dispatch_KB\n")
    for r in rows: f.write("%s,%s,%d,%.1f,%.1f\n" % r)
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl):
    lines = [l for l in open(bl).read().splitlines() if l.startswith("{")]
    if lines: open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(lines[-1] + "\n")
if len(traffic) == 2:
    json.dump({"kernel": "k3_decode_forward_literal_kernel", "two_pass_kernel_traffic_bytes_per_launch": (traffic2.get("FETCH_SIZE", 0) + traffic2.get("WRITE_SIZE", 0)) or None, "source": f"profiles/{tag}_bench_full_pipeline_pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 512 x 10 s utts)",
               "fetch_bytes_per_launch": traffic["FETCH_SIZE"], "write_bytes_per_launch": traffic["WRITE_SIZE"], "traffic_bytes_per_launch": traffic["FETCH_SIZE"] + traffic["WRITE_SIZE"],
               "note": "FETCH_SIZE/WRITE_SIZE as reported (KB x 1024).  Calibrated with tools/microbench/hbm_random_access.hip (profiles/hbm_counter_calibration_r04.json): a random 4 - 16 B read counts one 64 B line (16x / 4x the requested bytes), a random 4 - 16 B write or atomic counts 32 B (8x / 2x), streaming 16 B reads count 0.5x, streaming writes 1.0x -- so for the decoder's access mix (narrow random accesses) the counters are memory-side bytes at line granularity, not requested bytes: traffic / algorithmic bytes = 5.5 is mostly the 64 B / 32 B granule around 4 - 16 B items"},
              open(os.path.join(dst, f"hbm_traffic_{tag[:3]}.json"), "w"), indent=1)
fm = find("pmc_MFMA", "*counter_collection.csv")
if fm:      # MfmaUtil (gfx94x formula) per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); rocprofv3 reports GRBM_GUI_ACTIVE summed over the XCDs
    acc = {}
    for r in csv.DictReader(open(fm)):
        a = acc.setdefault(short(r["Kernel_Name"]), {"d": set(), "SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0}); a["d"].add(r["Dispatch_Id"])
        if r["Counter_Name"] in a: a[r["Counter_Name"]] += float(r["Counter_Value"])
    with open(os.path.join(dst, f"{tag}_bench_full_pipeline_pmc_mfma.csv"), "w") as f:
        f.write(f"# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass, no trace domains) over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --no-extras --no-two-pass`, {tag}\n"
                "# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs) -- the gfx94x MfmaUtil formula (GRBM_GUI_ACTIVE comes summed over the 8 XCDs)\nkernel,dispatches,SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch,GRBM_GUI_ACTIVE_per_dispatch,mfma_util\n")
        for k, a in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]):
            if a["SQ_VALU_MFMA_BUSY_CYCLES"] > 0 and a["GRBM_GUI_ACTIVE"] > 0: f.write("%s,%d,%.0f,%.0f,%.3f\n" % (k, len(a["d"]), a["SQ_VALU_MFMA_BUSY_CYCLES"] / len(a["d"]), a["GRBM_GUI_ACTIVE"] / len(a["d"]), a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)))
ts = find("train", "*kernel_stats.csv")
if ts: open(os.path.join(dst, f"{tag}_chain_train_kernel_stats.csv"), "w").write("# rocprofv3 --kernel-trace --stats of kaldi_amd/adapter/_build/nnet3-chain-train, 24 iterations, benchmark model, 64 sequences x 50 frames (tools/debug_chain_train.py, K3_TRAIN_BIG=1); iteration times without the profiler: " + (open(os.path.join(src, "chain_train_iterations.txt")).read().strip() if os.path.exists(os.path.join(src, "chain_train_iterations.txt")) else "") + "\n" + open(ts).read())
fp = os.path.join(src, "literal_frames_by_path.txt")
if os.path.exists(fp): open(os.path.join(dst, f"{tag}_literal_frames_by_path.txt"), "w").write("".join(l for l in open(fp) if "amdgpu.ids" not in l))
print("summaries in", dst, os.listdir(dst))
