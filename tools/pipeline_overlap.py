#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace csv of the default (pipelined) bench.py run to what it shows about the overlap of batches: for every launch of the token-passing kernel,
how much of the next batch's front end (feature kernel + TDNN-F GEMMs, launched on the second stream) ran before that launch ended.
   rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --no-cpu-baseline --no-two-pass --steps 4 --warmup 1;  python tools/pipeline_overlap.py out/t_kernel_trace.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
def nm(r): n = r["Kernel_Name"]; i = n.find("k3_"); return n[i:n.find("(", i)] if i >= 0 else n[:40]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows), key=lambda e: e[0])
dec = [e for e in ev if e[2].startswith("k3_decode_forward")]
fe = [e for e in ev if e[2].startswith("k3_tdnn_gemm_kernel") or e[2].startswith("k3_feat_kernel")]
t00 = ev[0][0]
print("# token-passing launches of the traced run (the first timed-or-warm-up step has nothing queued behind it; the serial pass at the end of bench.py runs one stream).")
print("# `queued` = front-end kernels (feature kernel + TDNN-F GEMMs of the NEXT batch, second stream) whose dispatch began while this launch ran: their workgroups take the CUs its lanes free;")
print("# `front end done` = end of the last of them relative to this launch's end; `period` = start of the next token-passing launch - start of this one.")
print("# launch  kernel                              start_ms    dur_ms | queued  front end done   period")
for i, d in enumerate(dec):
    nxt = dec[i + 1][0] if i + 1 < len(dec) else None
    inside = [f for f in fe if d[0] < f[0] < d[1]]
    done = (max(f[1] for f in inside) - d[1]) / 1e6 if inside else float("nan")
    print("%2d %-34s %10.2f %9.2f | %5d   %+10.2f ms   %s" % (i, d[2], (d[0] - t00) / 1e6, (d[1] - d[0]) / 1e6, len(inside), done, "%8.2f ms" % ((nxt - d[0]) / 1e6) if nxt else "       -"))
# ---- per launch: where the next batch's front end sits relative to it, and when the next token-passing launch starts relative to the end of that front end
prn = [e for e in ev if e[2].startswith("k3_decode_prune")]
print("# launch   dur_ms | next batch's front end: first kernel starts at, last ends at (ms after this launch's start), sum of its kernels' durations | next launch starts (ms after this one's start; after that front end's end) | pruning kernel of this launch: start, dur")
for i, d in enumerate(dec):
    nxt = dec[i + 1][0] if i + 1 < len(dec) else None
    inside = [f for f in fe if d[0] < f[0] < (nxt if nxt else d[1])]
    if not inside or nxt is None: continue
    f0 = min(f[0] for f in inside); f1 = max(f[1] for f in inside); busy = sum(f[1] - f[0] for f in inside)
    pr = [q for q in prn if q[0] >= d[1] - 1000 and q[0] < d[1] + 30e6][:1]
    print("%2d %9.2f | %7.2f %7.2f %7.2f | %7.2f %+7.2f | %s" % (i, (d[1] - d[0]) / 1e6, (f0 - d[0]) / 1e6, (f1 - d[0]) / 1e6, busy / 1e6, (nxt - d[0]) / 1e6, (nxt - f1) / 1e6,
                                                        "%7.2f %6.2f" % ((pr[0][0] - d[0]) / 1e6, (pr[0][1] - pr[0][0]) / 1e6) if pr else "-"))
