#!/usr/bin/env python3
"""tools/train_trace_report.py <tag> [iteration]: what one training iteration of tools/train_trace.sh's traces spent its time on -- kernels by name (count, total), gaps between kernels
(host-side waits), and the GEMMs by shape class (timed one by one, ~12 us of synchronisation included in each)."""
import sys, os, re, collections, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); d = os.path.join(ROOT, "gpurun_out", sys.argv[1]); which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
R = [l.split() for l in open(os.path.join(d, "kernel_trace_compact.txt"))]; R = [(int(a), int(b), c) for a, b, c in R]
den = [i for i, r in enumerate(R) if "k3_chain_den" in r[2]]
print("kernels per iteration:", [den[i + 1] - den[i] for i in range(len(den) - 1)])
i0, i1 = den[which - 1], den[which]; seg = R[i0:i1]; span = seg[-1][0] + seg[-1][1] - seg[0][0]; busy = sum(r[1] for r in seg)
g = np.array([seg[i + 1][0] - (seg[i][0] + seg[i][1]) for i in range(len(seg) - 1)])
print("iteration %d: span %.2f ms, kernels busy %.2f ms (%d launches), gaps %.2f ms (> 20 us: %d = %.2f ms; > 100 us: %d = %.2f ms)" % (which, span / 1e6, busy / 1e6, len(seg), g[g > 0].sum() / 1e6, (g > 20000).sum(), g[g > 20000].sum() / 1e6, (g > 100000).sum(), g[g > 100000].sum() / 1e6))
agg = collections.Counter(); cnt = collections.Counter()
for s, dur, n in seg: k = re.sub(r"\(anonymous_namespace\)::|void_", "", n)[:48]; agg[k] += dur; cnt[k] += 1
for n, dur in agg.most_common(18): print("%7.2f ms %5d  %s" % (dur / 1e6, cnt[n], n))
L = [l for l in open(os.path.join(d, "gemm_trace.log")) if l.startswith("k3 gemm") or "frames;" in l]; its = [[]]
for l in L:
    if "frames;" in l: its.append([])
    else: its[-1].append(l)
it = its[which - 1 if which < 0 else which]; agg = {}
for l in it:
    m = re.match(r"k3 gemm M (\d+) N (\d+) K (\d+) ta (\d) tb (\d) lda (\d+) ldb (\d+) beta (\S+) us (\S+)", l); M, N, K, ta, tb = map(int, m.groups()[:5]); us = float(m.group(9))
    cls = lambda x: x if x < 2000 else (14300 if x > 10000 else 4700)
    a = agg.setdefault((cls(M), N, cls(K), ta, tb, m.group(8)), [0, 0.0, 0]); a[0] += 1; a[1] += us; a[2] += 2 * M * N * K
print(len(it), "gemms, %.2f ms one by one, %.1f GFLOP" % (sum(a[1] for a in agg.values()) / 1e3, sum(a[2] for a in agg.values()) / 1e9))
for k, a in sorted(agg.items(), key=lambda x: -x[1][1])[:int(os.environ.get("TOP", "30"))]: print(k, "n", a[0], "us/call %.1f" % (a[1] / a[0]), "total ms %.2f" % (a[1] / 1e3), "TFLOP/s %.1f" % (a[2] / a[1] / 1e6))
