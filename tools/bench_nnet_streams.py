#!/usr/bin/env python3
"""GPU box: the TDNN-F forward of the bench batch (512 x 10 s) as ONE plan on one stream against the same utterances split over S plans on S streams (the tails of one
split's launches -- a last round of tiles that does not fill the chip, the ramps at both ends of 36 launches -- run under the other split's tiles).  Prints ms per forward.
The experiment behind the opt-in two-halves plan of k3_nnet_batch_create (K3_NNET_SPLIT=1; the plans above it are built without; the last line is the library's own split)."""
import os, sys, time, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import nnet3, synth
U = int(sys.argv[1]) if len(sys.argv) > 1 else 512; T = 998; dev = torch.device("cuda:0")
mp = os.path.join(tempfile.gettempdir(), "bns.raw"); synth.make_tdnnf(seed=1).write(mp); net = nnet3.Nnet(mp)
feats = torch.randn(U * T, 40, device=dev)
def run(S, iters=12, lib_split="0"):
    os.environ["K3_NNET_SPLIT"] = lib_split
    per = U // S; nbs = [nnet3.NnetBatch(net, [T] * per, 3) for _ in range(S)]; streams = [torch.cuda.Stream() for _ in range(S)]
    out = torch.empty((sum(nb.total_out_rows for nb in nbs), net.info.output_dim), device=dev); rows = np.cumsum([0] + [nb.total_out_rows for nb in nbs])
    def once():
        ev = torch.cuda.Event(); ev.record()
        for s in range(S):
            streams[s].wait_event(ev)
            with torch.cuda.stream(streams[s]): nbs[s].forward(feats[s * per * T:(s + 1) * per * T], out=out[rows[s]:rows[s + 1]])
        for s in range(S): torch.cuda.current_stream().wait_stream(streams[s])
    for _ in range(3): once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): once()
    torch.cuda.synchronize(); return 1000 * (time.perf_counter() - t0) / iters, out
ref = None
for S in (1, 2, 4, 1, 2):
    ms, out = run(S)
    if ref is None: ref = out.clone()
    print("splits", S, "ms per forward %.2f" % ms, "identical to one plan:", bool(torch.equal(out, ref)), flush=True)
os.environ["K3_NNET_SPLIT"] = "1"
nb = nnet3.NnetBatch(net, [T] * U, 3); out = torch.empty((nb.total_out_rows, net.info.output_dim), device=dev)
for _ in range(3): nb.forward(feats, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(12): nb.forward(feats, out=out)
torch.cuda.synchronize(); print("K3_NNET_SPLIT=1 (two halves inside k3_nnet_forward, caller on the default stream): ms per forward %.2f" % (1000 * (time.perf_counter() - t0) / 12), "identical to one plan:", bool(torch.equal(out, ref)))
