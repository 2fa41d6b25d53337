#!/usr/bin/env python3
"""Streaming i-vector extraction, cost per chunk round: N channels, each receiving a chunk of C feature frames per call -- k3_ivector_stream_accept_batch (one launch per stage for the
batch) next to k3_ivector_stream_accept channel by channel.  Random model of the recipes' size (40-dim features, splice 3+1+3 -> LDA 40, 512 Gaussians, 100-dim i-vectors, period 10).
  gpurun -- 'python tools/bench_ivector_stream.py [N] [C]'"""
import importlib.util, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd.ivector import BatchedIvectorExtractor, IvectorStream, AcceptFramesBatch
spec = importlib.util.spec_from_file_location("tiv", os.path.join(ROOT, "tests", "test_ivector_gpu.py")); tiv = importlib.util.module_from_spec(spec); spec.loader.exec_module(tiv)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256; C = int(sys.argv[2]) if len(sys.argv) > 2 else 50; ROUNDS = 12
rng = np.random.default_rng(0); F, lc, rc, D, G, R = 40, 3, 3, 40, 512, 100
lda, st, ubm, ie = tiv._random_model(rng, F, lc, rc, D, G, R, False)
il = np.tril_indices(D); packed = np.stack([ie["sigma_inv"][g][il] for g in range(G)])
ex = BatchedIvectorExtractor.FromArrays(lda, st, ubm["gconsts"], ubm["means_invvars"], ubm["inv_vars"], ie["M"], packed, ie["prior_offset"], left_context=lc, right_context=rc, ivector_period=10)
dev = torch.device("cuda:0"); feats = torch.from_numpy((rng.standard_normal((N * C, F)) * 2.0).astype(np.float32)).to(dev); fo = np.arange(N + 1) * C
res = {}
for mode in ("batched", "per_channel"):
    streams = [IvectorStream(ex) for _ in range(N)]; ts = []
    for r in range(ROUNDS):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "batched": AcceptFramesBatch(streams, feats, fo, [False] * N)
        else:
            for i, s in enumerate(streams): s.AcceptFrames(feats[i * C:(i + 1) * C], False)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res[mode] = 1e3 * float(np.median(ts[2:]))
    last = torch.stack([s.Latest() for s in streams]).clone() if mode == "per_channel" else torch.stack([s.Latest() for s in streams]).clone(); res[mode + "_last"] = last
assert torch.equal(res["batched_last"], res["per_channel_last"])
print(f"{N} channels x {C} frames per round ({N * C / 100:.0f} s of audio): batched {res['batched']:.2f} ms, channel by channel {res['per_channel']:.2f} ms per round; identical estimates")
