#!/usr/bin/env python3
"""GPU box: time of the LF-MMI denominator launch (k3_chain_den_forward_backward, forward + backward) for a few graph / minibatch sizes."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import synth, chain
for S, P, md, B, T in ((3000, 4000, 12.0, 64, 50), (3000, 4000, 2.0, 64, 50), (3000, 4000, 12.0, 128, 50), (3000, 4000, 12.0, 256, 50), (3000, 4000, 12.0, 8, 50), (1000, 1000, 12.0, 128, 50)):
    f = synth.make_den_fst(S, P, mean_degree=md); g = chain.DenominatorGraph(f, P)
    out = (torch.randn(T * B, P, device="cuda") * 2.0).contiguous(); d = torch.zeros_like(out)
    comp = chain.DenominatorComputation(chain.ChainTrainingOptions(1e-5), g, B, out)
    comp.Backward(-1.0, d); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): comp.Backward(-1.0, d)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    comp.Forward(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): comp.Forward()
    torch.cuda.synchronize(); msf = (time.perf_counter() - t0) / 10 * 1e3
    E = int(f.arc_offsets[-1])
    print(f"states {S} transitions {E} pdfs {P} sequences {B} frames {T}: forward+backward {ms:.3f} ms, forward only {msf:.3f} ms; per sequence-frame-step {ms * 1e3 / (2 * T):.1f} us; edge-steps/s {E * B * T * 2 / ms / 1e6:.1f} G")
