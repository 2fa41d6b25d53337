#!/usr/bin/env python3
"""Generates tests/golden/chain_den_golden.npz from the REFERENCE's LF-MMI denominator (oracle/_ref/bin/ref-chain-den = chain/chain-den-graph.cc +
chain/chain-denominator.cc compiled unmodified, see oracle/build_ref.sh).  Run in the build container (needs /root/reference once):
    python tests/golden/make_chain_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from kaldi_amd import synth
from oracle import chain_oracle as co

CASES = {   # name: (states, pdfs, graph seed, mean degree, hub degree, sequences, frames, output std, leaky, deriv weight)
    "small": (60, 25, 3, 4.0, 30, 5, 12, 2.0, 1.0e-05, -1.0),
    "leaky_large": (150, 40, 4, 6.0, 70, 3, 20, 3.0, 0.1, -0.5),
    "wide_range": (200, 300, 5, 8.0, 90, 4, 9, 12.0, 1.0e-05, -1.0),      # outputs beyond +-30: the exp is limited
}
def make(name):
    S, P, seed, md, hd, B, T, std, leaky, dw = CASES[name]
    f = synth.make_den_fst(S, P, seed=seed, mean_degree=md, hub_degree=hd)
    out = (np.random.default_rng(seed + 100).standard_normal((T * B, P)) * std).astype(np.float32)
    return f, P, out, B, leaky, dw

if __name__ == "__main__":
    assert co.available(), "oracle/_ref/bin/ref-chain-den missing: run oracle/build_ref.sh"
    d = {}
    for name in CASES:
        f, P, out, B, leaky, dw = make(name); r = co.ref_den(f, P, out, B, leaky, dw)
        d[name + ".objf"] = np.float32(r["objf"]); d[name + ".ok"] = np.int32(r["ok"]); d[name + ".initial_probs"] = r["initial_probs"]; d[name + ".deriv"] = r["deriv"]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "chain_den_golden.npz"), **d)
    print("wrote", {k: (v.shape if hasattr(v, "shape") else v) for k, v in d.items()})
