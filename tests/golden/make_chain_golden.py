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

OBJF_CASES = {   # name: (den case, sequences, frames, supervision seed, output std, l2_regularize, supervision weight)
    "objf_small": ("small", 5, 12, 40, 2.0, 0.0, 1.0),
    "objf_l2_weight": ("leaky_large", 3, 20, 50, 3.0, 0.0005, 0.7),
}
def make_objf(name):
    den_case, B, T, sseed, std, l2, w = OBJF_CASES[name]; S, P, seed, md, hd = CASES[den_case][:5]; leaky = CASES[den_case][8]
    den = synth.make_den_fst(S, P, seed=seed, mean_degree=md, hub_degree=hd)
    fsts = [synth.make_supervision_fst(T, P, seed=sseed + i) for i in range(B)]
    out = (np.random.default_rng(sseed + 7).standard_normal((T * B, P)) * std).astype(np.float32)
    return den, P, fsts, out, leaky, l2, w

if __name__ == "__main__":
    assert co.available(), "oracle/_ref/bin/ref-chain-den missing: run oracle/build_ref.sh"
    d = {}
    for name in CASES:
        f, P, out, B, leaky, dw = make(name); r = co.ref_den(f, P, out, B, leaky, dw)
        d[name + ".objf"] = np.float32(r["objf"]); d[name + ".ok"] = np.int32(r["ok"]); d[name + ".initial_probs"] = r["initial_probs"]; d[name + ".deriv"] = r["deriv"]
    assert co.objf_available()
    for name in OBJF_CASES:
        den, P, fsts, out, leaky, l2, w = make_objf(name); r = co.ref_objf(den, P, synth.merge_supervision_fsts(fsts), out, len(fsts), leaky, l2, w)
        for k, v in r.items(): d[name + "." + k] = np.float32(v) if np.isscalar(v) else v
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "chain_den_golden.npz"), **d)
    print("wrote", {k: (v.shape if hasattr(v, "shape") else v) for k, v in d.items()})
