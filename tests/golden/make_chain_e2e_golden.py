#!/usr/bin/env python3
"""Generates tests/golden/chain_e2e_golden.npz from the REFERENCE's end-to-end LF-MMI objective: chain::ComputeChainObjfAndDeriv with Supervision::e2e_fsts set, i.e.
chain::GenericNumeratorComputation (chain/chain-generic-numerator.cc) + the denominator, all compiled unmodified into oracle/_ref/bin/ref-chain-objf (oracle/build_ref.sh).
Run in the build container:   python tests/golden/make_chain_e2e_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from kaldi_amd import synth
from oracle import chain_oracle as co

CASES = {   # name: (den states, pdfs, den seed, mean degree, hub degree, leaky, sequences, frames, phones (None = frames // 5), supervision seed, output std, l2_regularize, supervision weight)
    "e2e_small": (60, 25, 3, 4.0, 30, 1.0e-05, 5, 24, None, 140, 2.0, 0.0, 1.0),
    "e2e_l2_weight": (150, 40, 4, 6.0, 70, 0.1, 3, 40, 12, 150, 3.0, 0.0005, 0.7),
    "e2e_long": (100, 60, 6, 5.0, 40, 1.0e-05, 4, 150, None, 160, 1.5, 0.0, 1.0),
}
def make(name):
    S, P, seed, md, hd, leaky, B, T, K, sseed, std, l2, w = CASES[name]
    den = synth.make_den_fst(S, P, seed=seed, mean_degree=md, hub_degree=hd)
    fsts = [synth.make_e2e_fst(T, P, seed=sseed + i, num_phones=K) for i in range(B)]
    out = (np.random.default_rng(sseed + 7).standard_normal((T * B, P)) * std).astype(np.float32)
    return den, P, fsts, out, leaky, l2, w

if __name__ == "__main__":
    assert co.objf_available(), "oracle/_ref/bin/ref-chain-objf missing: run oracle/build_ref.sh"
    d = {}
    for name in CASES:
        den, P, fsts, out, leaky, l2, w = make(name); r = co.ref_objf_e2e(den, P, fsts, out, leaky, l2, w)
        for k in ("objf", "l2_term", "weight"): d[f"{name}.{k}"] = np.float32(r[k])
        d[name + ".deriv"] = r["deriv"]; d[name + ".xent_deriv"] = r["xent_deriv"]
        print(name, r["objf"], r["l2_term"], r["weight"], float(np.abs(r["deriv"]).max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "chain_e2e_golden.npz"), **d)
