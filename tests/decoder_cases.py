"""Decoder cases shared by the reference-pinning test and the script that records the golden digests."""
import numpy as np
from kaldi_amd import synth

N = 40
CASES = {   # name: (seed, graph (states, arcs), frames, LatticeFasterDecoderConfig overrides)
    "default": (1, (1500, 4000), 50, {}),
    "default_b": (2, (1500, 4000), 50, {}),
    "max_active": (3, (1500, 4000), 50, dict(max_active=300)),
    "narrow_beams": (4, (1500, 4000), 50, dict(beam=8.0, lattice_beam=4.0)),
    "min_active_loosens": (5, (1500, 4000), 50, dict(min_active=2000)),
    "prune_every_3": (6, (1500, 4000), 50, dict(prune_interval=3)),
    "hash_ratio": (7, (1500, 4000), 50, dict(hash_ratio=3.7)),
    "both_limits": (8, (1500, 4000), 50, dict(max_active=150, min_active=50, beam_delta=0.25)),
    "big_graph_hash_order": (9, (40000, 100000), 40, dict(hash_ratio=2.0)),
    "long": (10, (3000, 8000), 160, dict(beam=13.0, lattice_beam=6.0, prune_interval=25)),
}

def make(name):
    seed, (S, A), T, kw = CASES[name]
    f = synth.make_hclg(S, A, N, seed=seed, start_degree=30)
    ll = (np.random.default_rng(seed + 1).standard_normal((T, N)) * 2.5).astype(np.float32)
    return f, synth.tid2pdf(N), ll, dict(dict(beam=15.0, lattice_beam=8.0), **kw)
