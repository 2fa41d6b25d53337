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
    # BASELINE configs[2] as bench.py runs it: the 2 M-state / 5 M-arc graph, 6024 pdfs, one 10 s utterance (333 frames), beam 15,
    # lattice-beam 8, max-active 10000 (which binds on the busiest frames: up to ~23 k tokens)
    "bench_config": (3, None, 333, dict(max_active=10000), 6024),
}

_bench_graph = None
def make(name):
    global _bench_graph
    case = CASES[name]; seed, size, T, kw = case[:4]; n = case[4] if len(case) > 4 else N
    if size is None:                                   # the bench graph (built once per process: ~5 s)
        if _bench_graph is None: _bench_graph = synth.make_hclg(num_pdfs=n)
        f = _bench_graph; ll = (np.random.default_rng(seed).standard_normal((T, n)) * 2.0).astype(np.float32)
    else:
        f = synth.make_hclg(size[0], size[1], n, seed=seed, start_degree=30)
        ll = (np.random.default_rng(seed + 1).standard_normal((T, n)) * 2.5).astype(np.float32)
    return f, synth.tid2pdf(n), ll, dict(dict(beam=15.0, lattice_beam=8.0), **kw)
