#!/usr/bin/env python3
"""Developer aid (GPU box): decode small cases with literal_order and print where the HIP decoder first departs from the oracle's literal mode."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import decoder, synth
from oracle import lattice_oracle as lo
from tests import decoder_cases as dcases
names = sys.argv[1:] or ["default", "max_active", "hash_ratio", "big_graph_hash_order", "long"]
caps = dict(frame_tokens_cap=65536, frame_cands_cap=262144, lane_tokens_cap=2_500_000, lane_links_cap=3_500_000)
for name in names:
    f, t2p, ll, kw = dcases.make(name); N = ll.shape[1]
    k2 = {k: v for k, v in kw.items() if k in ("beam", "max_active", "min_active", "lattice_beam", "beam_delta", "hash_ratio")}
    cf = decoder.CudaFst(f, t2p)
    dec = decoder.CudaDecoder(cf, decoder.decoder_config(literal_order=1, **dict(caps, **k2)), 1, N)
    dec.DecodeBatch(torch.from_numpy(ll).cuda(), np.array([0, ll.shape[0]]))
    info = dec.LatticeInfo(check=False)
    print(name, "info", info[0].tolist())
    if info[0, 2] < 0: continue
    ref, oi = lo.decode(f, ll, t2p, lo.Config(**kw), mode=0)
    st = dec.FrameStats(0)
    bad = None
    for fr in range(ll.shape[0]):
        if st["ntoks"][fr] != oi["ntoks"][fr] or any(st[k][fr:fr + 1].view(np.int32)[0] != oi[k][fr:fr + 1].view(np.int32)[0] for k in ("cur_cutoff", "adaptive_beam", "next_cutoff", "cost_offset")):
            bad = fr; break
    if bad is not None:
        print("  first differing frame", bad, {k: (st[k][max(0, bad - 1):bad + 2].tolist(), oi[k][max(0, bad - 1):bad + 2].tolist()) for k in st})
    lat = dec.GetRawLattices()[0]
    d = lat.diff(ref)
    print("  lattice", lat.num_states, lat.num_arcs, "oracle", ref.num_states, ref.num_arcs, "order-sensitive", int(dec.OrderSensitiveEvents()[0]), oi["extra_links"], "DIFF: " + d[:300] if d else "identical")
