#!/usr/bin/env python3
"""GPU box: the first-frame template on the bench graph with a few lanes of random log-likelihoods: with / without the template, lattices compared (developer aid)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import synth, decoder
U = int(sys.argv[1]) if len(sys.argv) > 1 else 4; T = int(sys.argv[2]) if len(sys.argv) > 2 else 6; N = 6024
rng = np.random.default_rng(3)
cf = decoder.CudaFst(synth.make_hclg(), synth.tid2pdf(N))
lls = [(rng.standard_normal((T, N)) * 0.7 - 8.0).astype(np.float32) for _ in range(U)]
ro = np.concatenate([[0], np.cumsum([l.shape[0] for l in lls])]); ll = torch.from_numpy(np.concatenate(lls)).cuda()
out = []
for no in (False, True):
    if no: os.environ["K3_LIT_NO_FRAME0_TEMPLATE"] = "1"
    dec = decoder.CudaDecoder(cf, decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, frame_tokens_cap=65536, frame_cands_cap=131072, lane_tokens_cap=600000, lane_links_cap=900000, literal_order=1), U, N)
    dec.DecodeBatch(ll, ro); torch.cuda.synchronize(); print("decoded", "without" if no else "with", "template", flush=True)
    info = dec.LatticeInfo(); print(info[:, :10], flush=True)
    out.append(dec.GetRawLattices(copy=True))
for u in range(U): print(u, "diff:", repr(out[0][u].diff(out[1][u])[:200]))
