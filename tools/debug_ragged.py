#!/usr/bin/env python3
"""GPU box: the literal decoder on a ragged batch (U(2 s, 20 s), seed 1235) in generation order (the library sorts the lanes itself) and pre-sorted by the caller: kernel times."""
import os, sys, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import feat, nnet3, synth, decoder, lib as _l
U = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
lens0 = (np.random.default_rng(1235).uniform(2.0, 20.0, U) * 16000).astype(np.int64)
sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
w0 = torch.from_numpy(synth.gaussian_pcm16(160000, 1234).astype(np.float32)).to(dev)
calib = sf.ComputeFeatures(w0, *sf.offsets([160000], dev)[:3]).cpu().numpy()[:600]
mp = os.path.join(tempfile.gettempdir(), "dbgragged.raw"); synth.make_tdnnf(seed=1, calib_feats=calib).write(mp)
net = nnet3.Nnet(mp)
cf = decoder.CudaFst(synth.make_hclg(), synth.tid2pdf(net.info.output_dim))
for name, lens in (("generation order", lens0), ("sorted by the caller", np.sort(lens0)[::-1].copy())):
    lens = [int(x) for x in lens]
    waves = torch.cat([torch.from_numpy(synth.gaussian_pcm16(n, 1234 + i).astype(np.float32)) for i, n in enumerate(lens)]).to(dev)
    wo, fo, total, fo_h = sf.offsets(lens, dev)
    feats = sf.ComputeFeatures(waves, wo, fo, total)
    nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3); ll = nb.forward(feats); torch.cuda.synchronize()
    dec = decoder.CudaDecoder(cf, decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, frame_tokens_cap=65536, frame_cands_cap=131072,
                                                         lane_tokens_cap=int(4500 * 20 * 33.4) + 65536, lane_links_cap=int(6000 * 20 * 33.4) + 131072, literal_order=1), U, net.info.output_dim)
    dec.SetProfiling(True)
    for it in range(3):
        dec.DecodeBatch(ll, nb.out_offsets); torch.cuda.synchronize()
        info = dec.LatticeInfo()
        print(name, "token passing ms %.2f prune ms %.2f" % dec.KernelTimes(), "tokens", int(info[:, 4].sum()), "frames", int(info[:, 9].sum()), "tokens/frame of the 8 shortest / longest lanes",
              [int(info[u, 4] / max(1, info[u, 9])) for u in np.argsort(lens)[:8]], [int(info[u, 4] / max(1, info[u, 9])) for u in np.argsort(lens)[-8:]], flush=True)
    del dec
