#!/usr/bin/env python3
"""GPU box, K3HIP_LIB=build/libk3hip_fp.so (-DK3_FAST_PROF): cycles per phase of the LDS-resident frame path (k3_decoder_fast.h), bench configuration."""
import os, sys, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import feat, nnet3, synth, decoder, lib as _l
U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0"); nsamp = 160000
waves = torch.cat([torch.from_numpy(synth.gaussian_pcm16(nsamp, 1234 + i).astype(np.float32)) for i in range(U)]).to(dev)
sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40)); wo, fo, total, fo_h = sf.offsets([nsamp] * U, dev)
feats = sf.ComputeFeatures(waves, wo, fo, total)
mp = os.path.join(tempfile.gettempdir(), "proffast.raw"); synth.make_tdnnf(seed=1, calib_feats=feats[:600].cpu().numpy()).write(mp)
net = nnet3.Nnet(mp); nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3); ll = nb.forward(feats); torch.cuda.synchronize()
cf = decoder.CudaFst(synth.make_hclg(), synth.tid2pdf(net.info.output_dim))
dec = decoder.CudaDecoder(cf, decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, frame_tokens_cap=65536, frame_cands_cap=131072, lane_tokens_cap=1_600_000, lane_links_cap=2_200_000, literal_order=1), U, net.info.output_dim)
dec.SetProfiling(True)
for it in range(2): dec.DecodeBatch(ll, nb.out_offsets); torch.cuda.synchronize()
print("token passing ms %.2f prune ms %.2f" % dec.KernelTimes())
cyc = np.zeros(16, np.int64); _l.load().k3_decoder_phase_cycles(dec._h, cyc.ctypes.data)
names = ["cutoff", "init+passA", "scan+passB", "c0/labels", "closure", "buckets+subgraph", "order1", "queue+replay", "labels", "order2+publish"]
names += ["queue+union (of queue+replay)", "count+offsets+roots (of queue+replay)"]      # fs.prof[10], [11]; "queue+replay" is then the workers alone
fr = max(1, cyc[12]); tot = cyc[:12].sum()
print("fast frames", cyc[12], "gave up", cyc[13], "general", cyc[14], "| cycles per fast frame", tot / fr, "(whole call incl. import: %.0f)" % (cyc[15] / fr))
print({n: "%.0f (%.0f%%)" % (c / fr, 100.0 * c / tot) for n, c in zip(names, cyc[:12])})
