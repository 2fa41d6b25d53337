#!/usr/bin/env python3
"""GPU box, K3HIP_LIB=build/libk3hip_stats<NT>.so (-DK3_LIT_STATS=NT): sizes of the closure sub-graph / queue / label space of the frames with at most NT tokens (capacity planning of the LDS-resident frame path)."""
import os, sys, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import feat, nnet3, synth, decoder, lib as _l
U = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0"); nsamp = 160000
waves = torch.cat([torch.from_numpy(synth.gaussian_pcm16(nsamp, 1234 + i).astype(np.float32)) for i in range(U)]).to(dev)
sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40)); wo, fo, total, fo_h = sf.offsets([nsamp] * U, dev)
feats = sf.ComputeFeatures(waves, wo, fo, total)
mp = os.path.join(tempfile.gettempdir(), "litstats.raw"); synth.make_tdnnf(seed=1, calib_feats=feats[:600].cpu().numpy()).write(mp)
net = nnet3.Nnet(mp); nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3); ll = nb.forward(feats); torch.cuda.synchronize()
cf = decoder.CudaFst(synth.make_hclg(), synth.tid2pdf(net.info.output_dim))
dec = decoder.CudaDecoder(cf, decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, frame_tokens_cap=65536, frame_cands_cap=131072, lane_tokens_cap=1_600_000, lane_links_cap=2_200_000, literal_order=1), U, net.info.output_dim)
dec.DecodeBatch(ll, nb.out_offsets); torch.cuda.synchronize(); info = dec.LatticeInfo()
h = np.zeros(16 * U, np.int64)
import ctypes
_l.check(ctypes.c_int(0).value)
# raw per-lane counters (k3_decoder_phase_cycles sums over lanes; maxima need the per-lane rows)
L = _l.load(); buf = np.zeros(16, np.int64); L.k3_decoder_phase_cycles(dec._h, buf.ctypes.data)
fr = max(1, buf[5])
print("summed over lanes: frames", buf[5], "of", 333 * U, "| n_cid>1024:", buf[6] / fr, "n_cid>1536:", buf[7] / fr, "n_iq>1024:", buf[8] / fr, "eps links>2048:", buf[9] / fr, "labels>32768:", buf[10] / fr, "n_arc>2048:", buf[15] / fr)
print("means: n_cid", buf[11] / fr, "n_arc", buf[12] / fr, "n_iq", buf[13] / fr, "eps links", buf[14] / fr, "| sums of per-lane maxima / U: n_cid", buf[0] / U, "n_arc", buf[1] / U, "n_iq", buf[2] / U, "eps links", buf[3] / U, "labels", buf[4] / U)
