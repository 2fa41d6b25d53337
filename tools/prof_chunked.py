#!/usr/bin/env python3
"""GPU box: the bench batch decoded in chunks (AdvanceDecoding every C decoder frames, nothing else on the GPU): token-passing time per call against the one-call decode.
  python tools/prof_chunked.py [lanes=512] [chunk frames=17]"""
import os, sys, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import feat, nnet3, synth, decoder
U = int(sys.argv[1]) if len(sys.argv) > 1 else 512; C = int(sys.argv[2]) if len(sys.argv) > 2 else 17
dev = torch.device("cuda:0"); nsamp = 160000
waves = torch.cat([torch.from_numpy(synth.gaussian_pcm16(nsamp, 1234 + i).astype(np.float32)) for i in range(U)]).to(dev)
sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40)); wo, fo, total, fo_h = sf.offsets([nsamp] * U, dev)
feats = sf.ComputeFeatures(waves, wo, fo, total)
mp = os.path.join(tempfile.gettempdir(), "profchunk.raw"); synth.make_tdnnf(seed=1, calib_feats=feats[:600].cpu().numpy()).write(mp)
net = nnet3.Nnet(mp); nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3); ll = nb.forward(feats); torch.cuda.synchronize()
oo = np.asarray(nb.out_offsets); T = int(oo[1] - oo[0])
cf = decoder.CudaFst(synth.make_hclg(), synth.tid2pdf(net.info.output_dim))
cfg = decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, frame_tokens_cap=65536, frame_cands_cap=131072, lane_tokens_cap=1_600_000, lane_links_cap=2_200_000, literal_order=1)
dec = decoder.CudaDecoder(cf, cfg, U, net.info.output_dim); dec.SetProfiling(True)
for _ in range(2): dec.DecodeBatch(ll, nb.out_offsets); torch.cuda.synchronize()
print("one call: token passing ms %.2f" % dec.KernelTimes()[0])
# chunked: lane u's rows [oo[u] + f0, oo[u] + f1) gathered into one block per call (row offsets k * n)
for rep in range(2):
    dec.InitDecoding(U, T + 8); times = []
    for f0 in range(0, T, C):
        n = min(C, T - f0)
        idx = (torch.from_numpy(oo[:-1]).to(dev)[:, None] + torch.arange(f0, f0 + n, device=dev)[None, :]).reshape(-1)
        blk = ll.index_select(0, idx).contiguous(); torch.cuda.synchronize()
        dec.AdvanceDecoding(blk, np.arange(U + 1, dtype=np.int64) * n); torch.cuda.synchronize()
        times.append(dec.KernelTimes()[0])
    dec.FinalizeDecoding(); torch.cuda.synchronize()
    t = np.array(times); print("chunks of %d frames: %d calls, token passing ms per call: first %.2f second %.2f median %.2f mean %.2f | sum %.2f ms" % (C, len(t), t[0], t[1], np.median(t), t.mean(), t.sum()))
if os.environ.get("K3HIP_LIB", "").endswith("framecyc.so"):      # per-frame shader cycles (FrameStats' adaptive_beam column in that build): the frames of a call by their position in it
    cyc = np.stack([np.abs(np.asarray(dec.FrameStats(u)["adaptive_beam"], np.float64))[:T] for u in range(0, U, 8)])      # [lanes/8, T]
    cyc[~np.isfinite(cyc)] = 0.0
    pos = np.arange(T) % C
    print("mean cycles (k) of a frame by its position in the call (frames >= %d only):" % (2 * C), [int(cyc[:, (pos == k) & (np.arange(T) >= 2 * C)].mean() / 1e3) for k in range(C)])
    per_call = cyc[:, 2 * C:(T // C) * C].reshape(cyc.shape[0], -1, C).sum(2)      # [lanes/8, calls]
    print("cycles of a lane's frames per call (M): mean %.2f, mean over calls of the max over lanes %.2f" % (per_call.mean() / 1e6, per_call.max(0).mean() / 1e6))
