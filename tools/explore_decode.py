#!/usr/bin/env python3
"""exploration: full pipeline statistics on the GPU box (tokens/links per utterance, kernel times)."""
import os, sys, time, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
if not os.environ.get("K3HIP_LIB"): ge.build()
from kaldi_amd import feat, nnet3, synth, decoder
U = int(sys.argv[1]) if len(sys.argv) > 1 else 64
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
dev = torch.device("cuda:0"); nsamp = int(16000 * secs)
g = torch.Generator(device="cpu"); g.manual_seed(1234)
waves = (torch.randn(U * nsamp, generator=g) * 3000).round().clamp(-32768, 32767).to(dev)
sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
wo, fo, total_frames, fo_h = sf.offsets([nsamp] * U, dev)
calib = sf.ComputeFeatures(waves[:nsamp].contiguous(), *sf.offsets([nsamp], dev)[:3]).cpu().numpy()[:600]
mp = os.path.join(tempfile.gettempdir(), "explore.raw"); synth.make_tdnnf(seed=1, calib_feats=calib).write(mp)
net = nnet3.Nnet(mp); nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3)
feats = sf.ComputeFeatures(waves, wo, fo, total_frames); ll = nb.forward(feats); torch.cuda.synchronize()
print("loglikes", tuple(ll.shape), "mean", ll.mean().item(), "std", ll.std().item(), "max", ll.max().item(), "row-max mean", ll.max(dim=1).values.mean().item())
t = time.time(); f = synth.make_hclg(); print("graph", f.stats(), "gen %.1fs" % (time.time() - t))
t = time.time(); cf = decoder.CudaFst(f, synth.tid2pdf(net.info.output_dim)); print("upload %.2fs" % (time.time() - t))
cfg = decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, min_active=int(os.environ.get("MINACT", 200)), literal_order=int(os.environ.get("LITERAL", 0)), frame_tokens_cap=65536, frame_cands_cap=131072,
                             lane_tokens_cap=int(os.environ.get("TOKCAP", 6_000_000)), lane_links_cap=int(os.environ.get("LINKCAP", 12_000_000)))
dec = decoder.CudaDecoder(cf, cfg, U, net.info.output_dim); dec.SetProfiling(True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(2):
    ev[0].record(); dec.DecodeBatch(ll, nb.out_offsets); ev[1].record(); torch.cuda.synchronize()
    print("decode (forward+prune) ms:", ev[0].elapsed_time(ev[1]), "kernels (token passing, prune):", dec.KernelTimes())
info = dec.LatticeInfo(check=False)
print("status", np.unique(info[:, 2], return_counts=True), "reached_final", info[:, 3].mean())
for k, name in enumerate(decoder.CudaDecoder.INFO): print(name, "min/mean/max", info[:, k].min(), info[:, k].mean(), info[:, k].max())
t = time.time(); lats = dec.GetRawLattices(); print("GetRawLattices %.3fs" % (time.time() - t))
st = dec.FrameStats(0, int(nb.out_offsets[1])); print("ntoks[:30]", st["ntoks"][:30], "mean", st["ntoks"].mean(), "ab", st["adaptive_beam"][:10])
audio = U * secs; print("audio-s", audio)
import ctypes
from kaldi_amd import lib as _l
cyc = np.zeros(16, np.int64); _l.load().k3_decoder_phase_cycles(dec._h, cyc.ctypes.data)
names = ["(loop top/err)", "cutoff", "prepass", "pass1 expand", "-", "pass2 claim+links", "-", "closure init", "-", "closure exit check", "finalize+clear", "round: top", "round: wl read", "round: cost+offs+exch", "round: expand(arcs,claims,links)", "round: recycle+barrier"]
tot = cyc.sum()
if tot: print("phase share:", {n: round(100.0 * c / tot, 1) for n, c in zip(names, cyc) if n != "-"}, "cycles/lane/frame", tot / U / 333 / 2)
allnt = np.concatenate([dec.FrameStats(u, int(nb.out_offsets[u + 1] - nb.out_offsets[u]))["ntoks"] for u in range(0, U, max(1, U // 32))])
print("ntoks percentiles 10/50/75/90/95/99/max:", [int(np.percentile(allnt, q)) for q in (10, 50, 75, 90, 95, 99, 100)], "frac <= 1024:", float((allnt <= 1024).mean()), "<= 2048:", float((allnt <= 2048).mean()), "<= 3072:", float((allnt <= 3072).mean()))
if os.environ.get("K3HIP_LIB", "").endswith("_prof.so"):
    xs, ys = [], []
    for u in range(0, U, max(1, U // 64)):
        st = dec.FrameStats(u, int(nb.out_offsets[u + 1] - nb.out_offsets[u])); xs.append(st["ntoks"][1:]); ys.append(st["adaptive_beam"][1:])
    x = np.concatenate(xs).astype(np.float64); y = np.concatenate(ys).astype(np.float64)
    A = np.stack([np.ones_like(x), x], 1); coef = np.linalg.lstsq(A, y, rcond=None)[0]
    print("frame cycles ~ %.0f + %.2f * ntoks; mean cycles %.0f, mean ntoks %.0f; fixed share %.2f" % (coef[0], coef[1], y.mean(), x.mean(), coef[0] / y.mean()))
    for lo, hi in ((0, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 4096), (4096, 8192), (8192, 1 << 30)):
        m = (x >= lo) & (x < hi)
        if m.any(): print("  ntoks [%d,%d): frames %.3f, time share %.3f, mean cycles %.0f" % (lo, hi, m.mean(), y[m].sum() / y.sum(), y[m].mean()))
