#!/usr/bin/env python3
"""GPU box, K3HIP_LIB=build/libk3hip_framecyc.so (tools/build_variant.sh framecyc -DK3_LIT_FRAMECYC -Os): the literal_order kernel frame by frame at the bench configuration --
shader cycles of every frame of every lane against the number of tokens it was built from, by path (LDS-resident / general).  Prints the table DESIGN.md 4 quotes and writes
gpurun_out/literal_frames_by_size.json."""
import json, os, sys, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import feat, nnet3, synth, decoder
U = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0"); nsamp = 160000
waves = torch.cat([torch.from_numpy(synth.gaussian_pcm16(nsamp, 1234 + i).astype(np.float32)) for i in range(U)]).to(dev)
sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40)); wo, fo, total, fo_h = sf.offsets([nsamp] * U, dev)
feats = sf.ComputeFeatures(waves, wo, fo, total)
mp = os.path.join(tempfile.gettempdir(), "profframes.raw"); synth.make_tdnnf(seed=1, calib_feats=feats[:600].cpu().numpy()).write(mp)
net = nnet3.Nnet(mp); nb = nnet3.NnetBatch(net, [fo_h[i + 1] - fo_h[i] for i in range(U)], 3); ll = nb.forward(feats); torch.cuda.synchronize()
cf = decoder.CudaFst(synth.make_hclg(), synth.tid2pdf(net.info.output_dim))
cfg = decoder.decoder_config(beam=15.0, lattice_beam=8.0, max_active=10000, frame_tokens_cap=65536, frame_cands_cap=131072, lane_tokens_cap=1_600_000, lane_links_cap=2_200_000, literal_order=1)
dec = decoder.CudaDecoder(cf, cfg, U, net.info.output_dim); dec.SetProfiling(True)
for _ in range(2): dec.DecodeBatch(ll, nb.out_offsets); torch.cuda.synchronize()
kt = dec.KernelTimes(); info = dec.LatticeInfo(check=False)
nt, cyc = [], []
for u in range(U):
    st = dec.FrameStats(u); nt.append(st["ntoks"]); cyc.append(st["adaptive_beam"])
nt = np.concatenate(nt).astype(np.int64); cyc = np.concatenate(cyc).astype(np.float64)
cyc[~np.isfinite(cyc)] = 0.0      # (frame 0 is made by the template kernel: its slot holds the adaptive beam, +inf, not a cycle count)
fast = cyc > 0; cyc = np.abs(cyc)
edges = [0, 256, 512, 768, 1024, 1280, 1536, 1792, 2048, 2560, 3072, 4096, 6144, 8192, 16384, 65536]
tot = cyc.sum(); rows = []
print("token passing ms %.2f | frames %d | cycles per lane %.1f M | tokens/frame mean %.0f median %.0f" % (kt[0], nt.size, tot / U / 1e6, nt.mean(), np.median(nt)))
print("%-14s %8s %7s %7s | %9s %9s | %7s %7s" % ("tokens in", "frames", "frac", "on LDS", "cyc LDS", "cyc gen", "%cycles", "cum%"))
cum = 0.0
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (nt > lo) & (nt <= hi)
    if not m.any(): continue
    f_ = m & fast; g_ = m & ~fast; share = cyc[m].sum() / tot; cum += share
    rows.append({"tokens_lo": lo, "tokens_hi": hi, "frames": int(m.sum()), "frames_frac": float(m.mean()), "lds_path_frac": float(f_.sum() / m.sum()), "cycles_lds_mean": float(cyc[f_].mean()) if f_.any() else None,
                 "cycles_general_mean": float(cyc[g_].mean()) if g_.any() else None, "cycles_share": float(share)})
    print("%6d..%-6d %8d %7.3f %7.3f | %9.0f %9.0f | %7.2f %7.2f" % (lo + 1, hi, m.sum(), m.mean(), f_.sum() / m.sum(), cyc[f_].mean() if f_.any() else 0, cyc[g_].mean() if g_.any() else 0, 100 * share, 100 * cum))
first = np.concatenate([np.arange(333) for _ in range(U)])[:nt.size] if nt.size == 333 * U else None
if first is not None:
    head = first < 12
    print("first 12 frames of every utterance: %.2f%% of the frames, %.2f%% of the cycles, mean tokens %.0f" % (100 * head.mean(), 100 * cyc[head].sum() / tot, nt[head].mean()))
if first is not None:
    print("by frame index: frame, mean tokens in, mean cycles (k), share of the lane's cycles %, on LDS path")
    for fi in list(range(16)) + [20, 30, 50, 100, 200, 332]:
        m = first == fi; print("  %3d %8.0f %10.0f %6.2f %6.3f" % (fi, nt[m].mean(), cyc[m].mean() / 1e3, 100 * cyc[m].sum() / tot, fast[m].mean()))
if first is not None:
    per_lane = cyc.reshape(U, 333).sum(1); tok_lane = nt.reshape(U, 333).sum(1)
    q = np.percentile(per_lane, [0, 10, 50, 90, 100]) / 1e6
    print("cycles per lane (M): min %.1f p10 %.1f median %.1f p90 %.1f max %.1f | corr with the lane's token count %.3f" % (*q, np.corrcoef(per_lane, tok_lane)[0, 1]))
    print("mean cycles per lane (M) by lane %% 8:", np.round([per_lane[k::8].mean() / 1e6 for k in range(8)], 1).tolist(), "| first / second half of the lanes:", round(per_lane[:U // 2].mean() / 1e6, 1), round(per_lane[U // 2:].mean() / 1e6, 1))
    print("cycles per token of the lane: min %.0f median %.0f max %.0f" % tuple(np.percentile(per_lane / tok_lane, [0, 50, 100])))
    print("shader clock implied by the slowest lane: %.1f M cycles of frames / %.2f ms of kernel = %.2f GHz (a lower bound: the lane's prologue and InitDecoding are not in its frame cycles)" % (per_lane.max() / 1e6, kt[0], per_lane.max() / kt[0] / 1e6))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"token_passing_ms": kt[0], "lanes": U, "cycles_per_lane": tot / U, "rows": rows, "head12_cycles_share": float(cyc[head].sum() / tot) if first is not None else None},
          open(os.path.join(ROOT, "gpurun_out", "literal_frames_by_size.json"), "w"), indent=1)
