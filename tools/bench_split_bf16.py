#!/usr/bin/env python3
"""EXPLORATORY: the TDNN-F forward of the benchmark model with its affine products on the bf16 matrix core, operands split three ways
(k3_nnet_batch_set_precision(batch, 1): six bf16 MFMA products per product, operands split by the loader; (batch, 2): the activations' planes written by the producing
epilogue, the loader only loads), next to the FP32 matrix-core forward: time of a forward at the bench size, and the
distance of both to the float64 forward of the same network (oracle/nnet3_oracle.py, the checker) on a few utterances.
  python tools/bench_split_bf16.py [utts=512] [truth_utts=8]"""
import os, sys, time, json, tempfile, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
from kaldi_amd import feat, nnet3, synth
from oracle import nnet3_oracle as no
U = int(sys.argv[1]) if len(sys.argv) > 1 else 512; TU = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0"); nsamp = 160000
sf = feat.SpectralFeatures(feat.fbank_options(dither=0.0, num_bins=40))
pcm = [synth.gaussian_pcm16(nsamp, 1234 + u) for u in range(max(TU, 1))]
w0 = torch.from_numpy(pcm[0].astype(np.float32)).to(dev)
calib = sf.ComputeFeatures(w0, *sf.offsets([nsamp], dev)[:3]).cpu().numpy()[:600]
mp = os.path.join(tempfile.gettempdir(), "k3_x6_model.raw"); synth.make_tdnnf(seed=1, calib_feats=calib).write(mp)
net = nnet3.Nnet(mp)
feats_u = [sf.ComputeFeatures(torch.from_numpy(p.astype(np.float32)).to(dev), *sf.offsets([nsamp], dev)[:3]) for p in pcm]
T = feats_u[0].shape[0]
# ---- accuracy on TU utterances
nb = nnet3.NnetBatch(net, [T] * TU, 3); x = torch.cat(feats_u[:TU], 0).contiguous()
y32 = nb.forward(x).clone(); torch.cuda.synchronize()
nb.set_precision(1); y6 = nb.forward(x).clone(); torch.cuda.synchronize()
onet = no.read_nnet(mp); oo = np.asarray(nb.out_offsets)
e32 = []; e6 = []; d = []
for u in range(TU):
    t64 = no.compute(onet, feats_u[u].cpu().numpy(), 3, dtype=np.float64)
    a = y32[oo[u]:oo[u + 1]].cpu().numpy(); b = y6[oo[u]:oo[u + 1]].cpu().numpy()
    e32.append((np.abs(a - t64).max(), np.abs(a - t64).mean())); e6.append((np.abs(b - t64).max(), np.abs(b - t64).mean())); d.append(np.abs(a - b).max())
# ---- time at the bench size
nbt = nnet3.NnetBatch(net, [T] * U, 3); xt = torch.cat([feats_u[u % TU] for u in range(U)], 0).contiguous(); out = torch.empty((nbt.total_out_rows, net.info.output_dim), dtype=torch.float32, device=dev)
def timed(mode):
    nbt.set_precision(mode)
    for _ in range(3): nbt.forward(xt, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(5): nbt.forward(xt, out=out)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 5
t32 = timed(0); t6 = timed(1); t6p = timed(2); t32b = timed(0)
nb.set_precision(2); y6p = nb.forward(x).clone(); torch.cuda.synchronize(); nb.set_precision(0)
print(json.dumps({"utts": U, "frames_per_utt": int(T), "forward_ms_fp32_mfma": t32, "forward_ms_fp32_mfma_again": t32b, "forward_ms_split_bf16": t6, "speedup": t32 / t6, "forward_ms_split_bf16_producer_planes": t6p, "speedup_producer_planes": t32 / t6p, "tflops_fp32_equiv_producer_planes": nbt.flops / (t6p * 1e-3) / 1e12,
                  "producer_planes_output_equals_loader_split_bitwise": bool(torch.equal(y6p, y6)),
                  "tflops_fp32_equiv_split_bf16": nbt.flops / (t6 * 1e-3) / 1e12, "tflops_fp32_mfma": nbt.flops / (t32 * 1e-3) / 1e12,
                  "truth_utts": TU, "fp32_vs_f64_max_abs": float(max(e[0] for e in e32)), "fp32_vs_f64_mean_abs": float(np.mean([e[1] for e in e32])),
                  "split_bf16_vs_f64_max_abs": float(max(e[0] for e in e6)), "split_bf16_vs_f64_mean_abs": float(np.mean([e[1] for e in e6])), "split_bf16_vs_fp32_max_abs": float(max(d))}))
