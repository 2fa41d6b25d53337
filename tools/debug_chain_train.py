#!/usr/bin/env python3
"""Writes a small TDNN-F (orthonormal-constrained bottlenecks), a minibatch and a chain spec into a directory and runs both builds of kaldi_amd/adapter/nnet3-chain-train.cc on it
(oracle on the CPU where oracle/_ref exists; the adapter build on the GPU box).  With K3_ADAPTER_LIST_MISSING=1 the adapter build lists the CuMatrix members it reached without an
implementation instead of stopping at the first.   tools/debug_chain_train.py <dir> [iters]"""
import os, sys, struct, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import synth
td = sys.argv[1]; iters = sys.argv[2] if len(sys.argv) > 2 else "3"; os.makedirs(td, exist_ok=True)
B, T, P, s = 8, 12, 50, 3
BIG = bool(os.environ.get("K3_TRAIN_BIG"))      # the benchmark model (17L-768/96-6024) and a training-sized minibatch
if BIG: B, T, P = 64, 50, 6024
def kaldi_matrix(path, m):
    m = np.ascontiguousarray(m, "<f4"); open(path, "wb").write(b"\0BFM " + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1]) + m.tobytes())
if not os.path.exists(f"{td}/chain.spec"):
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    (synth.make_tdnnf(seed=1, calib_feats=calib, orthonormal_constraint=-1.0) if BIG else synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=P, calib_feats=calib, out_std=1.5, orthonormal_constraint=-1.0)).write(f"{td}/m.raw")
    lc = rc = (40 if BIG else 8);      # 1 (tdnn1) + the sum of the TDNN-F time strides
    Tin = (T - 1) * s + 1 + lc + rc; rng = np.random.default_rng(B * 100 + T)
    kaldi_matrix(f"{td}/in.mat", rng.standard_normal((Tin * B, 40)) * 1.2 + 16.5)
    den = synth.make_den_fst(3000, P) if BIG else synth.make_den_fst(120, P, seed=5, mean_degree=6.0, hub_degree=60); fsts = [synth.make_supervision_fst(T, P, seed=200 + i) for i in range(B)]; merged = synth.merge_supervision_fsts(fsts)
    fb = lambda f: (np.ascontiguousarray(f.arc_offsets, np.int64).tobytes() + np.ascontiguousarray(f.ilabel, np.int32).tobytes() + np.ascontiguousarray(f.nextstate, np.int32).tobytes() +
                    np.ascontiguousarray(f.weight, np.float32).tobytes() + np.ascontiguousarray(f.final, np.float32).tobytes())
    so = np.concatenate([[0], np.cumsum([f.num_states for f in fsts])]).astype(np.int32); ab = np.concatenate([[0], np.cumsum([int(f.arc_offsets[-1]) for f in fsts])])
    with open(f"{td}/chain.spec", "wb") as fh:
        fh.write(struct.pack("<11i3f", 0x4b36, den.num_states, den.start, int(den.arc_offsets[-1]), P, B, T, merged.num_states, int(merged.arc_offsets[-1]), int(so[-1]), int(ab[-1]), 1.0e-05, 5.0e-05, 1.0))
        fh.write(fb(den)); fh.write(fb(merged)); fh.write(so.tobytes())
        fh.write(np.concatenate([[0]] + [np.asarray(f.arc_offsets[1:], np.int64) + b for f, b in zip(fsts, ab[:-1])]).astype(np.int64).tobytes())
        for k, dt in (("ilabel", np.int32), ("nextstate", np.int32), ("weight", np.float32), ("final", np.float32)): fh.write(np.concatenate([getattr(f, k) for f in fsts]).astype(dt).tobytes())
env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
ref = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-nnet3-chain-train"); exe = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-chain-train")
args = [f"{td}/m.raw", str(s), f"{td}/in.mat", f"{td}/chain.spec", iters, "0.002", "0.0"]
if os.path.exists(ref) and os.environ.get("RUN_REF", "1") == "1":
    r = subprocess.run([ref] + args + [f"{td}/ref.raw", f"{td}/ref.objf"], capture_output=True, text=True, env=dict(env, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"))); open(f"{td}/ref.log", "w").write(r.stderr); print("ref rc", r.returncode, r.stderr[-1500:])
try:
    import torch; gpu = torch.cuda.is_available()
except Exception: gpu = False
if gpu:
    g = subprocess.run([exe] + args + [f"{td}/gpu.raw", f"{td}/gpu.objf"], capture_output=True, text=True, env=env); open(f"{td}/gpu.log", "w").write(g.stderr); print("gpu rc", g.returncode, g.stderr[-4000:])
    import re; its = [float(x) for x in re.findall(r"frames; ([0-9.]+) ms", g.stderr)]; print("iteration ms:", its)
