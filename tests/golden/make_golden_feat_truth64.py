#!/usr/bin/env python3
"""Generates tests/golden/feat_truth64.npz: the feature cases of tests/feat_cases.REF_CASES evaluated in FLOAT64 (oracle/feat_oracle.c compiled with
-Dfloat=double and the libm double functions: the reference's formulas, no float32 rounding anywhere).  It measures how far the REFERENCE's own
float32 binaries are from the exact value of what they compute (tests/test_feat_gpu.py, tests/test_oracle_feat.py): on the 40-cepstra lifted
MFCCs the reference itself is 1.2e-4 .. 3.7e-4 away."""
import ctypes, os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle import feat_oracle as fo
from tests import feat_cases as fc
g = np.load(os.path.join(ROOT, "tests", "golden", "feat_golden.npz"))
with tempfile.TemporaryDirectory() as td:
    so = os.path.join(td, "libfeat64.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-shared", "-fPIC", "-Dfloat=double", "-Dcosf=cos", "-Dsinf=sin", "-Dlogf=log", "-Dexpf=exp", "-Dsqrtf=sqrt", "-Dpowf=pow", "-Dfabsf=fabs",
                           "-Dfmaxf=fmax", "-Dfminf=fmin", "-o", so, os.path.join(ROOT, "oracle", "feat_oracle.c"), "-lm"])
    L = ctypes.CDLL(so); L.k3o_num_frames.restype = ctypes.c_int32; L.k3o_feat_dim.restype = ctypes.c_int32
    class Opts64(ctypes.Structure): _fields_ = [(n, ctypes.c_double if t is ctypes.c_float else t) for n, t in fo.FeatOpts._fields_]
    out = {}
    for name in sorted(fc.REF_CASES):
        kind, kw, wkey = fc.REF_CASES[name]
        o32 = fo.mfcc_opts(**kw) if kind == "mfcc" else fo.fbank_opts(**kw); o = Opts64(*[getattr(o32, n) for n, _ in fo.FeatOpts._fields_])
        w = np.ascontiguousarray(g[wkey], np.float64); T = L.k3o_num_frames(ctypes.c_int64(w.size), ctypes.byref(o)); D = L.k3o_feat_dim(ctypes.byref(o))
        t = np.zeros((T, D), np.float64); L.k3o_compute_features(ctypes.byref(o), w.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(w.size), t.ctypes.data_as(ctypes.c_void_p))
        out["truth64_" + name] = t
        print(name, "reference vs float64: %.2e" % np.abs(g["ref_" + name] - t).max())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "feat_truth64.npz"), **out)
