#!/usr/bin/env python3
"""Would the network's products survive being computed as split-bf16 MFMAs?  (CPU study, numpy: no GPU involved.)

The TDNN-F forward is the second consumer of a benchmark step (30 ms next to the decoder's 75 ms) and runs on FP32 MFMA because of the 1e-4 bound of the path.  gfx950's bf16 MFMA
rate is 16x the fp32 rate, so a product written as a few bf16 products of split operands (x = x_hi + x_lo (+ x_lo2), each part a bf16; products of bf16 pairs are exact in fp32) could
be faster IF the bound survives:  x3 = hi*hi + hi*lo + lo*hi  (3 MFMAs),  x6 = the six terms down to 2^-24 (6 MFMAs).  This script evaluates the benchmark model (17 layers,
768 / 96 / 6024) with the products replaced by their split forms (accumulation in float64, so only the operand rounding is measured) and reports the distance to the float64 forward
next to the float32 oracle's own distance.  Result recorded in DESIGN.md section 5."""
import os, re, sys, tempfile, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)

def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)

def split(x, parts):
    out = []; r = np.asarray(x, np.float32).astype(np.float64)
    for _ in range(parts):
        p = bf16(r.astype(np.float32)); out.append(p.astype(np.float64)); r = r - p
    return out

def make_mm(mode):
    def mm(a, b):
        if mode == "f64": return np.asarray(a, np.float64) @ np.asarray(b, np.float64)
        if mode == "f32": return (np.asarray(a, np.float32) @ np.asarray(b, np.float32))
        if mode == "bf16x1": (a0,), (b0,) = split(a, 1), split(b, 1); return (a0 @ b0).astype(np.float32)
        if mode == "bf16x3": a0, a1 = split(a, 2); b0, b1 = split(b, 2); return (a0 @ b0 + a0 @ b1 + a1 @ b0).astype(np.float32)
        if mode == "bf16x6": a0, a1, a2 = split(a, 3); b0, b1, b2 = split(b, 3); return (a0 @ b0 + a0 @ b1 + a1 @ b0 + a0 @ b2 + a1 @ b1 + a2 @ b0).astype(np.float32)
        raise ValueError(mode)
    return mm

def oracle_with(mm, dtype):
    src = open(os.path.join(ROOT, "oracle", "nnet3_oracle.py")).read()
    src = src.replace('(x @ f["<LinearParams>"].T + f["<BiasParams>"])', '(_mm(x, f["<LinearParams>"].T) + f["<BiasParams>"])').replace('(x @ f["<Params>"].T)', '(_mm(x, f["<Params>"].T))')
    src = src.replace('y += x[s:s + n] @ W[:, i * D:(i + 1) * D].T', 'y += _mm(x[s:s + n], W[:, i * D:(i + 1) * D].T)')
    assert src.count("_mm(") == 3
    m = types.ModuleType("nnet3_oracle_split"); m._mm = mm; exec(compile(src, "nnet3_oracle_split", "exec"), m.__dict__); return m

if __name__ == "__main__":
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_golden_nnet_bench.py")); mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    with tempfile.TemporaryDirectory() as td:
        feats, _ = mk.bench_model_and_feats(f"{td}/m.raw"); feats = feats[:int(sys.argv[1]) if len(sys.argv) > 1 else 90]
        ref = None
        for mode in ("f64", "f32", "bf16x6", "bf16x3", "bf16x1"):
            o = oracle_with(make_mm(mode), np.float64 if mode == "f64" else np.float32); net = o.read_nnet(f"{td}/m.raw")
            out = o.compute(net, feats, 3, dtype=np.float64) if mode == "f64" else o.compute(net, feats, 3)
            if ref is None: ref = out.astype(np.float64); print(f"output: {out.shape}, max |x| = {np.abs(ref).max():.3f}"); continue
            d = np.abs(out.astype(np.float64) - ref)
            print(f"{mode:7s} max |y - y_f64| = {d.max():.3g}   mean = {d.mean():.3g}   (bound of the path: 1e-4)")
