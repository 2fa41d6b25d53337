#!/usr/bin/env python3
"""Generates tests/golden/nnet_renorm.raw and nnet_renorm_io.npz.  Run in the BUILD container (needs oracle/_ref/bin).
A small TDNN of the relu-renorm kind with the remaining nonlinearities of nnet3's simple components -- NormalizeComponent (target-rms 0.5 and the default),
SigmoidComponent, TanhComponent, a LogSoftmax output layer -- created by the REFERENCE's nnet3-init (random parameters, --srand=3) and evaluated by the reference's
nnet3-compute on its CPU matrices at frame-subsampling-factor 1 and 3."""
import os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio
BIN = os.path.join(ROOT, "oracle/_ref/bin"); G = os.path.join(ROOT, "tests/golden")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
def run(*a): subprocess.check_call([os.path.join(BIN, a[0])] + list(a[1:]), env=ENV, stderr=subprocess.DEVNULL)
CONFIG = """input-node name=input dim=20
component name=a1 type=NaturalGradientAffineComponent input-dim=60 output-dim=32
component-node name=a1 component=a1 input=Append(Offset(input,-1), input, Offset(input,1))
component name=r1 type=RectifiedLinearComponent dim=32
component-node name=r1 component=r1 input=a1
component name=n1 type=NormalizeComponent dim=32 target-rms=0.5
component-node name=n1 component=n1 input=r1
component name=a2 type=NaturalGradientAffineComponent input-dim=64 output-dim=24
component-node name=a2 component=a2 input=Append(Offset(n1,-3), Offset(n1,3))
component name=s2 type=SigmoidComponent dim=24
component-node name=s2 component=s2 input=a2
component name=a3 type=NaturalGradientAffineComponent input-dim=24 output-dim=24
component-node name=a3 component=a3 input=s2
component name=t3 type=TanhComponent dim=24
component-node name=t3 component=t3 input=a3
component name=n3 type=NormalizeComponent dim=24
component-node name=n3 component=n3 input=t3
component name=a4 type=NaturalGradientAffineComponent input-dim=24 output-dim=16
component-node name=a4 component=a4 input=n3
component name=ls type=LogSoftmaxComponent dim=16
component-node name=ls component=ls input=a4
output-node name=output input=ls
"""
if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as td:
        open(f"{td}/n.config", "w").write(CONFIG)
        run("nnet3-init", "--srand=3", f"{td}/n.config", f"{G}/nnet_renorm.raw")
        rng = np.random.default_rng(21); out = {}
        feats = {"u0": (rng.standard_normal((57, 20)) * 2.0).astype(np.float32), "u1": (rng.standard_normal((9, 20)) * 30.0).astype(np.float32),      # (large inputs: saturated sigmoid / tanh)
                 "u2": np.zeros((12, 20), np.float32)}
        kio.write_ark(f"{td}/f.ark", feats)
        for k, v in feats.items(): out["feats_" + k] = v
        for s in (1, 3):
            run("nnet3-compute", "--use-gpu=no", f"--frame-subsampling-factor={s}", f"{G}/nnet_renorm.raw", f"ark:{td}/f.ark", f"ark:{td}/o.ark")
            for k, v in kio.read_ark(f"{td}/o.ark").items(): out[f"ref_s{s}_{k}"] = v
        np.savez_compressed(f"{G}/nnet_renorm_io.npz", **out)
        print({k: v.shape for k, v in out.items()}, os.path.getsize(f"{G}/nnet_renorm.raw"))
