#!/usr/bin/env python3
"""Generates tests/golden/nnet_small.{txt,raw} and nnet_small_io.npz.  Run in the BUILD container.
The model is a small TDNN-F (dims 48/12, strides 1,0,3) written by kaldi_amd.synth in Kaldi binary format,
then RE-WRITTEN by the reference's own nnet3-copy in text and binary (so the fixtures are files produced by
reference code), and evaluated by the reference's nnet3-compute (--frame-subsampling-factor 1 and 3)."""
import os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio
from kaldi_amd import synth
BIN = os.path.join(ROOT, "oracle/_ref/bin"); G = os.path.join(ROOT, "tests/golden")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
def run(*a): subprocess.check_call([os.path.join(BIN, a[0])] + list(a[1:]), env=ENV, stderr=subprocess.DEVNULL)
with tempfile.TemporaryDirectory() as td:
    net = synth.make_tdnnf(seed=5, dim=48, bottleneck=12, strides=(1, 0, 3), prefinal_small=24, num_pdfs=96, calib_frames=300)
    net.write(f"{td}/synth.raw")
    run("nnet3-copy", "--binary=false", f"{td}/synth.raw", f"{G}/nnet_small.txt")
    run("nnet3-copy", "--binary=true", f"{td}/synth.raw", f"{G}/nnet_small.raw")
    rng = np.random.default_rng(11)
    feats = (rng.standard_normal((83, 40)) * 1.2 + 16.5).astype(np.float32)
    kio.write_ark(f"{td}/f.ark", {"u": feats})
    out = {"feats": feats}
    for fmt in ("raw", "txt"):       # the text dump keeps ~7 digits, so it is a (slightly) different model
        for s in (1, 3):
            run("nnet3-compute", "--use-gpu=no", f"--frame-subsampling-factor={s}", f"{G}/nnet_small.{fmt}", f"ark:{td}/f.ark", f"ark:{td}/o.ark")
            out[f"ref_out_{fmt}_s{s}"] = kio.read_ark(f"{td}/o.ark")["u"]
    np.savez_compressed(f"{G}/nnet_small_io.npz", **out)
    print({k: v.shape for k, v in out.items()}, os.path.getsize(f"{G}/nnet_small.txt"), os.path.getsize(f"{G}/nnet_small.raw"))
