#!/usr/bin/env python3
"""Generates tests/golden/nnet_ivector.raw and nnet_ivector_io.npz.  Run in the BUILD container (needs oracle/_ref/bin).
A small TDNN-F with the recipe's i-vector input ("input dim=N name=ivector", tdnn1 input Append(-1,0,1,ReplaceIndex(ivector, t, 0))) written by
kaldi_amd.synth, re-written by the REFERENCE's nnet3-copy, and evaluated by the reference's nnet3-compute with --online-ivectors (several
--frames-per-chunk / --online-ivector-period / --frame-subsampling-factor) and with --ivectors (one i-vector per utterance)."""
import os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio
from kaldi_amd import synth
BIN = os.path.join(ROOT, "oracle/_ref/bin"); G = os.path.join(ROOT, "tests/golden")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
def run(*a): subprocess.check_call([os.path.join(BIN, a[0])] + list(a[1:]), env=ENV, stderr=subprocess.DEVNULL)
CASES = {  # name: (subsampling, frames_per_chunk, period, utterance-level)
    "s1_c50_p10": (1, 50, 10, False), "s3_c50_p10": (3, 50, 10, False), "s3_c21_p7": (3, 21, 7, False), "s1_c20_p10_short": (1, 20, 10, False), "s3_utt": (3, 50, 0, True), "s1_utt": (1, 50, 0, True)}
with tempfile.TemporaryDirectory() as td:
    net = synth.make_tdnnf(seed=6, dim=48, bottleneck=12, strides=(1, 0, 3), prefinal_small=24, num_pdfs=96, calib_frames=300, ivector_dim=12)
    net.write(f"{td}/synth.raw")
    run("nnet3-copy", "--binary=true", f"{td}/synth.raw", f"{G}/nnet_ivector.raw")
    rng = np.random.default_rng(12); T = 131
    feats = (rng.standard_normal((T, 40)) * 1.2 + 16.5).astype(np.float32); kio.write_ark(f"{td}/f.ark", {"u": feats})
    out = {"feats": feats}
    for name, (s, chunk, period, utt) in CASES.items():
        if utt:
            iv = rng.standard_normal(12).astype(np.float32)
            open(f"{td}/iv.txt", "w").write("u  [ " + " ".join(repr(float(x)) for x in iv) + " ]\n")
            args = [f"--ivectors=ark,t:{td}/iv.txt"]; out["iv_" + name] = iv
        else:
            rows = (T + period - 1) // period - (2 if name.endswith("short") else 0)        # "short": the last chunks fall back to the last row (margin * period <= 50)
            iv = (rng.standard_normal((rows, 12)) * 0.8).astype(np.float32); kio.write_ark(f"{td}/iv.ark", {"u": iv})
            args = [f"--online-ivectors=ark:{td}/iv.ark", f"--online-ivector-period={period}"]; out["iv_" + name] = iv
        run("nnet3-compute", "--use-gpu=no", f"--frame-subsampling-factor={s}", f"--frames-per-chunk={chunk}", *args, f"{G}/nnet_ivector.raw", f"ark:{td}/f.ark", f"ark:{td}/o.ark")
        out["ref_" + name] = kio.read_ark(f"{td}/o.ark")["u"]
    np.savez_compressed(f"{G}/nnet_ivector_io.npz", **out)
    print({k: v.shape for k, v in out.items()}, os.path.getsize(f"{G}/nnet_ivector.raw"))
