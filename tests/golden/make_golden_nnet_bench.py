#!/usr/bin/env python3
"""Generates tests/golden/nnet_bench_io.npz: the REFERENCE's own nnet3-compute (oracle/_ref, built from /root/reference)
evaluated on the BENCHMARK model (17L-768/96-6024, kaldi_amd.synth.make_tdnnf(seed=1) calibrated on the oracle's fbank
features of the seed-1234 utterance -- exactly what bench.py builds) for the first 150 frames of that utterance.
The 25 MB model is not committed: synth regenerates it bit-identically (sha256 stored); every 8th output column is kept.
Run in the BUILD container (needs /root/reference)."""
import hashlib, os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio, feat_oracle as fo
from kaldi_amd import synth
BIN = os.path.join(ROOT, "oracle/_ref/bin"); G = os.path.join(ROOT, "tests/golden")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")

def bench_model_and_feats(path):
    w = synth.gaussian_pcm16(160000, 1234).astype(np.float32)
    feats = fo.compute_features(w, fo.fbank_opts(dither=0.0, num_bins=40))
    synth.make_tdnnf(seed=1, calib_feats=feats[:600]).write(path)
    return feats, hashlib.sha256(open(path, "rb").read()).hexdigest()

if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as td:
        feats, sha = bench_model_and_feats(f"{td}/m.raw")
        f = feats[:150]
        kio.write_ark(f"{td}/f.ark", {"u": f})
        subprocess.check_call([f"{BIN}/nnet3-compute", "--use-gpu=no", "--frame-subsampling-factor=3", f"{td}/m.raw", f"ark:{td}/f.ark", f"ark:{td}/o.ark"], env=ENV, stderr=subprocess.DEVNULL)
        out = kio.read_ark(f"{td}/o.ark")["u"]
        np.savez_compressed(f"{G}/nnet_bench_io.npz", feats=f, ref_out_cols8=out[:, ::8].copy(), model_sha256=np.array(sha), max_abs=np.float32(np.abs(out).max()))
        print(out.shape, "max|x|", np.abs(out).max(), sha, os.path.getsize(f"{G}/nnet_bench_io.npz"))
