#!/usr/bin/env python3
"""Generates tests/golden/feat_resample_golden.npz: the REFERENCE's compute-fbank-feats / compute-mfcc-feats with --allow-downsample=true --allow-upsample=true on files whose
rate differs from --sample-frequency (ResampleWaveform, feat/resample.cc:363-372).  Synthetic Gaussian PCM16.  Run in the BUILD container (needs oracle/_ref/bin)."""
import os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio
from tests import feat_cases as fc
BIN = os.path.join(ROOT, "oracle/_ref/bin"); ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
FLAG = {"samp_freq": "sample-frequency", "num_bins": "num-mel-bins", "snip_edges": "snip-edges", "dither": "dither"}
out = {}
with tempfile.TemporaryDirectory() as td:
    for name, (kind, kw, rate, nsamp, seed) in fc.RESAMPLE_CASES.items():
        wav = np.clip(np.rint(np.random.default_rng(seed).normal(0.0, 3000.0, nsamp)), -32768, 32767).astype(np.int16)
        kio.write_wav(f"{td}/{name}.wav", wav, rate=rate); open(f"{td}/{name}.scp", "w").write(f"u {td}/{name}.wav\n")
        flags = ["--allow-downsample=true", "--allow-upsample=true"]
        for k, v in kw.items():
            if k == "snip_edges": v = "true" if v else "false"
            flags.append(f"--{FLAG[k]}={v}")
        subprocess.check_call([os.path.join(BIN, f"compute-{kind}-feats")] + flags + [f"scp:{td}/{name}.scp", f"ark:{td}/{name}.ark"], env=ENV, stderr=subprocess.DEVNULL)
        out["wav_" + name] = wav; out["ref_" + name] = kio.read_ark(f"{td}/{name}.ark")["u"]
np.savez_compressed(os.path.join(ROOT, "tests/golden/feat_resample_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
