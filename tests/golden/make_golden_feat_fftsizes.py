#!/usr/bin/env python3
"""Generates tests/golden/feat_fftsizes_golden.npz: the REFERENCE's compute-fbank-feats / compute-mfcc-feats on windows that pad to 256 and 1024
samples (8 kHz speech at 25 ms; 32 kHz at 25 ms; 16 kHz at 50 ms) -- the FFT sizes next to the 512 of 16 kHz / 25 ms.  Synthetic Gaussian PCM16.
Run in the BUILD container (needs oracle/_ref/bin)."""
import os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio
from tests import feat_cases as fc
BIN = os.path.join(ROOT, "oracle/_ref/bin"); ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
FLAG = {"samp_freq": "sample-frequency", "num_bins": "num-mel-bins", "frame_length_ms": "frame-length", "frame_shift_ms": "frame-shift", "num_ceps": "num-ceps", "low_freq": "low-freq", "high_freq": "high-freq",
        "use_energy": "use-energy", "snip_edges": "snip-edges", "dither": "dither", "window_type": "window-type"}
out = {}
with tempfile.TemporaryDirectory() as td:
    for name, (kind, kw, rate, nsamp, seed) in fc.FFTSIZE_CASES.items():
        wav = np.clip(np.rint(np.random.default_rng(seed).normal(0.0, 3000.0, nsamp)), -32768, 32767).astype(np.int16)
        kio.write_wav(f"{td}/{name}.wav", wav, rate=rate); open(f"{td}/{name}.scp", "w").write(f"u {td}/{name}.wav\n")
        flags = []
        for k, v in kw.items():
            if isinstance(v, int) and k in ("use_energy", "snip_edges"): v = "true" if v else "false"
            flags.append(f"--{FLAG[k]}={v}")
        subprocess.check_call([os.path.join(BIN, f"compute-{kind}-feats")] + flags + [f"scp:{td}/{name}.scp", f"ark:{td}/{name}.ark"], env=ENV, stderr=subprocess.DEVNULL)
        out["wav_" + name] = wav; out["ref_" + name] = kio.read_ark(f"{td}/{name}.ark")["u"]
np.savez_compressed(os.path.join(ROOT, "tests/golden/feat_fftsizes_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
