#!/usr/bin/env python3
"""Generates tests/golden/ivector/ivector_adapt_golden.npz: the REFERENCE's ivector-extract-online2 on the committed extractor and features
(tests/golden/ivector) with TWO utterances per speaker (spkA = utt0 utt1, spkB = utt2 utt3), i.e. with the adaptation state (CMVN statistics and
i-vector statistics) carried from a speaker's first utterance to the second (online2bin/ivector-extract-online2.cc:100-178).
Run in the BUILD container (needs oracle/_ref/bin)."""
import os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio
DIR = os.path.join(ROOT, "tests/golden/ivector"); EXE = os.path.join(ROOT, "oracle/_ref/bin/ivector-extract-online2")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
g = np.load(os.path.join(DIR, "ivector_golden.npz")); utts = ["utt0", "utt1", "utt2", "utt3"]
with tempfile.TemporaryDirectory() as td:
    kio.write_ark(f"{td}/feats.ark", {u: g["feat_" + u] for u in utts})
    open(f"{td}/spk2utt", "w").write("spkA utt0 utt1\nspkB utt2 utt3\n")
    subprocess.check_call([EXE, "--config=ivector_extractor.conf", f"ark:{td}/spk2utt", f"ark:{td}/feats.ark", f"ark:{td}/iv.ark"], env=ENV, cwd=DIR, stderr=subprocess.DEVNULL)
    iv = kio.read_ark(f"{td}/iv.ark")
    # --repeat=true: one row per frame, and the statistics handed on hold EVERY frame of the first utterance (GetFrame(T - 1), :121-127)
    subprocess.check_call([EXE, "--config=ivector_extractor.conf", "--repeat=true", f"ark:{td}/spk2utt", f"ark:{td}/feats.ark", f"ark:{td}/ivr.ark"], env=ENV, cwd=DIR, stderr=subprocess.DEVNULL)
    ivr = kio.read_ark(f"{td}/ivr.ark")
np.savez_compressed(os.path.join(DIR, "ivector_adapt_golden.npz"), **{"iv_" + u: iv[u] for u in utts}, **{"ivrep_" + u: ivr[u][::10] for u in utts})      # (every 10th row = one per ivector period)
print({u: (iv[u].shape, float(np.abs(iv[u] - g["iv_default_" + u]).max())) for u in utts})
