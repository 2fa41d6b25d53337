#!/usr/bin/env python3
"""Generates tests/golden/ivector/ivector_weighted_golden.npz: the REFERENCE's ivector-extract-online2 --frame-weights-rspecifier (silence weighting of the i-vector
statistics, online2bin/ivector-extract-online2.cc:130-153 -> OnlineIvectorFeature::UpdateFrameWeights / UpdateStatsUntilFrameWeighted) on the committed extractor and
features, two utterances per speaker.  The weights: utt0 runs of 0/1 (a silence detector's output), utt1 fractional weights incl. some below min_post/0.99 (the pruning
threshold saturates at 0.99) and a few negative ones, utt2 a vector two frames SHORT (--length-tolerance=2: the missing frames weigh 0), utt3 all ones.
Run in the BUILD container (needs oracle/_ref/bin)."""
import os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio
DIR = os.path.join(ROOT, "tests/golden/ivector"); EXE = os.path.join(ROOT, "oracle/_ref/bin/ivector-extract-online2")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
g = np.load(os.path.join(DIR, "ivector_golden.npz")); utts = ["utt0", "utt1", "utt2", "utt3"]
rng = np.random.default_rng(77); T = {u: g["feat_" + u].shape[0] for u in utts}
w = {}
w["utt0"] = np.repeat(rng.integers(0, 2, (T["utt0"] + 6) // 7), 7)[:T["utt0"]].astype(np.float32)
w["utt1"] = rng.uniform(0.0, 1.0, T["utt1"]).astype(np.float32); w["utt1"][::9] = 0.01; w["utt1"][4::23] = -0.2; w["utt1"][5::31] = 0.0
w["utt2"] = (rng.uniform(0, 1, T["utt2"] - 2) > 0.3).astype(np.float32)
w["utt3"] = np.ones(T["utt3"], np.float32)

def write_weights(path, d):
    with open(path, "w") as f:
        for k, v in d.items(): f.write(k + "  [ " + " ".join(repr(float(x)) for x in v) + " ]\n")

if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as td:
        kio.write_ark(f"{td}/feats.ark", {u: g["feat_" + u] for u in utts}); write_weights(f"{td}/w.txt", w)
        open(f"{td}/spk2utt", "w").write("spkA utt0 utt1\nspkB utt2 utt3\n")
        out = {"w_" + u: w[u] for u in utts}
        for tag, extra in (("iv", []), ("ivrep", ["--repeat=true"])):
            subprocess.check_call([EXE, "--config=ivector_extractor.conf", "--length-tolerance=2", f"--frame-weights-rspecifier=ark,t:{td}/w.txt"] + extra +
                                  [f"ark:{td}/spk2utt", f"ark:{td}/feats.ark", f"ark:{td}/{tag}.ark"], env=ENV, cwd=DIR, stderr=subprocess.DEVNULL)
            iv = kio.read_ark(f"{td}/{tag}.ark")
            for u in utts: out[f"{tag}_{u}"] = iv[u] if tag == "iv" else iv[u][::10]
        # too short for the default tolerance: utt2 is an error, utt3 starts from fresh statistics
        r = subprocess.run([EXE, "--config=ivector_extractor.conf", f"--frame-weights-rspecifier=ark,t:{td}/w.txt", f"ark:{td}/spk2utt", f"ark:{td}/feats.ark", f"ark:{td}/tol0.ark"], env=ENV, cwd=DIR, capture_output=True, text=True)
        t0 = kio.read_ark(f"{td}/tol0.ark"); assert sorted(t0) == ["utt0", "utt1", "utt3"], sorted(t0); out["tol0_utt3"] = t0["utt3"]
        print(r.stderr.strip().splitlines()[-2][-100:])
    ref = np.load(os.path.join(DIR, "ivector_adapt_golden.npz"))
    np.savez_compressed(os.path.join(DIR, "ivector_weighted_golden.npz"), **out)
    print({u: (out["iv_" + u].shape, float(np.abs(out["iv_" + u] - ref["iv_" + u]).max())) for u in utts})
