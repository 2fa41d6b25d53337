#!/usr/bin/env python3
"""Generates tests/golden/cmvn_online_golden.npz from the reference's own online2bin/apply-cmvn-online (built into oracle/_ref by
oracle/build_ref.sh).  Run in the BUILD container; the fixture travels to the GPU box.

Contents: feats_a [730 x 13], feats_b [411 x 13] synthetic features (seed 77; AR(1) in time so that window means move), feats_c = the
142 x 40 fbank of the reference's test.wav; global [2 x 14] / global40 [2 x 41] stats; for every case in CASES the output matrices."""
import os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio
BIN = os.path.join(ROOT, "oracle/_ref/bin")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")

CASES = {   # name -> (flags, feature set, spk2utt?)
    "default": ([], "ab", False),
    "w100": (["--cmn-window=100", "--speaker-frames=60", "--global-frames=25"], "ab", False),
    "w100_vars": (["--cmn-window=100", "--speaker-frames=60", "--global-frames=25", "--norm-vars=true"], "ab", False),
    "w100_spk": (["--cmn-window=100", "--speaker-frames=60", "--global-frames=25"], "ab", True),
    "w100_spk_vars_skip": (["--cmn-window=100", "--speaker-frames=100", "--global-frames=10", "--norm-vars=true", "--skip-dims=0:5"], "ab", True),
    "nomeans": (["--norm-means=false"], "ab", False),
    "fbank40": (["--cmn-window=50", "--speaker-frames=50", "--global-frames=20"], "c", False),
}

def text_stats(path, st):
    with open(path, "w") as f:
        f.write(" [\n" + "\n".join("  " + " ".join(repr(float(x)) for x in row) for row in st) + " ]\n")

def acc_stats(mats):
    dim = mats[0].shape[1]; st = np.zeros((2, dim + 1))
    for m in mats:
        m = m.astype(np.float64); st[0, :dim] += m.sum(0); st[1, :dim] += (m * m).sum(0); st[0, dim] += m.shape[0]
    return st

def main():
    rng = np.random.default_rng(77); out = {}
    def ar(T, dim):
        x = np.zeros((T, dim)); e = rng.normal(0, 1, (T, dim)); drift = np.cumsum(rng.normal(0, 0.05, (T, dim)), 0)
        for t in range(1, T): x[t] = 0.9 * x[t - 1] + e[t]
        return (3.0 * x + drift * 4 + rng.normal(0, 5, (1, dim))).astype(np.float32)
    out["feats_a"] = ar(730, 13); out["feats_b"] = ar(411, 13)
    out["feats_c"] = np.load(os.path.join(ROOT, "tests/golden/feat_golden.npz"))["ref_fbank_default40"]
    out["global"] = acc_stats([ar(2000, 13)]); out["global40"] = acc_stats([out["feats_c"] * 0.9 + 0.3])
    with tempfile.TemporaryDirectory() as td:
        text_stats(f"{td}/g13.txt", out["global"]); text_stats(f"{td}/g40.txt", out["global40"])
        kio.write_ark(f"{td}/ab.ark", {"utt_a": out["feats_a"], "utt_b": out["feats_b"]}); kio.write_ark(f"{td}/c.ark", {"utt_c": out["feats_c"]})
        open(f"{td}/spk2utt", "w").write("spk1 utt_a utt_b\n")
        for name, (flags, fs, spk) in CASES.items():
            cmd = [os.path.join(BIN, "apply-cmvn-online")] + flags + ([f"--spk2utt=ark:{td}/spk2utt"] if spk else [])
            cmd += [f"{td}/g13.txt" if fs == "ab" else f"{td}/g40.txt", f"ark:{td}/{fs}.ark", f"ark:{td}/o.ark"]
            subprocess.check_call(cmd, env=ENV, stderr=subprocess.DEVNULL)
            for k, v in kio.read_ark(f"{td}/o.ark").items(): out[f"ref_{name}_{k}"] = v
    np.savez_compressed(os.path.join(ROOT, "tests/golden/cmvn_online_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})

if __name__ == "__main__":
    main()
