#!/usr/bin/env python3
"""Oracle fixtures for the NEXT row of the scope table (online i-vector extraction, the input the stock LibriSpeech
TDNN-F needs next to the MFCCs).  Everything is made with the REFERENCE's own programs (oracle/_ref, built by oracle/build_ref.sh):
  synthetic MFCC-like features -> splice +-3 -> random LDA (20 x 91)        [numpy, the same arithmetic OnlineSpliceFrames/OnlineTransform do]
  gmm-global-init-from-feats (32 Gaussians) -> final.dubm ; gmm-global-to-fgmm ; ivector-extractor-init (dim 16) -> final.ie
  ivector-extract-online2 (= OnlineIvectorFeature: online CMVN with global stats, splice, LDA, UBM posteriors with gselect / min-post,
  statistics, conjugate-gradient solution every --ivector-period frames)   -> the golden i-vectors
Writes tests/golden/ivector/: the model files (binary + the reference's text dumps), the configuration, ivector_golden.npz with the
input features and the reference's output.  No GPU code consumes these yet; they are the oracle the row will be built against."""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio
BIN = os.path.join(ROOT, "oracle/_ref/bin"); OUT = os.path.join(ROOT, "tests/golden/ivector")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")

def run(*a):
    r = subprocess.run([os.path.join(BIN, a[0])] + list(a[1:]), capture_output=True, text=True, env=ENV)
    if r.returncode != 0: raise RuntimeError(" ".join(a) + "\n" + r.stderr[-3000:])
    return r

def text_matrix(path, m):
    with open(path, "w") as f: f.write(" [\n" + "\n".join("  " + " ".join(repr(float(x)) for x in row) for row in m) + " ]\n")

def main():
    os.makedirs(OUT, exist_ok=True); rng = np.random.default_rng(2024); D = 13
    # "speech-like": a few hidden classes with their own means, AR(1) noise around them, per-utterance offset
    classes = rng.normal(0, 6, (12, D)); feats = {}
    for u, T in enumerate([310, 187, 451, 96]):
        lab = np.repeat(rng.integers(0, 12, T // 7 + 1), 7)[:T]; x = np.zeros((T, D)); e = rng.normal(0, 1.5, (T, D))
        for t in range(1, T): x[t] = 0.7 * x[t - 1] + e[t]
        feats["utt%d" % u] = (classes[lab] + x + rng.normal(0, 2, (1, D))).astype(np.float32)
    kio.write_ark(f"{OUT}/feats.ark", feats)
    allf = np.concatenate(list(feats.values())).astype(np.float64)
    stats = np.zeros((2, D + 1)); stats[0, :D] = allf.sum(0); stats[1, :D] = (allf ** 2).sum(0); stats[0, D] = allf.shape[0]
    text_matrix(f"{OUT}/global_cmvn.stats", stats)
    lda = (rng.normal(0, 1, (20, 7 * D)) / np.sqrt(7 * D)).astype(np.float32); text_matrix(f"{OUT}/final.mat", lda)
    open(f"{OUT}/splice.conf", "w").write("--left-context=3\n--right-context=3\n")
    open(f"{OUT}/online_cmvn.conf", "w").write("# defaults of OnlineCmvnOptions\n")
    # features in the extractor's space for training the UBM: global CMN, splice with edge repetition, LDA
    mean = stats[0, :D] / stats[0, D]; train = {}
    for k, f in feats.items():
        c = f.astype(np.float64) - mean; T = c.shape[0]
        sp = np.concatenate([c[np.clip(np.arange(T) + o, 0, T - 1)] for o in range(-3, 4)], axis=1)
        train[k] = (sp @ lda.astype(np.float64).T).astype(np.float32)
    kio.write_ark(f"{OUT}/train.ark", train)
    run("gmm-global-init-from-feats", "--num-gauss=32", "--num-iters=4", "--num-frames=100000", f"ark:{OUT}/train.ark", f"{OUT}/final.dubm")
    run("gmm-global-to-fgmm", f"{OUT}/final.dubm", f"{OUT}/final.fgmm")
    run("ivector-extractor-init", "--ivector-dim=16", "--use-weights=false", f"{OUT}/final.fgmm", f"{OUT}/final.ie")
    run("gmm-global-copy", "--binary=false", f"{OUT}/final.dubm", f"{OUT}/final.dubm.txt")
    run("ivector-extractor-copy", "--binary=false", f"{OUT}/final.ie", f"{OUT}/final.ie.txt")
    conf = ["--lda-matrix=" + f"{OUT}/final.mat", "--global-cmvn-stats=" + f"{OUT}/global_cmvn.stats", "--cmvn-config=" + f"{OUT}/online_cmvn.conf", "--splice-config=" + f"{OUT}/splice.conf",
            "--diag-ubm=" + f"{OUT}/final.dubm", "--ivector-extractor=" + f"{OUT}/final.ie", "--num-gselect=5", "--min-post=0.025", "--posterior-scale=0.1", "--max-remembered-frames=1000", "--max-count=100",
            "--ivector-period=10"]
    open(f"{OUT}/ivector_extractor.conf", "w").write("\n".join(c.replace(OUT + "/", "") for c in conf) + "\n")        # relative paths: run it from this directory
    open(f"{OUT}/spk2utt", "w").write("".join(f"{k} {k}\n" for k in feats))
    out = {"feat_" + k: v for k, v in feats.items()}
    for tag, extra in (("default", []), ("greedy_off", ["--greedy-ivector-extractor=false"]), ("repeat", ["--repeat=true"])):
        r = subprocess.run([os.path.join(BIN, "ivector-extract-online2")] + conf + extra + [f"ark:{OUT}/spk2utt", f"ark:{OUT}/feats.ark", f"ark:{OUT}/iv_{tag}.ark"], capture_output=True, text=True, env=ENV)
        if r.returncode != 0: raise RuntimeError(r.stderr[-3000:])
        iv = kio.read_ark(f"{OUT}/iv_{tag}.ark")
        for k, v in iv.items(): out[f"iv_{tag}_{k}"] = v
        print(tag, {k: v.shape for k, v in iv.items()}, r.stderr.strip().splitlines()[-1][-120:])
        os.remove(f"{OUT}/iv_{tag}.ark")
    np.savez_compressed(f"{OUT}/ivector_golden.npz", **out)
    for f in ("train.ark", "final.fgmm", "feats.ark"): os.remove(f"{OUT}/{f}")
    print("wrote", sorted(os.listdir(OUT)))

if __name__ == "__main__":
    main()
