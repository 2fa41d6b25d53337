#!/usr/bin/env python3
"""Generates tests/golden/feat_golden.npz.  Run in the BUILD container (needs /root/reference and
oracle/_ref built by oracle/build_ref.sh); the fixture it writes is what travels to the GPU box.

Contents:
  wav            int16 samples of the reference's src/feat/test_data/test.wav (16 kHz mono)
  htk_fbank_{1..4}, htk_mfcc_{1..6}
                 the reference's HTK golden vectors (src/feat/test_data/test.wav.fbank_htk.N,
                 test.wav.fea_htk.N), parsed to float32 matrices (data files, not source)
  ref_<name>     outputs of the reference's own compute-fbank-feats / compute-mfcc-feats /
                 compute-cmvn-stats|apply-cmvn binaries for the command lines in REF_RUNS
  syn_wav        synthetic Gaussian PCM16 (seed 1234, sigma 3000), 1.37 s -- the SURVEY 8d recipe
"""
import os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kaldi_io as kio
TD = "/root/reference/src/feat/test_data"
BIN = os.path.join(ROOT, "oracle/_ref/bin")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle/_ref/mkl"), MKL_THREADING_LAYER="SEQUENTIAL")

REF_RUNS = {  # name -> (binary, flags, which wav)
    "fbank_default40": ("compute-fbank-feats", ["--dither=0", "--num-mel-bins=40"], "wav"),
    "fbank_default23": ("compute-fbank-feats", ["--dither=0"], "wav"),
    "fbank_energy_nosnip": ("compute-fbank-feats", ["--dither=0", "--num-mel-bins=40", "--use-energy=true", "--snip-edges=false"], "wav"),
    "fbank_hamming_nopow": ("compute-fbank-feats", ["--dither=0", "--window-type=hamming", "--use-power=false", "--remove-dc-offset=false", "--raw-energy=false", "--use-energy=true"], "wav"),
    "mfcc_default": ("compute-mfcc-feats", ["--dither=0"], "wav"),
    "mfcc_hires": ("compute-mfcc-feats", ["--dither=0", "--num-mel-bins=40", "--num-ceps=40", "--low-freq=20", "--high-freq=-400", "--use-energy=false"], "wav"),
    "mfcc_htkcompat": ("compute-mfcc-feats", ["--dither=0", "--htk-compat=true", "--use-energy=false", "--snip-edges=false"], "wav"),
    "fbank_syn40": ("compute-fbank-feats", ["--dither=0", "--num-mel-bins=40"], "syn_wav"),
    "mfcc_syn_hires": ("compute-mfcc-feats", ["--dither=0", "--num-mel-bins=40", "--num-ceps=40", "--low-freq=20", "--high-freq=-400", "--use-energy=false"], "syn_wav"),
}

def main():
    out = {}
    wav, rate = kio.read_wav(os.path.join(TD, "test.wav")); assert rate == 16000
    out["wav"] = wav
    rng = np.random.default_rng(1234)
    out["syn_wav"] = np.clip(np.rint(rng.normal(0.0, 3000.0, 21920)), -32768, 32767).astype(np.int16)
    for i in range(1, 5): out[f"htk_fbank_{i}"] = kio.read_htk(os.path.join(TD, f"test.wav.fbank_htk.{i}"))
    for i in range(1, 7): out[f"htk_mfcc_{i}"] = kio.read_htk(os.path.join(TD, f"test.wav.fea_htk.{i}"))
    with tempfile.TemporaryDirectory() as td:
        for wname in ("wav", "syn_wav"):
            kio.write_wav(os.path.join(td, wname + ".wav"), out[wname])
            open(os.path.join(td, wname + ".scp"), "w").write(f"u {td}/{wname}.wav\n")
        for name, (binary, flags, wname) in REF_RUNS.items():
            ark = os.path.join(td, name + ".ark")
            subprocess.check_call([os.path.join(BIN, binary)] + flags + [f"scp:{td}/{wname}.scp", f"ark:{ark}"], env=ENV, stderr=subprocess.DEVNULL)
            out["ref_" + name] = kio.read_ark(ark)["u"]
        # per-utterance CMVN through the reference's compute-cmvn-stats | apply-cmvn
        kio.write_ark(os.path.join(td, "f.ark"), {"u": out["ref_fbank_default40"]})
        subprocess.check_call([os.path.join(BIN, "compute-cmvn-stats"), f"ark:{td}/f.ark", f"ark:{td}/c.ark"], env=ENV, stderr=subprocess.DEVNULL)
        for nv in (0, 1):
            subprocess.check_call([os.path.join(BIN, "apply-cmvn"), f"--norm-vars={'true' if nv else 'false'}", f"ark:{td}/c.ark", f"ark:{td}/f.ark", f"ark:{td}/o{nv}.ark"], env=ENV, stderr=subprocess.DEVNULL)
            out[f"ref_cmvn_normvars{nv}"] = kio.read_ark(f"{td}/o{nv}.ark")["u"]
    np.savez_compressed(os.path.join(ROOT, "tests/golden/feat_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})

if __name__ == "__main__":
    main()
