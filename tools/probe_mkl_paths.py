#!/usr/bin/env python3
"""Which environment settings make the reference's nnet3-compute (MKL) take a different float32 code path on THIS host?  (bench.py's
e2e_parity.reference_vs_itself needs one.)  Prints max |delta log-like| of every variant against the default run, one 10 s utterance."""
import os, subprocess, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kaldi_amd import synth
from oracle import kaldi_io as kio
b = os.path.join(ROOT, "oracle", "_ref", "bin"); env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
with tempfile.TemporaryDirectory() as td:
    kio.write_wav(f"{td}/u.wav", synth.gaussian_pcm16(160000, 1234)); open(f"{td}/wav.scp", "w").write(f"u {td}/u.wav\n")
    subprocess.check_call([f"{b}/compute-fbank-feats", "--dither=0", "--num-mel-bins=40", f"scp:{td}/wav.scp", f"ark:{td}/f.ark"], env=env, stderr=subprocess.DEVNULL)
    synth.make_tdnnf(seed=1, calib_feats=kio.read_ark(f"{td}/f.ark")["u"][:600]).write(f"{td}/m.raw")
    def run(extra, chunk=150):
        subprocess.check_call([f"{b}/nnet3-compute", "--use-gpu=no", "--frame-subsampling-factor=3", f"--frames-per-chunk={chunk}", f"{td}/m.raw", f"ark:{td}/f.ark", f"ark:{td}/o.ark"], env=dict(env, **extra), stderr=subprocess.DEVNULL)
        return kio.read_ark(f"{td}/o.ark")["u"]
    base = run({})
    for extra in ({"MKL_ENABLE_INSTRUCTIONS": "SSE4_2"}, {"MKL_ENABLE_INSTRUCTIONS": "AVX2"}, {"MKL_ENABLE_INSTRUCTIONS": "AVX512"}, {"MKL_CBWR": "COMPATIBLE"}, {"MKL_CBWR": "SSE4_2"}, {"MKL_CBWR": "AVX2"},
                  {"MKL_CBWR": "AVX512"}, {"MKL_DEBUG_CPU_TYPE": "5"}, {"MKL_VERBOSE": "0"}):
        try: print(extra, float(np.abs(run(extra) - base).max()))
        except Exception as e: print(extra, "failed", repr(e)[:100])
    print("frames-per-chunk 50", float(np.abs(run({}, 50) - base).max()))
    print("frames-per-chunk 51", float(np.abs(run({}, 51) - base).max()))
