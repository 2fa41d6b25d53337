"""Compile the oracle's C/C++ restatements (gcc/g++, CPU only) and, when /root/reference is
present, the reference's own binaries into oracle/_ref (oracle/build_ref.sh).  Test infra only."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))

TARGETS = [
    ("libk3oracle_feat.so", ["feat_oracle.c"], "gcc", ["-O2", "-std=gnu11"]),      # includes feat_oracle_path.inc twice (float32 / float64 data path), see DEPS
    ("libk3oracle_nnet.so", ["nnet_oracle.c"], "gcc", ["-O3", "-std=gnu11", "-march=native"]),
    ("libk3oracle_dec.so", ["lattice_faster_oracle.cc"], "g++", ["-O2", "-std=c++17", "-ffp-contract=off", "-Wall"]),
]

DEPS = {"libk3oracle_feat.so": ["feat_oracle_path.inc"]}      # included files: rebuild when they change

def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)

def build(verbose=False, with_ref=True):
    for out, srcs, cc, flags in TARGETS:
        srcs_abs = [os.path.join(HERE, s) for s in srcs]
        if not all(os.path.exists(s) for s in srcs_abs):
            continue
        out_abs = os.path.join(HERE, out)
        if _stale(out_abs, srcs_abs + [os.path.join(HERE, d) for d in DEPS.get(out, [])]):
            cmd = [cc] + flags + ["-shared", "-fPIC", "-o", out_abs] + srcs_abs + ["-lm"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    if with_ref and os.path.isdir(os.environ.get("KALDI_REFERENCE", "/root/reference") + "/src"):
        if not os.path.exists(os.path.join(HERE, "_ref", "bin", "ref-word-align")):      # the newest of the reference tools
            subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")])

if __name__ == "__main__":
    build(verbose=True)
