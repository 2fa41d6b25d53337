"""The LF-MMI denominator oracle (oracle/chain_oracle.py: numpy restatement of chain/chain-den-graph.cc:52-143 and chain/chain-denominator.cc:106-440)
against the REFERENCE's own code: committed outputs of oracle/_ref/bin/ref-chain-den (tests/golden/chain_den_golden.npz, generator
tests/golden/make_chain_golden.py) and, where oracle/_ref exists, the binary itself on fresh random cases."""
import importlib.util, os, numpy as np, pytest
from kaldi_amd import synth
from oracle import chain_oracle as co
HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_chain_golden", os.path.join(HERE, "golden", "make_chain_golden.py")); mg = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(mg)
GOLD = np.load(os.path.join(HERE, "golden", "chain_den_golden.npz"))

@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_oracle_equals_the_reference_fixture(name):
    f, P, out, B, leaky, dw = mg.make(name); o = co.den_oracle(f, P, out, B, leaky, dw)
    assert abs(o["objf"] - float(GOLD[name + ".objf"])) <= 2e-6 * abs(float(GOLD[name + ".objf"])) + 1e-5
    assert o["ok"] == bool(GOLD[name + ".ok"])
    assert np.abs(o["initial_probs"] - GOLD[name + ".initial_probs"]).max() <= 1e-7
    assert np.abs(o["deriv"] - GOLD[name + ".deriv"]).max() <= 2e-6

def test_derivative_is_the_gradient_of_the_objective():
    """Backward's occupation probabilities are d objf / d nnet_output (inside the [-30, 30] range): central differences on a few entries"""
    f, P, out, B, leaky, dw = mg.make("small"); o = co.den_oracle(f, P, out.astype(np.float64), B, leaky, 1.0)
    rng = np.random.default_rng(0)
    for _ in range(6):
        i, j = int(rng.integers(0, out.shape[0])), int(rng.integers(0, P)); e = 1e-2
        a = out.copy(); a[i, j] += e; b = out.copy(); b[i, j] -= e
        g = (co.den_oracle(f, P, a, B, leaky, 1.0)["objf"] - co.den_oracle(f, P, b, B, leaky, 1.0)["objf"]) / (2 * e)
        assert abs(g - o["deriv"][i, j]) <= 2e-3 * max(1.0, abs(g)), (i, j, g, o["deriv"][i, j])

def test_posteriors_sum_to_one_per_frame_and_sequence():
    f, P, out, B, leaky, dw = mg.make("leaky_large"); o = co.den_oracle(f, P, out, B, leaky, 1.0)
    assert np.abs(o["deriv"].sum(1) - 1.0).max() <= 1e-4

@pytest.mark.skipif(not co.available(), reason="oracle/_ref not built (needs /root/reference once)")
def test_oracle_equals_the_reference_binary_on_random_cases():
    rng = np.random.default_rng(11)
    for it in range(5):
        S, P = int(rng.choice([30, 200, 900])), int(rng.choice([20, 150, 700])); B, T = int(rng.integers(1, 7)), int(rng.integers(1, 25))
        f = synth.make_den_fst(S, P, seed=int(rng.integers(0, 1 << 30)), mean_degree=float(rng.choice([3.0, 10.0])), hub_degree=int(min(S, 100)))
        out = (rng.standard_normal((T * B, P)) * float(rng.choice([1.0, 3.0, 8.0]))).astype(np.float32); leaky = float(rng.choice([1e-5, 1e-2, 0.3])); dw = float(rng.choice([-1.0, 0.7]))
        r = co.ref_den(f, P, out, B, leaky, dw); o = co.den_oracle(f, P, out, B, leaky, dw)
        assert abs(o["objf"] - r["objf"]) <= 2e-6 * abs(r["objf"]) + 1e-5 and o["ok"] == r["ok"], (it, o["objf"], r["objf"])
        assert np.abs(o["initial_probs"] - r["initial_probs"]).max() <= 1e-7 and np.abs(o["deriv"] - r["deriv"]).max() <= 2e-6, it

# ---- numerator + objective ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(mg.OBJF_CASES))
def test_objective_oracle_equals_the_reference_fixture(name):
    """chain::ComputeChainObjfAndDeriv of the reference (merged supervision FST, numerator on its CPU path) vs the numpy restatement working sequence by sequence"""
    den, P, fsts, out, leaky, l2, w = mg.make_objf(name); o = co.objf_oracle(den, P, fsts, out, leaky, l2, weight=w)
    for k in ("objf", "l2_term", "weight"): assert abs(o[k] - float(GOLD[name + "." + k])) <= 2e-6 * abs(float(GOLD[name + "." + k])) + 1e-5, k
    assert np.abs(o["deriv"] - GOLD[name + ".deriv"]).max() <= 2e-6 and np.abs(o["xent_deriv"] - GOLD[name + ".xent_deriv"]).max() <= 2e-6

def test_numerator_posteriors_sum_to_the_weight_per_frame():
    den, P, fsts, out, leaky, l2, w = mg.make_objf("objf_l2_weight"); lp, post = co.num_oracle(fsts, P, out, w)
    assert np.abs(post.sum(1) - w).max() <= 1e-5 and np.isfinite(lp)

@pytest.mark.skipif(not co.objf_available(), reason="oracle/_ref not built (needs /root/reference once)")
def test_objective_oracle_equals_the_reference_binary_on_random_cases():
    rng = np.random.default_rng(31)
    for it in range(4):
        S, P = int(rng.choice([40, 300])), int(rng.choice([25, 200])); B, T = int(rng.integers(1, 6)), int(rng.integers(2, 20)); w = float(rng.choice([1.0, 0.5])); l2 = float(rng.choice([0.0, 1e-3]))
        den = synth.make_den_fst(S, P, seed=int(rng.integers(0, 1 << 30)), mean_degree=5.0, hub_degree=int(min(S, 60)))
        fsts = [synth.make_supervision_fst(T, P, seed=int(rng.integers(0, 1 << 30)), width=int(rng.choice([1, 3, 5]))) for _ in range(B)]
        out = (rng.standard_normal((T * B, P)) * float(rng.choice([1.0, 4.0]))).astype(np.float32)
        r = co.ref_objf(den, P, synth.merge_supervision_fsts(fsts), out, B, 1e-5, l2, w); o = co.objf_oracle(den, P, fsts, out, 1e-5, l2, weight=w)
        assert abs(o["objf"] - r["objf"]) <= 2e-6 * abs(r["objf"]) + 1e-5 and abs(o["l2_term"] - r["l2_term"]) <= 2e-6 * abs(r["l2_term"]) + 1e-6, it
        assert np.abs(o["deriv"] - r["deriv"]).max() <= 2e-6 and np.abs(o["xent_deriv"] - r["xent_deriv"]).max() <= 2e-6, it


def test_reference_training_iterations_are_rounding_sensitive_beyond_one_step(tmp_path):
    """Backs the multi-iteration acceptance rule of tests/test_adapter_gpu.py::test_chain_training_iterations_equal_the_reference: the reference's OWN training program
    (oracle/_ref/bin/ref-nnet3-chain-train, CPU) run under two of MKL's code paths is already 3.5e-3 of the parameter change apart from itself after ONE iteration (measured; bound here: 1e-2) and tens of percent after three
    (natural-gradient preconditioning amplifies float32 rounding discontinuously).  Informational where MKL offers only one path on the host (skip)."""
    import struct, subprocess
    ROOT = os.path.dirname(HERE); ref = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-nnet3-chain-train")
    if not os.path.exists(ref): pytest.skip("oracle/_ref not built")
    td = str(tmp_path); B, T, P, s = 8, 12, 50, 3
    calib = (np.random.default_rng(1).standard_normal((200, 40)) * 1.2 + 16.5).astype(np.float32)
    net = synth.make_tdnnf(seed=3, dim=96, bottleneck=24, strides=(1, 0, 3, 3), prefinal_small=48, num_pdfs=P, calib_feats=calib, out_std=1.5, orthonormal_constraint=-1.0); net.write(f"{td}/m.raw")
    Tin = (T - 1) * s + 1 + 16; rng = np.random.default_rng(B * 100 + T); m = np.ascontiguousarray(rng.standard_normal((Tin * B, 40)) * 1.2 + 16.5, "<f4")
    open(f"{td}/in.mat", "wb").write(b"\0BFM " + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1]) + m.tobytes())
    den = synth.make_den_fst(120, P, seed=5, mean_degree=6.0, hub_degree=60); fsts = [synth.make_supervision_fst(T, P, seed=200 + i) for i in range(B)]; merged = synth.merge_supervision_fsts(fsts)
    fb = lambda f: (np.ascontiguousarray(f.arc_offsets, np.int64).tobytes() + np.ascontiguousarray(f.ilabel, np.int32).tobytes() + np.ascontiguousarray(f.nextstate, np.int32).tobytes() +
                    np.ascontiguousarray(f.weight, np.float32).tobytes() + np.ascontiguousarray(f.final, np.float32).tobytes())
    so = np.concatenate([[0], np.cumsum([f.num_states for f in fsts])]).astype(np.int32); ab = np.concatenate([[0], np.cumsum([int(f.arc_offsets[-1]) for f in fsts])])
    with open(f"{td}/chain.spec", "wb") as fh:
        fh.write(struct.pack("<11i3f", 0x4b36, den.num_states, den.start, int(den.arc_offsets[-1]), P, B, T, merged.num_states, int(merged.arc_offsets[-1]), int(so[-1]), int(ab[-1]), 1.0e-05, 5.0e-05, 1.0))
        fh.write(fb(den)); fh.write(fb(merged)); fh.write(so.tobytes())
        fh.write(np.concatenate([[0]] + [np.asarray(f.arc_offsets[1:], np.int64) + b for f, b in zip(fsts, ab[:-1])]).astype(np.int64).tobytes())
        for k, dt in (("ilabel", np.int32), ("nextstate", np.int32), ("weight", np.float32), ("final", np.float32)): fh.write(np.concatenate([getattr(f, k) for f in fsts]).astype(dt).tobytes())
    def run(iters, path):
        env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_ENABLE_INSTRUCTIONS=path)
        r = subprocess.run([ref, f"{td}/m.raw", str(s), f"{td}/in.mat", f"{td}/chain.spec", str(iters), "0.002", "0.0", f"{td}/o.raw", f"{td}/o.vec"], capture_output=True, text=True, env=env); assert r.returncode == 0, r.stderr[-1500:]
        b = open(f"{td}/o.vec", "rb").read(); n = struct.unpack("<i", b[6:10])[0]; return np.frombuffer(b, "<f4", n, 10)[3 * iters:].copy()
    p0 = np.concatenate([np.concatenate([c[2]["W"].ravel()] + ([c[2]["b"].ravel()] if "b" in c[2] and c[2]["b"].size else [])) for c in net.components if c[1] in ("affine", "tdnn", "linear")])
    a1, b1 = run(1, "AVX2"), run(1, "AVX512")
    if np.array_equal(a1, b1): pytest.skip("MKL runs the same code path under both settings on this host")
    assert np.linalg.norm(a1 - b1) <= 1e-2 * np.linalg.norm(a1 - p0)
    a3, b3 = run(3, "AVX2"), run(3, "AVX512")
    if not np.linalg.norm(a3 - b3) > 1e-2 * np.linalg.norm(a3 - p0): pytest.skip("on this host the two MKL paths stay together over three iterations (measured elsewhere: 35 % apart)")
