"""GPU parity of the LF-MMI denominator (kaldi_amd/csrc/k3_chain.hip through the C ABI k3_chain_den_*): the HIP forward-backward against the
REFERENCE's DenominatorComputation -- its committed outputs (tests/golden/chain_den_golden.npz), the reference binary itself where oracle/_ref
travels with the snapshot, and the numpy oracle.  Bar: objective within 1e-4 relative (it is a sum of ~frames x sequences logs), initial
probabilities within 1e-7, derivatives (posteriors scaled by deriv_weight, all in [-1, 1]) within 1e-5 absolute; the reference's own CPU and GPU
paths differ by as much (different summation orders of float products)."""
import importlib.util, os, time, numpy as np, pytest, torch
from kaldi_amd import synth
from oracle import chain_oracle as co
pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_chain_golden", os.path.join(HERE, "golden", "make_chain_golden.py")); mg = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(mg)
GOLD = np.load(os.path.join(HERE, "golden", "chain_den_golden.npz"))

def _run(f, P, out, B, leaky, dw, deriv_init=None):
    from kaldi_amd import chain
    g = chain.DenominatorGraph(f, P); o = torch.from_numpy(out).cuda()
    comp = chain.DenominatorComputation(chain.ChainTrainingOptions(leaky), g, B, o)
    objf_fwd = comp.Forward()
    d = torch.zeros_like(o) if deriv_init is None else torch.from_numpy(deriv_init).cuda()
    ok = comp.Backward(dw, d); torch.cuda.synchronize()
    return dict(objf=comp._objf, objf_forward_only=objf_fwd, ok=ok, initial_probs=g.InitialProbs(), deriv=d.cpu().numpy())

def _close(h, r):
    assert abs(h["objf"] - r["objf"]) <= 1e-4 * abs(r["objf"]) + 1e-4, (h["objf"], r["objf"])
    assert h["objf_forward_only"] == h["objf"] and h["ok"] == r["ok"]
    assert np.abs(h["initial_probs"] - r["initial_probs"]).max() <= 1e-7
    assert np.abs(h["deriv"] - r["deriv"]).max() <= 1e-5, np.abs(h["deriv"] - r["deriv"]).max()

@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_hip_equals_the_reference_fixture(name):
    f, P, out, B, leaky, dw = mg.make(name); h = _run(f, P, out, B, leaky, dw)
    _close(h, dict(objf=float(GOLD[name + ".objf"]), ok=bool(GOLD[name + ".ok"]), initial_probs=GOLD[name + ".initial_probs"], deriv=GOLD[name + ".deriv"]))

def test_hip_adds_into_the_derivative_matrix():
    f, P, out, B, leaky, dw = mg.make("small"); base = np.random.default_rng(2).standard_normal(out.shape).astype(np.float32)
    h = _run(f, P, out, B, leaky, dw, deriv_init=base)
    assert np.abs(h["deriv"] - (base + GOLD["small.deriv"])).max() <= 1e-5

def test_hip_equals_the_reference_on_a_training_sized_minibatch():
    """3000-state / 51 k-transition graph with hub states (wavefront-walked), 4000 pdfs, 64 sequences x 50 frames: the reference binary live when
    oracle/_ref is there, else the numpy oracle (pinned to it by tests/test_oracle_chain.py)"""
    f = synth.make_den_fst(3000, 4000); P, B, T = 4000, 64, 50
    out = (np.random.default_rng(5).standard_normal((T * B, P)) * 2.0).astype(np.float32)
    r = co.ref_den(f, P, out, B, 1.0e-05, -1.0) if co.available() else co.den_oracle(f, P, out, B, 1.0e-05, -1.0)
    h = _run(f, P, out, B, 1.0e-05, -1.0)
    _close(h, r)
    assert np.abs(h["deriv"].reshape(T, B, P).sum(2) + 1.0).max() <= 1e-3      # posteriors of a frame sum to one (deriv_weight = -1)
    # timing of the one launch (forward + backward) for the record
    from kaldi_amd import chain
    g = chain.DenominatorGraph(f, P); o = torch.from_numpy(out).cuda(); d = torch.zeros_like(o); comp = chain.DenominatorComputation(chain.ChainTrainingOptions(1.0e-05), g, B, o)
    comp.Backward(-1.0, d); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): comp.Backward(-1.0, d)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"\nchain denominator forward-backward: {B} x {T} frames, {f.num_states} states / {int(f.arc_offsets[-1])} transitions, {P} pdfs: {ms:.3f} ms per minibatch ({B * T / ms * 1e3:.0f} frames/s)")

def test_random_cases_against_the_oracle():
    rng = np.random.default_rng(21)
    for it in range(6):
        S, P = int(rng.choice([30, 200, 900, 2500])), int(rng.choice([20, 150, 700, 3000])); B, T = int(rng.integers(1, 40)), int(rng.integers(1, 30))
        f = synth.make_den_fst(S, P, seed=int(rng.integers(0, 1 << 30)), mean_degree=float(rng.choice([3.0, 10.0])), hub_degree=int(min(S, 200)))
        out = (rng.standard_normal((T * B, P)) * float(rng.choice([1.0, 3.0, 8.0]))).astype(np.float32); leaky = float(rng.choice([1e-5, 1e-2, 0.3])); dw = float(rng.choice([-1.0, 0.7]))
        _close(_run(f, P, out, B, leaky, dw), co.den_oracle(f, P, out, B, leaky, dw))

def test_graphs_beyond_the_lds_budget_are_refused_loudly():
    from kaldi_amd import chain, lib
    f = synth.make_den_fst(9000, 12000, mean_degree=3.0)
    g = chain.DenominatorGraph(f, 12000); o = torch.zeros((4, 12000), device="cuda")
    with pytest.raises(lib.K3Error): chain.DenominatorComputation(chain.ChainTrainingOptions(1e-5), g, 2, o).Forward()

# ---- numerator + objective ------------------------------------------------------------------------------------------------------------------
def _objf(den, P, fsts, out, leaky, l2, w, xent=True, oor=0.0, apply_oor=False, derivs=True):
    from kaldi_amd import chain
    g = chain.DenominatorGraph(den, P); T = out.shape[0] // len(fsts); sup = chain.Supervision(fsts, T, P, w); o = torch.from_numpy(out).cuda()
    d = torch.full_like(o, 7.0) if derivs else None; x = torch.full_like(o, 7.0) if (xent and derivs) else None      # (both are overwritten)
    objf, l2t, wt = chain.ComputeChainObjfAndDeriv(chain.ChainTrainingOptions(leaky, l2, oor, apply_oor), g, sup, o, d, x); torch.cuda.synchronize()
    return dict(objf=objf, l2_term=l2t, weight=wt, deriv=d.cpu().numpy() if d is not None else None, xent_deriv=x.cpu().numpy() if x is not None else None)

def _close_objf(h, r, tol_d=1e-5):
    assert abs(h["objf"] - r["objf"]) <= 1e-4 * abs(r["objf"]) + 1e-4, (h["objf"], r["objf"])
    assert abs(h["l2_term"] - r["l2_term"]) <= 1e-5 * abs(r["l2_term"]) + 1e-6 and abs(h["weight"] - r["weight"]) <= 1e-5 * r["weight"]
    if h["deriv"] is not None: assert np.abs(h["deriv"] - r["deriv"]).max() <= tol_d, np.abs(h["deriv"] - r["deriv"]).max()
    if h["xent_deriv"] is not None: assert np.abs(h["xent_deriv"] - r["xent_deriv"]).max() <= tol_d

@pytest.mark.parametrize("name", sorted(mg.OBJF_CASES))
def test_objective_and_derivatives_equal_the_reference_fixture(name):
    den, P, fsts, out, leaky, l2, w = mg.make_objf(name); r = {k: GOLD[name + "." + k] for k in ("objf", "l2_term", "weight", "deriv", "xent_deriv")}
    r = {k: (float(v) if v.ndim == 0 else v) for k, v in r.items()}
    _close_objf(_objf(den, P, fsts, out, leaky, l2, w), r)
    h = _objf(den, P, fsts, out, leaky, l2, w, xent=False); assert h["xent_deriv"] is None; _close_objf(h, r)           # numerator straight into the derivative
    h = _objf(den, P, fsts, out, leaky, l2, w, derivs=False); assert abs(h["objf"] - r["objf"]) <= 1e-4 * abs(r["objf"]) + 1e-4      # objective only

def test_numerator_class_and_out_of_range_penalty_against_the_oracle():
    from kaldi_amd import chain
    den, P, fsts, out, leaky, l2, w = mg.make_objf("objf_l2_weight"); out = out * 6.0                                 # some outputs beyond +-30
    T = out.shape[0] // len(fsts); sup = chain.Supervision(fsts, T, P, w); o = torch.from_numpy(out).cuda(); d = torch.zeros_like(o)
    num = chain.NumeratorComputation(sup, o); lp = num.Forward(); num.Backward(d); torch.cuda.synchronize()
    olp, opost = co.num_oracle(fsts, P, out, w)
    assert abs(lp - olp) <= 1e-5 * abs(olp) + 1e-4 and np.abs(d.cpu().numpy() - opost).max() <= 1e-5
    r = co.objf_oracle(den, P, fsts, out, leaky, l2, out_of_range_regularize=0.01, apply_out_of_range_penalty=True, weight=w)
    _close_objf(_objf(den, P, fsts, out, leaky, l2, w, oor=0.01, apply_oor=True), r, tol_d=2e-5)

def test_objective_on_a_training_sized_minibatch():
    """128 sequences x 50 frames, 3000-state denominator graph, 4000 pdfs: against the reference binary when oracle/_ref travels (merged supervision FST), else the oracle"""
    P, B, T = 4000, 128, 50; den = synth.make_den_fst(3000, P); rng = np.random.default_rng(9)
    fsts = [synth.make_supervision_fst(T, P, seed=1000 + i, width=4) for i in range(B)]
    out = (rng.standard_normal((T * B, P)) * 2.0).astype(np.float32)
    r = co.ref_objf(den, P, synth.merge_supervision_fsts(fsts), out, B, 1e-5, 5e-5, 1.0) if co.objf_available() else co.objf_oracle(den, P, fsts, out, 1e-5, 5e-5)
    _close_objf(_objf(den, P, fsts, out, 1e-5, 5e-5, 1.0), r)
    from kaldi_amd import chain
    g = chain.DenominatorGraph(den, P); sup = chain.Supervision(fsts, T, P, 1.0); o = torch.from_numpy(out).cuda(); d = torch.zeros_like(o); x = torch.zeros_like(o); opts = chain.ChainTrainingOptions(1e-5, 5e-5)
    chain.ComputeChainObjfAndDeriv(opts, g, sup, o, d, x); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): chain.ComputeChainObjfAndDeriv(opts, g, sup, o, d, x)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"\nLF-MMI objective + derivatives (denominator, numerator, l2, xent derivative): {B} x {T} frames, {P} pdfs: {ms:.3f} ms per minibatch")

def test_supervision_fsts_that_break_the_contract_are_refused():
    from kaldi_amd import chain, lib
    f = synth.make_supervision_fst(6, 20, seed=1)
    with pytest.raises(lib.K3Error): chain.Supervision([f], 7, 20)                          # paths are 6 arcs long, not 7
    g = synth.make_supervision_fst(6, 20, seed=1); g.ilabel[0] = 0
    with pytest.raises(lib.K3Error): chain.Supervision([g], 6, 20)                          # epsilon label
