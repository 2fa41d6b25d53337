"""End-to-end (flat-start) chain supervisions: chain::GenericNumeratorComputation (chain/chain-generic-numerator.cc) and ComputeChainObjfAndDeriv's end-to-end branch
(chain/chain-training.cc:86-215).  The reference's own sources, compiled unmodified (oracle/_ref/bin/ref-chain-objf), wrote tests/golden/chain_e2e_golden.npz
(tests/golden/make_chain_e2e_golden.py).  CPU: the float64 restatement (oracle/chain_oracle.py) against those fixtures and, where oracle/_ref exists, against the binary on random
cases.  GPU: k3_chain_supervision_create_e2e + k3_chain_objf_and_deriv / k3_chain_numerator against the fixtures and the restatement."""
import os, importlib.util, numpy as np, pytest
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "chain_e2e_golden.npz"))
_spec = importlib.util.spec_from_file_location("mk_e2e", os.path.join(HERE, "golden", "make_chain_e2e_golden.py")); mk = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(mk)

def _check(r, name, tol_rel=2e-5, tol_d=2e-5):
    for k in ("objf", "l2_term", "weight"): assert abs(r[k] - float(GOLD[f"{name}.{k}"])) <= tol_rel * max(1.0, abs(float(GOLD[f"{name}.{k}"]))), (name, k, r[k], float(GOLD[f"{name}.{k}"]))
    for k in ("deriv", "xent_deriv"): assert np.abs(r[k] - GOLD[f"{name}.{k}"]).max() <= tol_d, (name, k, np.abs(r[k] - GOLD[f"{name}.{k}"]).max())

@pytest.mark.parametrize("name", list(mk.CASES))
def test_restatement_equals_the_reference_fixture(name):
    from oracle import chain_oracle as co
    den, P, fsts, out, leaky, l2, w = mk.make(name)
    _check(co.objf_oracle_e2e(den, P, fsts, out, leaky, l2, w), name)

def test_restatement_equals_the_reference_binary_on_random_cases():
    from oracle import chain_oracle as co
    from kaldi_amd import synth
    if not co.objf_available(): pytest.skip("oracle/_ref/bin/ref-chain-objf not built")
    for seed in range(4):
        rng = np.random.default_rng(seed); P = int(rng.integers(10, 50)); B = int(rng.integers(1, 5)); T = int(rng.integers(6, 40)); w = float(rng.choice([1.0, 0.5]))
        den = synth.make_den_fst(int(rng.integers(20, 90)), P, seed=seed + 30, mean_degree=5.0, hub_degree=15)
        fsts = [synth.make_e2e_fst(T, P, seed=1000 * seed + i, num_phones=int(rng.integers(1, T // 2 + 1))) for i in range(B)]
        out = (rng.standard_normal((T * B, P)) * float(rng.choice([1.0, 4.0]))).astype(np.float32)
        r = co.ref_objf_e2e(den, P, fsts, out, 1.0e-05, 0.0, w); o = co.objf_oracle_e2e(den, P, fsts, out, 1.0e-05, 0.0, w)
        assert abs(r["objf"] - o["objf"]) <= 2e-5 * max(1.0, abs(r["objf"])) and np.abs(r["deriv"] - o["deriv"]).max() <= 2e-5 and np.abs(r["xent_deriv"] - o["xent_deriv"]).max() <= 2e-5, seed
        assert abs(float(r["xent_deriv"].sum()) - w * T * B) <= 1e-3 * T * B      # the occupation probabilities of a frame sum to one

def _gpu_objf(den, P, fsts, out, leaky, l2, w, T):
    import torch
    from kaldi_amd import chain
    g = chain.DenominatorGraph(den, P); sup = chain.Supervision(fsts, T, P, weight=w, e2e=True)
    o = torch.from_numpy(out).cuda(); d = torch.full_like(o, 7.0); x = torch.full_like(o, -3.0)      # (both derivative matrices are overwritten)
    objf, l2t, wt = chain.ComputeChainObjfAndDeriv(chain.ChainTrainingOptions(leaky_hmm_coefficient=leaky, l2_regularize=l2, out_of_range_regularize=0.0), g, sup, o, d, x)
    return dict(objf=objf, l2_term=l2t, weight=wt, deriv=d.cpu().numpy(), xent_deriv=x.cpu().numpy())

@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mk.CASES))
def test_hip_objective_and_derivatives_equal_the_reference_fixture(name):
    den, P, fsts, out, leaky, l2, w = mk.make(name); T = mk.CASES[name][7]
    _check(_gpu_objf(den, P, fsts, out, leaky, l2, w, T), name)

@pytest.mark.gpu
def test_hip_numerator_alone_and_a_training_sized_minibatch_against_the_restatement():
    import torch
    from kaldi_amd import chain, synth
    from oracle import chain_oracle as co
    P, B, T = 400, 32, 100; rng = np.random.default_rng(5)
    fsts = [synth.make_e2e_fst(T, P, seed=700 + i, num_phones=int(rng.integers(5, 40))) for i in range(B)]
    out = (rng.standard_normal((T * B, P)) * 2.0).astype(np.float32)
    sup = chain.Supervision(fsts, T, P, weight=0.5, e2e=True); num = chain.NumeratorComputation(sup, torch.from_numpy(out).cuda())
    lp, post = co.e2e_num_oracle(fsts, P, out, 0.5)
    assert abs(num.Forward() - lp) <= 2e-5 * abs(lp)                      # the log-probability as the reference returns it: without the supervision weight
    d = torch.zeros(T * B, P, device="cuda"); num.Backward(d); got = d.cpu().numpy()
    assert np.abs(got - post).max() <= 2e-5 and abs(float(got.sum()) - 0.5 * T * B) <= 1e-2

@pytest.mark.gpu
def test_e2e_fsts_that_break_the_contract_are_refused():
    from kaldi_amd import chain, synth
    f = synth.make_e2e_fst(20, 30, seed=1); g = synth.make_e2e_fst(20, 30, seed=2)
    g.final[:] = np.inf
    with pytest.raises(Exception, match="without a final state"): chain.Supervision([f, g], 20, 30, e2e=True)
    h = synth.make_e2e_fst(20, 30, seed=3); h.ilabel[0] = 0
    with pytest.raises(Exception, match="epsilon-free"): chain.Supervision([h], 20, 30, e2e=True)
