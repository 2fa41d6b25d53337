"""CPU tests of the restated LatticeFasterDecoder oracle (oracle/lattice_faster_oracle.cc).  The reference holds no decoder
tests or fixtures (decoder/Makefile:6), so the oracle is checked against hand-computed cases that restate the reference's
float32 arithmetic (lattice-faster-decoder.cc:789-793, 174-181, 339-341) and against invariants of the algorithm."""
import numpy as np, pytest
from kaldi_amd.fst import Fst
from kaldi_amd import synth
from oracle import lattice_oracle as lo
F = np.float32
INF = np.inf

def _fst(n, start, arcs, finals):
    """arcs: (src, ilabel, olabel, weight, dst)"""
    final = np.full(n, np.inf, np.float32)
    for s, c in finals.items(): final[s] = c
    a = np.array(arcs, dtype=object).reshape(-1, 5)
    return Fst.from_arcs(n, start, a[:, 0].astype(np.int64), a[:, 1].astype(np.int32), a[:, 2].astype(np.int32), a[:, 3].astype(np.float32), a[:, 4].astype(np.int32), final)

def _arcs(lat):
    k = lat.keys()
    return sorted((int(lat.st_frame[s]), int(lat.st_state[s]), int(lat.st_frame[d]), int(lat.st_state[d]), int(i), int(o), float(g), float(a))
                  for s, d, i, o, g, a in zip(lat.arc_src, lat.arc_dst, lat.arc_ilabel, lat.arc_olabel, lat.arc_graph, lat.arc_ac))

T2P = np.array([0, 0, 1, 2], np.int32)      # tid 1,2,3 -> pdf 0,1,2

@pytest.mark.parametrize("mode", [0, 1])
def test_linear_chain_costs_follow_the_reference_arithmetic(mode):
    f = _fst(3, 0, [(0, 1, 7, 0.5, 1), (1, 2, 0, 0.25, 2)], {2: 0.125})
    ll = np.array([[-1.5, -9.0, 0.0], [-7.0, -2.25, 0.0]], np.float32)
    lat, info = lo.decode(f, ll, T2P, lo.Config(beam=10.0, lattice_beam=5.0), mode)
    # frame 0: best token cost 0 -> cost_offset = -0; ac = cost_offset - loglike; tot = cur + ac + graph
    co0 = F(-0.0); ac0 = F(co0 - F(-1.5)); c1 = F(F(F(0) + ac0) + F(0.5))
    co1 = F(-c1); ac1 = F(co1 - F(-2.25)); c2 = F(F(c1 + ac1) + F(0.25))
    assert info["reached_final"] and lat.num_states == 3
    assert _arcs(lat) == [(0, 0, 1, 1, 1, 7, 0.5, float(F(ac0 - co0))), (1, 1, 2, 2, 2, 0, 0.25, float(F(ac1 - co1)))]
    assert np.array_equal(np.sort(lat.st_cost), np.sort(np.array([0.0, c1, c2], np.float32)))
    fin = lat.st_final[np.isfinite(lat.st_final)]
    assert fin.tolist() == [0.125]
    il, ol, g, ac = lat.best_path()
    assert il == [1, 2] and ol == [7] and abs(g - 0.875) < 1e-6 and abs(ac - 3.75) < 1e-5

@pytest.mark.parametrize("mode", [0, 1])
def test_epsilon_closure_min_cost_and_both_links_kept(mode):
    # 0 -eps(0.5)-> 1, 0 -eps(0.25)-> 2 -eps(0.125)-> 1 : token at state 1 gets cost 0.375, BOTH links into it stay (:886-887)
    # 1 -tid1(1.0)-> 3 final
    f = _fst(4, 0, [(0, 0, 0, 0.5, 1), (0, 0, 5, 0.25, 2), (2, 0, 0, 0.125, 1), (1, 1, 0, 1.0, 3)], {3: 0.0})
    ll = np.zeros((1, 3), np.float32)
    lat, _ = lo.decode(f, ll, T2P, lo.Config(beam=10.0, lattice_beam=5.0), mode)
    a = _arcs(lat)
    assert (0, 0, 0, 1, 0, 0, 0.5, 0.0) in a and (0, 0, 0, 2, 0, 5, 0.25, 0.0) in a and (0, 2, 0, 1, 0, 0, 0.125, 0.0) in a
    k = {(int(fr), int(st)): float(c) for fr, st, c in zip(lat.st_frame, lat.st_state, lat.st_cost)}
    assert k[(0, 1)] == 0.375 and k[(1, 3)] == 1.375

@pytest.mark.parametrize("mode", [0, 1])
def test_lattice_beam_prunes_the_worse_branch_only_beyond_the_beam(mode):
    # two parallel 2-frame paths 0->1->3 and 0->2->3; the second is worse by `gap` on graph cost
    for gap, kept in ((1.0, True), (3.0, False)):
        f = _fst(4, 0, [(0, 1, 0, 0.5, 1), (0, 1, 0, 0.5 + gap, 2), (1, 1, 0, 0.5, 3), (2, 1, 0, 0.5, 3)], {3: 0.0})
        lat, _ = lo.decode(f, np.zeros((2, 3), np.float32), T2P, lo.Config(beam=10.0, lattice_beam=2.0), mode)
        states = {(int(a), int(b)) for a, b in zip(lat.st_frame, lat.st_state)}
        assert ((1, 2) in states) == kept, (gap, states)
        assert (1, 1) in states and (2, 3) in states

@pytest.mark.parametrize("mode", [0, 1])
def test_no_final_state_reached_all_last_tokens_final_with_weight_one(mode):
    f = _fst(3, 0, [(0, 1, 0, 0.5, 1), (1, 1, 0, 0.5, 1)], {2: 0.0})          # state 2 is never reached
    lat, info = lo.decode(f, np.zeros((3, 3), np.float32), T2P, lo.Config(beam=10.0, lattice_beam=5.0), mode)
    assert not info["reached_final"]
    last = lat.st_frame == 3
    assert last.sum() == 1 and lat.st_final[last].tolist() == [0.0]            # LatticeWeight::One()

def test_beam_prunes_against_the_frame_best():
    # from the start state: arcs with cost 0 and cost 9; beam 5 -> the second token must not exist on frame 1
    f = _fst(3, 0, [(0, 1, 0, 0.0, 1), (0, 2, 0, 9.0, 2), (1, 1, 0, 0.0, 1), (2, 1, 0, 0.0, 2)], {1: 0.0, 2: 0.0})
    cfg = lo.Config(beam=5.0, lattice_beam=20.0, min_active=0)
    for mode in (0, 1):
        lat, _ = lo.decode(f, np.zeros((2, 3), np.float32), T2P, cfg, mode)
        assert {(int(a), int(b)) for a, b in zip(lat.st_frame, lat.st_state)} == {(0, 0), (1, 1), (2, 1)}

def _rand_case(seed, S=1500, A=4000, N=40, T=50):
    f = synth.make_hclg(S, A, N, seed=seed, start_degree=30)
    rng = np.random.default_rng(seed + 1)
    return f, synth.tid2pdf(N), (rng.standard_normal((T, N)) * 2.5).astype(np.float32)

def _is_sub(small, big):
    ss, sa = small.canonical(); bs, ba = big.canonical()
    S = set(map(tuple, bs.tolist())); A = set(map(tuple, ba.tolist()))
    return all(tuple(r) in S for r in ss.tolist()) and all(tuple(r) in A for r in sa.tolist())

@pytest.mark.parametrize("seed", [1, 2, 3])
def test_two_pass_mode_is_the_order_independent_core_of_the_literal_mode(seed):
    """mode 1 (beam applied against the FINAL next_cutoff) gives a sub-lattice of mode 0 (serial tightening) with bit-identical
    costs and the same best path; mode 0's surplus arcs exist only because next_cutoff was still loose when they were visited."""
    f, t2p, ll = _rand_case(seed)
    cfg = lo.Config(beam=15.0, lattice_beam=8.0, max_active=10000)
    l0, i0 = lo.decode(f, ll, t2p, cfg, 0); l1, i1 = lo.decode(f, ll, t2p, cfg, 1)
    assert i1["extra_links"] == 0 and i0["extra_links"] > 0
    assert _is_sub(l1.connect(), l0.connect())
    b0, b1 = l0.connect().best_path(), l1.connect().best_path()
    assert b0[0] == b1[0] and b0[1] == b1[1] and abs(b0[2] - b1[2]) < 1e-3 and abs(b0[3] - b1[3]) < 1e-3

def test_literal_mode_depends_on_the_hash_size_two_pass_mode_does_not():
    """hash_ratio is a pure memory/speed knob of the reference (lattice-faster-decoder.h:56,90), yet it changes the HashList
    iteration order and with it the literal algorithm's arc set; the two-pass definition is invariant."""
    differs = 0
    for seed in (1, 2, 3, 4):
        f, t2p, ll = _rand_case(seed, S=40000, A=100000, N=60, T=40)      # states >> hash size, so state % hash_size matters
        a = [lo.decode(f, ll, t2p, lo.Config(beam=15.0, lattice_beam=8.0, hash_ratio=hr), 0)[0] for hr in (2.0, 3.7)]
        b = [lo.decode(f, ll, t2p, lo.Config(beam=15.0, lattice_beam=8.0, hash_ratio=hr), 1)[0] for hr in (2.0, 3.7)]
        differs += a[0].diff(a[1]) != ""
        assert b[0].diff(b[1]) == ""
    assert differs > 0

@pytest.mark.parametrize("mode", [0, 1])
def test_periodic_pruning_does_not_change_the_final_lattice(mode):
    """PruneActiveTokens every prune_interval frames only frees memory: same lattice with it disabled --
    the GPU decoder prunes once, after the last frame."""
    f, t2p, ll = _rand_case(5, T=70)
    a = lo.decode(f, ll, t2p, lo.Config(beam=15.0, lattice_beam=8.0, prune_interval=25), mode)[0]
    b = lo.decode(f, ll, t2p, lo.Config(beam=15.0, lattice_beam=8.0, prune_interval=10**9), mode)[0]
    c = lo.decode(f, ll, t2p, lo.Config(beam=15.0, lattice_beam=8.0, prune_interval=3), mode)[0]
    assert a.diff(b) == "" and a.diff(c) == ""

def test_max_active_and_min_active_cutoffs():
    f, t2p, ll = _rand_case(6, T=30)
    _, wide = lo.decode(f, ll, t2p, lo.Config(beam=15.0, lattice_beam=8.0), 1)
    _, capped = lo.decode(f, ll, t2p, lo.Config(beam=15.0, lattice_beam=8.0, max_active=100, min_active=10), 1)
    assert wide["ntoks"].max() > 300
    assert (capped["adaptive_beam"][5:] < 15.0).any()              # max_active tighter than the beam on some frame
    _, loose = lo.decode(f, ll, t2p, lo.Config(beam=1.0, lattice_beam=0.5, min_active=200), 1)
    assert (loose["adaptive_beam"] > 1.0).any()                    # min_active looser than the beam


# ---- the restated decoder against the REFERENCE's own decoder source ------------------------------------------------------------
# oracle/_ref/bin/ref-lattice-decoder is /root/reference/src/decoder/lattice-faster-decoder.cc compiled unmodified against a stand-in
# for the part of OpenFst it touches (third_party/minifst, oracle/build_ref.sh).  The oracle's literal mode (mode 0) must
# reproduce its GetRawLattice output exactly: same states per frame, same arcs, same labels, same float bits, same sharing of states
# -- compared up to state renaming (tests/lattice_sig.py), for all limits / beams / prune intervals / hash sizes of decoder_cases.
import json, os
from tests import decoder_cases as dcases, lattice_sig as lsig
_GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decoder_ref_golden.json")))

@pytest.mark.parametrize("name", sorted(dcases.CASES))
def test_literal_mode_equals_the_reference_decoder(name):
    f, t2p, ll, kw = dcases.make(name)
    lat, info = lo.decode(f, ll, t2p, lo.Config(**kw), 0)
    g = _GOLD[name]
    assert (lat.num_states, lat.num_arcs, info["reached_final"]) == (g["states"], g["arcs"], g["reached_final"])
    canon = lsig.canonical_of_raw(lat)
    assert lsig.digest(canon) == g["digest"]                 # recorded from the reference binary (tests/golden/make_golden_decoder.py)
    from oracle import ref_decoder as rd
    if rd.available():                                       # and live, where oracle/_ref is present
        ref = rd.decode(f, ll, t2p, lo.Config(**kw))
        assert lsig.canonical_of_reference(ref) == canon
        assert ref["num_frames"] == ll.shape[0]

def test_reference_decoder_edge_cases_match_too():
    """no token survives / final state not reached / a single frame: whatever the reference does, the oracle does"""
    from oracle import ref_decoder as rd
    if not rd.available(): pytest.skip("oracle/_ref not built (needs /root/reference)")
    f = _fst(4, 0, [(0, 1, 7, 0.5, 1), (1, 2, 0, 0.25, 2), (1, 3, 9, 4.0, 3), (2, 0, 0, 0.0, 3)], {3: 0.125})
    for T, cfg in ((1, lo.Config(beam=10.0, lattice_beam=5.0)), (2, lo.Config(beam=10.0, lattice_beam=5.0)), (2, lo.Config(beam=10.0, lattice_beam=0.5)), (3, lo.Config(beam=1.0, lattice_beam=1.0))):
        ll = (np.random.default_rng(T).standard_normal((T, 3)) * 2).astype(np.float32)
        ref = rd.decode(f, ll, T2P, cfg); lat, info = lo.decode(f, ll, T2P, cfg, 0)
        assert ref["reached_final"] == info["reached_final"]
        assert lsig.canonical_of_reference(ref) == lsig.canonical_of_raw(lat), (T, ref["frame"].size, lat.num_states)

def test_random_configurations_against_the_reference_decoder():
    """fuzz (live only): random graph sizes, utterance lengths, score spreads and every LatticeFasterDecoderConfig field; the literal mode
    equals the reference decoder's raw lattice each time (a 40-configuration run of this loop: 40 identical)"""
    from oracle import ref_decoder as rd
    if not rd.available(): pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(778)
    for it in range(8):
        N = int(rng.choice([20, 40, 80])); S = int(rng.choice([300, 1500, 6000])); A = int(S * rng.uniform(2.0, 3.5)); T = int(rng.integers(1, 70))
        f = synth.make_hclg(S, A, N, seed=int(rng.integers(0, 1 << 30)), start_degree=int(rng.choice([5, 30, 200])))
        ll = (rng.standard_normal((T, N)) * float(rng.choice([1.0, 2.5, 5.0]))).astype(np.float32)
        kw = dict(beam=float(rng.choice([4.0, 8.0, 15.0, 20.0])), lattice_beam=float(rng.choice([1.0, 4.0, 8.0, 12.0])), beam_delta=float(rng.choice([0.5, 0.1, 2.0])),
                  hash_ratio=float(rng.choice([2.0, 1.0, 3.7])), prune_interval=int(rng.choice([25, 1, 7, 1000])), prune_scale=float(rng.choice([0.1, 0.5])))
        if rng.random() < 0.4: kw["max_active"] = int(rng.choice([50, 200, 1000]))
        if rng.random() < 0.4: kw["min_active"] = int(rng.choice([0, 20, 500]))
        if "max_active" in kw and kw.get("min_active", 200) >= kw["max_active"]: kw["min_active"] = max(0, kw["max_active"] - 1)
        cfg = lo.Config(**kw)
        ref = rd.decode(f, ll, synth.tid2pdf(N), cfg); lat, info = lo.decode(f, ll, synth.tid2pdf(N), cfg, 0)
        assert ref["reached_final"] == info["reached_final"] and lsig.canonical_of_reference(ref) == lsig.canonical_of_raw(lat), (it, kw, ref["frame"].size, lat.num_states)


# ---- the literal algorithm in the phase structure of the GPU kernel (mode 2) and with every parallel phase shuffled (mode 3) --------
def _same_run(f, ll, t2p, cfg, modes=(2, 3, 4)):
    l0, i0 = lo.decode(f, ll, t2p, cfg, 0)
    for m in modes:
        l, i = lo.decode(f, ll, t2p, cfg, m)
        for k in ("ntoks", "cur_cutoff", "adaptive_beam", "next_cutoff", "cost_offset"):
            assert np.array_equal(np.asarray(i0[k]).view(np.int32), np.asarray(i[k]).view(np.int32)), (m, k)
        assert l.diff(l0) == "" and i["order_sensitive_events"] == i0["order_sensitive_events"] and i["reached_final"] == i0["reached_final"], m
    return i0

@pytest.mark.parametrize("name", sorted(n for n in dcases.CASES if n != "bench_config"))
def test_phase_parallel_literal_mode_equals_the_serial_literal_mode(name):
    """HashList visit order from (bucket first-occupation, insertion) ranks, acceptance by exclusive prefix-min over the arc sequence,
    creation times by min over accepted arc numbers, eps closure = order-free fixpoint + replay of the LIFO queue for the creation
    ORDER only: bit-identical lattices, per-frame cutoffs and order-sensitive-event counts to the serial algorithm (which the
    reference-decoder tests above pin to the reference's own source).  This is the algorithm k3_decoder_config.literal_order runs."""
    f, t2p, ll, kw = dcases.make(name)
    i0 = _same_run(f, ll, t2p, lo.Config(**kw))
    if name != "min_active_loosens": assert i0["order_sensitive_events"] > 0

def test_phase_parallel_literal_mode_random_configurations():
    """fuzz incl. cost grids that force exact ties (best-token ties, equal-cost arcs): 10 here, an 80-configuration run found none"""
    rng = np.random.default_rng(5)
    for it in range(10):
        N = int(rng.choice([20, 40, 80])); S = int(rng.choice([300, 1500, 6000])); A = int(S * rng.uniform(2.0, 3.5)); T = int(rng.integers(1, 50))
        f = synth.make_hclg(S, A, N, seed=int(rng.integers(0, 1 << 30)), start_degree=int(rng.choice([5, 30, 200])))
        ll = (rng.standard_normal((T, N)) * float(rng.choice([1.0, 2.5, 5.0]))).astype(np.float32)
        if it % 3 == 0: ll = np.round(ll * 2) / 2; f.weight[:] = np.round(f.weight * 4) / 4
        kw = dict(beam=float(rng.choice([4.0, 8.0, 15.0])), lattice_beam=float(rng.choice([1.0, 4.0, 8.0])), beam_delta=float(rng.choice([0.5, 0.1])), hash_ratio=float(rng.choice([2.0, 1.0, 3.7])))
        if rng.random() < 0.5: kw["max_active"] = int(rng.choice([50, 200, 1000]))
        if rng.random() < 0.5: kw["min_active"] = int(rng.choice([0, 20]))
        if "max_active" in kw and kw.get("min_active", 200) >= kw["max_active"]: kw["min_active"] = max(0, kw["max_active"] - 1)      # LatticeFasterDecoderConfig::Check(): min_active <= max_active
        _same_run(f, ll, synth.tid2pdf(N), lo.Config(**kw))
