"""Host-side word-level lattice determinization (kaldi_amd/host/k3_lattice.cc, the lattice-determinize-pruned program): CPU only.
Two kinds of checks: (1) at the end of the file, character-for-character equality with the REFERENCE's own determinizer source
(compiled unmodified against the OpenFst stand-in of third_party/minifst); (2) the defining properties the reference's own
determinize-lattice-pruned-test.cc checks through RandEquivalent, verified by exhaustive path enumeration on small random lattices:
  * the output is deterministic on word labels and has no epsilon arcs,
  * every word sequence whose best raw path is within the beam is present,
  * each word sequence present appears once, with the (graph, acoustic) cost and the transition-id string of its best raw path.
Tolerance: 2e-3 on path costs (float32 sums of up to ~30 arc costs below 10 accumulated in a different association order)."""
import os, subprocess, numpy as np, pytest
from tests import lattice_cases as lc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROG = os.path.join(ROOT, "kaldi_amd", "bin", "lattice-determinize-pruned")
TOL = 2e-3


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()
    assert os.path.exists(PROG)


def _run(args, inp, binary_out=False):
    r = subprocess.run([PROG] + args + ["ark:-", "ark:-" if binary_out else "ark,t:-"], input=inp, capture_output=True, timeout=120)
    return r


def _check(lat, out, beam, acoustic_scale=1.0, expect_complete=True):
    raw = lc.enumerate_raw(lat, acoustic_scale)
    best_raw = {w: min(v) for w, v in raw.items()}
    best = min(c[0] for c in best_raw.values())
    # deterministic, epsilon-free
    seen = set()
    for (s, d, w, g, a, t) in out["arcs"]:
        assert w != 0
        assert (s, w) not in seen, "two arcs with the same word leave state %d" % s
        seen.add((s, w))
    det = lc.enumerate_compact(out, acoustic_scale)
    for w, paths in det.items():
        assert len(paths) == 1
        assert w in best_raw, "word sequence %r is not in the raw lattice" % (w,)
        cost, g, a, tids = paths[0]; rc, rg, ra, rt = best_raw[w]
        assert abs(cost - rc) <= TOL and abs(g - rg) <= TOL and abs(a - ra) <= TOL, (w, paths[0], best_raw[w])
        runner_up = sorted(v[0] for v in raw[w])[1] if len(raw[w]) > 1 else np.inf
        if runner_up - rc > 2 * TOL: assert tids == rt, (w, tids, rt)
        else: assert any(tids == v[3] and abs(v[0] - rc) <= 2 * TOL for v in raw[w])
    if expect_complete:
        for w, (rc, _, _, _) in best_raw.items():
            if rc <= best + beam - TOL: assert w in det, "word sequence %r (cost %.4f, best %.4f, beam %g) was lost" % (w, rc, best, beam)
        assert any(abs(best_raw[w][0] - best) <= TOL for w in det)        # the best path survives
    return raw, det


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("beam", [0.5, 3.0, 1000.0])
def test_determinize_properties(seed, beam):
    lat = lc.random_lattice(seed, frames=5 + seed % 3, width=3 + seed % 2, words=2 + seed % 3)
    r = _run(["--beam=%g" % beam], lc.lattice_text("utt%d" % seed, lat).encode())
    assert r.returncode == 0, r.stderr.decode()
    out = lc.parse_compact_text(r.stdout.decode())["utt%d" % seed]
    raw, det = _check(lat, out, beam)
    if beam >= 1000.0: assert set(det) == set(raw)


def test_exact_ties_pick_one_alignment():
    """costs on a 1/4 grid: many exactly equal path costs; the result must still be deterministic and optimal, and the alignment
    chosen must be one of the cheapest (ties broken on graph cost, then the string order of CompactLatticeWeight)."""
    for seed in range(6):
        lat = lc.random_lattice(100 + seed, frames=6, width=3, words=2, tids=4, quant=4)
        r = _run(["--beam=50"], lc.lattice_text("u", lat).encode())
        assert r.returncode == 0, r.stderr.decode()
        _check(lat, lc.parse_compact_text(r.stdout.decode())["u"], 50.0)


def test_acoustic_scale_changes_ranking_and_is_undone_on_output():
    lat = lc.random_lattice(7, frames=6, width=4, words=3)
    outs = {}
    for sc in (1.0, 0.1):
        r = _run(["--beam=2.0", "--acoustic-scale=%g" % sc], lc.lattice_text("u", lat).encode())
        assert r.returncode == 0, r.stderr.decode()
        outs[sc] = lc.parse_compact_text(r.stdout.decode())["u"]
        _check(lat, outs[sc], 2.0, acoustic_scale=sc)
    assert outs[1.0] != outs[0.1]


def test_binary_io_matches_text_io():
    lats = [lc.random_lattice(40 + i, frames=5, width=3) for i in range(4)]
    text_in = "".join(lc.lattice_text("k%d" % i, l) for i, l in enumerate(lats)).encode()
    bin_in = b"".join(lc.lattice_binary("k%d" % i, l) for i, l in enumerate(lats))
    a = _run(["--beam=4"], text_in); b = _run(["--beam=4"], bin_in); c = _run(["--beam=4"], bin_in, binary_out=True)
    assert a.returncode == 0 and b.returncode == 0 and c.returncode == 0, (a.stderr, b.stderr, c.stderr)
    assert a.stdout == b.stdout
    ta = lc.parse_compact_text(a.stdout.decode()); tc = lc.parse_compact_binary(c.stdout)
    assert list(ta) == list(tc) == ["k0", "k1", "k2", "k3"]
    for k in ta:
        assert ta[k]["start"] == tc[k]["start"] and len(ta[k]["arcs"]) == len(tc[k]["arcs"])
        for x, y in zip(ta[k]["arcs"], tc[k]["arcs"]):
            assert x[:3] == y[:3] and x[5] == y[5] and np.allclose(x[3:5], y[3:5], rtol=1e-5, atol=1e-6)      # %g text keeps 6 digits
        assert set(ta[k]["finals"]) == set(tc[k]["finals"])


def test_output_is_topologically_sorted_with_start_zero():
    for seed in range(5):
        lat = lc.random_lattice(60 + seed, frames=7, width=4, words=3)
        out = lc.parse_compact_text(_run(["--beam=6"], lc.lattice_text("u", lat).encode()).stdout.decode())["u"]
        assert out["start"] == 0
        assert all(d > s for (s, d, *_) in out["arcs"])


def test_limits_stop_early_but_keep_a_valid_lattice():
    lat = lc.random_lattice(3, frames=8, width=4, words=4, p_word=0.6)
    full = lc.parse_compact_text(_run(["--beam=1000"], lc.lattice_text("u", lat).encode()).stdout.decode())["u"]
    r = _run(["--beam=1000", "--max-states=5", "--retry-cutoff=0"], lc.lattice_text("u", lat).encode())
    assert r.returncode == 0 and b"did not succeed" in r.stderr
    part = lc.parse_compact_text(r.stdout.decode())["u"]
    assert 0 < len(part["arcs"]) < len(full["arcs"])
    _check(lat, part, 1000.0, expect_complete=False)
    # --max-mem: tiny budget; with retries the raw lattice is pruned to a narrower beam and determinized again
    r = _run(["--beam=1000", "--max-mem=300"], lc.lattice_text("u", lat).encode())
    assert r.returncode == 0 and b"Did not reach requested beam" in r.stderr and b"retrying determinization" in r.stderr
    _check(lat, lc.parse_compact_text(r.stdout.decode())["u"], 1000.0, expect_complete=False)


def test_degenerate_inputs():
    # no final state reachable -> empty output lattice, warning, still "done"
    lat = dict(start=0, n=3, finals={}, arcs=[(0, 1, 1, 1, 1.0, 1.0), (1, 2, 2, 0, 1.0, 1.0)])
    r = _run([], lc.lattice_text("e", lat).encode())
    assert r.returncode == 0 and b"was empty" in r.stderr and r.stdout.decode().strip() == "e"
    # a single final start state
    lat = dict(start=0, n=1, finals={0: (0.5, 0.0)}, arcs=[])
    out = lc.parse_compact_text(_run([], lc.lattice_text("s", lat).encode()).stdout.decode())["s"]
    assert out["arcs"] == [] and list(out["finals"]) == [0] and abs(out["finals"][0][0] - 0.5) < 1e-6
    # words only on non-emitting arcs and a final weight with alignment left over after the last word
    lat = dict(start=0, n=4, finals={3: (0.25, 0.0)}, arcs=[(0, 1, 5, 0, 1.0, 2.0), (1, 2, 0, 7, 0.5, 0.0), (2, 3, 6, 0, 1.0, 1.0), (0, 2, 4, 7, 3.0, 3.0)])
    out = lc.parse_compact_text(_run([], lc.lattice_text("w", lat).encode()).stdout.decode())["w"]
    det = lc.enumerate_compact(out)
    assert list(det) == [(7,)] and det[(7,)][0][3] == (5, 6) and abs(det[(7,)][0][0] - 5.75) < 1e-5
    # a cycle cannot be sorted: error exit like the reference's KALDI_ERR in the wrapper
    lat = dict(start=0, n=2, finals={1: (0.0, 0.0)}, arcs=[(0, 1, 1, 1, 1.0, 1.0), (1, 0, 0, 0, 1.0, 0.0)])
    r = _run([], lc.lattice_text("c", lat).encode())
    assert r.returncode != 0 and b"Topological sorting" in r.stderr
    # command-line contract
    assert subprocess.run([PROG], capture_output=True).returncode == 1
    assert subprocess.run([PROG, "--acoustic-scale=0", "ark:-", "ark:-"], input=b"", capture_output=True).returncode != 0
    assert subprocess.run([PROG, "--write-compact=false", "ark:-", "ark:-"], input=b"", capture_output=True).returncode != 0


def _best_cost_for_words(lat, words):
    """cheapest raw path spelling exactly `words`: DP over (state, number of words consumed) in a topological order."""
    by_src = {}
    for a in lat["arcs"]: by_src.setdefault(a[0], []).append(a)
    indeg = {}
    for a in lat["arcs"]: indeg[a[1]] = indeg.get(a[1], 0) + 1
    order = []; ready = [s for s in range(lat["n"]) if indeg.get(s, 0) == 0]
    while ready:
        s = ready.pop(); order.append(s)
        for a in by_src.get(s, []):
            indeg[a[1]] -= 1
            if indeg[a[1]] == 0: ready.append(a[1])
    inf = float("inf"); best = {(lat["start"], 0): 0.0}; ans = inf
    for s in order:
        for k in range(len(words) + 1):
            c = best.get((s, k), inf)
            if c == inf: continue
            if k == len(words) and s in lat["finals"]: ans = min(ans, c + sum(lat["finals"][s]))
            for (_, d, tid, w, g, a) in by_src.get(s, []):
                if w == 0: key = (d, k)
                elif k < len(words) and words[k] == w: key = (d, k + 1)
                else: continue
                if c + g + a < best.get(key, inf): best[key] = c + g + a
    return ans


def _check_sampled(lat, out, nframes, samples=40, tol=0.02):
    """for lattices with too many paths to enumerate: determinism, one transition-id per frame on every path, sampled word sequences
    against a dynamic program over the raw lattice, and the overall best cost."""
    by_src = {}
    for a in out["arcs"]: by_src.setdefault(a[0], []).append(a)
    assert all(len({a[2] for a in v}) == len(v) and all(a[2] != 0 for a in v) for v in by_src.values())
    rng = np.random.default_rng(0)
    for _ in range(samples):
        s = out["start"]; words = []; cost = 0.0; ntid = 0
        while True:
            choices = by_src.get(s, [])
            if s in out["finals"] and (not choices or rng.random() < 0.3):
                cost += out["finals"][s][0] + out["finals"][s][1]; ntid += len(out["finals"][s][2]); break
            assert choices, "dead end at state %d" % s             # Connect ran: every state reaches a final state
            a = choices[int(rng.integers(len(choices)))]; words.append(a[2]); cost += a[3] + a[4]; ntid += len(a[5]); s = a[1]
        assert ntid == nframes
        assert abs(_best_cost_for_words(lat, words) - cost) <= tol, (words, cost)
    import sys; sys.setrecursionlimit(20000)
    by = {}
    for a in lat["arcs"]: by.setdefault(a[0], []).append(a)
    m1, m2 = {}, {}
    def bw(s):
        if s not in m1:
            m1[s] = min([sum(lat["finals"][s]) if s in lat["finals"] else np.inf] + [a[4] + a[5] + bw(a[1]) for a in by.get(s, [])])
        return m1[s]
    def bw_out(s):
        if s not in m2:
            m2[s] = min([out["finals"][s][0] + out["finals"][s][1] if s in out["finals"] else np.inf] + [a[3] + a[4] + bw_out(a[1]) for a in by_src.get(s, [])])
        return m2[s]
    assert abs(bw(lat["start"]) - bw_out(out["start"])) <= tol


def test_decoder_sized_lattice_sampled_paths():
    """a lattice of the size one utterance produces (200 frames, ~1000 states, several thousand arcs)"""
    lat = lc.random_lattice(11, frames=200, width=6, words=5, p_word=0.08)
    r = _run(["--beam=4"], lc.lattice_text("big", lat).encode())
    assert r.returncode == 0, r.stderr.decode()
    _check_sampled(lat, lc.parse_compact_text(r.stdout.decode())["big"], 200)


@pytest.mark.parametrize("seed", [1, 2])
def test_lattices_of_the_decoder_oracle(seed):
    """raw lattices exactly as the decoder emits them (states = tokens, epsilon-input arcs inside a frame, words on few arcs):
    the CPU restatement of LatticeFasterDecoder on a synthetic HCLG supplies them, binary table in, --beam = the lattice beam."""
    from kaldi_amd import synth
    from oracle import lattice_oracle as lo
    T = 60; N = 40
    f = synth.make_hclg(1500, 4000, N, seed=seed, start_degree=30)
    ll = (np.random.default_rng(seed + 1).standard_normal((T, N)) * 2.5).astype(np.float32)
    raw = lo.decode(f, ll, synth.tid2pdf(N), lo.Config(beam=15.0, lattice_beam=6.0, max_active=10000), 1)[0].connect()
    assert raw.num_arcs > 500 and (raw.arc_ilabel == 0).any() and (raw.arc_olabel != 0).any()
    lat = dict(start=raw.start_index(), n=raw.num_states, finals={int(s): (float(raw.st_final[s]), 0.0) for s in np.nonzero(np.isfinite(raw.st_final))[0]},
               arcs=[(int(s), int(d), int(i), int(o), float(g), float(a)) for s, d, i, o, g, a in zip(raw.arc_src, raw.arc_dst, raw.arc_ilabel, raw.arc_olabel, raw.arc_graph, raw.arc_ac)])
    r = _run(["--beam=6"], lc.lattice_binary("utt", lat))
    assert r.returncode == 0 and b"did not succeed" not in r.stderr, r.stderr.decode()
    out = lc.parse_compact_text(r.stdout.decode())["utt"]
    assert 0 < len(out["arcs"]) < raw.num_arcs
    _check_sampled(lat, out, T, samples=25, tol=0.01)
    # the best path of the determinized lattice is the decoder's best path
    bp = raw.best_path()
    by_src = {}
    for a in out["arcs"]: by_src.setdefault(a[0], []).append(a)
    memo = {}
    def best(s):
        if s not in memo:
            c = [(out["finals"][s][0] + out["finals"][s][1], (), out["finals"][s][2])] if s in out["finals"] else []
            for a in by_src.get(s, []):
                bc, bw_, bt = best(a[1]); c.append((a[3] + a[4] + bc, (a[2],) + bw_, a[5] + bt))
            memo[s] = min(c) if c else (np.inf, (), ())
        return memo[s]
    cost, words, tids = best(out["start"])
    assert list(words) == bp[1] and list(tids) == bp[0] and abs(cost - (bp[2] + bp[3])) < 1e-2


def test_parallel_program_writes_in_input_order_and_matches_the_serial_one():
    """lattice-determinize-pruned-parallel (worker threads behind a sequencer, = the pool the decoding programs hand their lattices
    to): same records for any thread count, input order kept, same lattices as the serial program up to state numbering."""
    par = PROG + "-parallel"
    lats = [lc.random_lattice(200 + i, frames=4 + i % 5, width=3 + i % 2, words=3) for i in range(40)]
    inp = b"".join(lc.lattice_binary("k%02d" % i, l) for i, l in enumerate(lats))
    outs = [subprocess.run([par, "--beam=5", "--acoustic-scale=0.5", "--num-threads=%d" % n, "ark:-", "ark,t:-"], input=inp, capture_output=True, timeout=120) for n in (1, 7)]
    assert all(o.returncode == 0 for o in outs), outs[1].stderr.decode()
    assert outs[0].stdout == outs[1].stdout
    par_l = lc.parse_compact_text(outs[0].stdout.decode())
    assert list(par_l) == ["k%02d" % i for i in range(40)]
    ser = _run(["--beam=5", "--acoustic-scale=0.5"], inp); assert ser.returncode == 0
    ser_l = lc.parse_compact_text(ser.stdout.decode())
    nonempty = 0
    for k in par_l:
        a, b = lc.enumerate_compact(par_l[k], 0.5), lc.enumerate_compact(ser_l[k], 0.5)
        assert a == b
        nonempty += len(a) > 0
    assert nonempty >= 30          # a few random lattices have no reachable final state: empty in both
    # a lattice that cannot be determinized (cycle) in the middle of the table: error exit, message on stderr
    bad = dict(start=0, n=2, finals={1: (0.0, 0.0)}, arcs=[(0, 1, 1, 1, 1.0, 1.0), (1, 0, 0, 0, 1.0, 0.0)])
    inp2 = b"".join(lc.lattice_binary("k%02d" % i, l) for i, l in enumerate(lats[:10])) + lc.lattice_binary("bad", bad) + lc.lattice_binary("after", lats[11])
    r = subprocess.run([par, "--num-threads=3", "ark:-", "ark,t:-"], input=inp2, capture_output=True, timeout=120)
    assert r.returncode != 0 and b"Topological sorting" in r.stderr
    assert subprocess.run([par], capture_output=True).returncode == 1


# ---- the two-pass (phone, then word) determinization the decoders apply: lattice-determinize-phone-pruned
PHONE_PROG = PROG.replace("lattice-determinize-pruned", "lattice-determinize-phone-pruned")

@pytest.fixture(scope="module")
def mdl(tmp_path_factory):
    """a transition model with 20 one-state phones: transition-ids 2p-1 (self-loop) and 2p (forward = start of the phone)"""
    from kaldi_amd import synth
    path = str(tmp_path_factory.mktemp("mdl") / "final.mdl")
    synth.make_tdnn(seed=1, dim=32, num_pdfs=20).write(path, as_mdl=True, num_pdfs=20, left_context=2, right_context=2)
    return path


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("beam", [1.0, 1000.0])
def test_phone_pruned_two_pass_properties(mdl, seed, beam):
    lat = lc.random_lattice(500 + seed, frames=5 + seed % 3, width=3 + seed % 2, words=2 + seed % 3, tids=40)
    inp = lc.lattice_text("u", lat).encode()
    r = subprocess.run([PHONE_PROG, "--beam=%g" % beam, mdl, "ark:-", "ark,t:-"], input=inp, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    out = lc.parse_compact_text(r.stdout.decode())["u"]
    raw, det = _check(lat, out, beam)
    # same word sequences and costs as the single word-level pass wherever the beam guarantees them
    one = lc.enumerate_compact(lc.parse_compact_text(_run(["--beam=%g" % beam], inp).stdout.decode())["u"])
    best = min(v[0][0] for v in det.values())
    for w, v in one.items():
        if v[0][0] <= best + beam - TOL:
            assert w in det and abs(det[w][0][0] - v[0][0]) <= TOL
    if beam >= 1000.0: assert set(det) == set(one) == set(raw)
    # --phone-determinize=false is the single pass
    r2 = subprocess.run([PHONE_PROG, "--beam=%g" % beam, "--phone-determinize=false", mdl, "ark:-", "ark,t:-"], input=inp, capture_output=True, timeout=120)
    assert r2.returncode == 0 and lc.enumerate_compact(lc.parse_compact_text(r2.stdout.decode())["u"]) == one


def test_phone_pruned_on_decoder_oracle_lattices_and_errors(mdl):
    from kaldi_amd import synth
    from oracle import lattice_oracle as lo
    T = 50; N = 20
    f = synth.make_hclg(1500, 4000, N, seed=5, start_degree=30)
    ll = (np.random.default_rng(9).standard_normal((T, N)) * 2.5).astype(np.float32)
    raw = lo.decode(f, ll, synth.tid2pdf(N), lo.Config(beam=15.0, lattice_beam=6.0, max_active=10000), 1)[0].connect()
    lat = dict(start=raw.start_index(), n=raw.num_states, finals={int(s): (float(raw.st_final[s]), 0.0) for s in np.nonzero(np.isfinite(raw.st_final))[0]},
               arcs=[(int(s), int(d), int(i), int(o), float(g), float(a)) for s, d, i, o, g, a in zip(raw.arc_src, raw.arc_dst, raw.arc_ilabel, raw.arc_olabel, raw.arc_graph, raw.arc_ac)])
    r = subprocess.run([PHONE_PROG, "--beam=6", mdl, "ark:-", "ark,t:-"], input=lc.lattice_binary("utt", lat), capture_output=True, timeout=120)
    assert r.returncode == 0 and b"did not succeed" not in r.stderr, r.stderr.decode()
    out = lc.parse_compact_text(r.stdout.decode())["utt"]
    assert 0 < len(out["arcs"]) < raw.num_arcs
    _check_sampled(lat, out, T, samples=25, tol=0.01)
    # a transition-id the model does not have, unsupported variants, usage
    bad = dict(start=0, n=3, finals={2: (0.0, 0.0)}, arcs=[(0, 1, 3, 1, 1.0, 1.0), (1, 2, 4000, 2, 1.0, 1.0)])
    r = subprocess.run([PHONE_PROG, mdl, "ark:-", "ark,t:-"], input=lc.lattice_text("b", bad).encode(), capture_output=True)
    assert r.returncode != 0 and b"transition-id 4000" in r.stderr
    assert subprocess.run([PHONE_PROG, "--write-compact=false", mdl, "ark:-", "ark:-"], input=b"", capture_output=True).returncode != 0
    assert subprocess.run([PHONE_PROG, mdl], capture_output=True).returncode == 1


def test_convert_lattice_folds_chains_and_keeps_every_path():
    """ConvertLattice (what --determinize-lattice=false writes in the reference's CUDA programs): no path is added, lost or re-costed;
    linear chains become single arcs carrying the chain's transition-ids; states come out topologically sorted."""
    tool = os.path.join(ROOT, "kaldi_amd", "bin", "k3-host-tool")
    for seed in range(6):
        lat = lc.random_lattice(700 + seed, frames=7, width=2 + seed % 3, words=3, p_word=0.2)
        r = subprocess.run([tool, "convert-lattice", "ark:-", "ark,t:-"], input=lc.lattice_binary("u", lat), capture_output=True, timeout=60)
        assert r.returncode == 0, r.stderr.decode()
        out = lc.parse_compact_text(r.stdout.decode())["u"]
        raw = lc.enumerate_raw(lat); conv = lc.enumerate_compact(out)
        assert set(raw) == set(conv)
        for w in raw:
            a = sorted((t, c) for c, _, _, t in raw[w]); b = sorted((t, c) for c, _, _, t in conv[w])
            assert [x[0] for x in a] == [x[0] for x in b] and np.allclose([x[1] for x in a], [x[1] for x in b], atol=2e-3), w      # text output: 6 significant digits
        assert all(d > s for (s, d, *_) in out["arcs"])
    # a pure chain with one word collapses into one arc (the final weight stays on the last state)
    chain = dict(start=0, n=5, finals={4: (0.5, 0.0)}, arcs=[(0, 1, 3, 9, 1.0, 2.0), (1, 2, 4, 0, 0.0, 1.0), (2, 3, 0, 0, 0.25, 0.0), (3, 4, 5, 0, 0.0, 1.5)])
    r = subprocess.run([tool, "convert-lattice", "ark:-", "ark,t:-"], input=lc.lattice_text("c", chain).encode(), capture_output=True)
    out = lc.parse_compact_text(r.stdout.decode())["c"]
    assert len(out["arcs"]) == 1 and out["arcs"][0][2] == 9 and out["arcs"][0][5] == (3, 4, 5) and abs(out["arcs"][0][3] - 1.25) < 1e-6 and abs(out["arcs"][0][4] - 4.5) < 1e-6
    assert list(out["finals"].values()) == [(0.5, 0.0, ())]


# ---- the restated determinizer against the REFERENCE's own source ----------------------------------------------------------------
# oracle/_ref/bin/ref-lattice-determinize is /root/reference/src/lat/determinize-lattice-pruned.cc compiled unmodified against a
# stand-in for the part of OpenFst it touches (third_party/minifst, oracle/build_ref.sh), driven like the reference's
# lattice-determinize-pruned / lattice-determinize-phone-pruned.  The programs here must print the same CompactLattices character for
# character: same states in the same order, same arcs, same weights to the 6 digits of the text format, same transition-id strings.
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-lattice-determinize")

def _ref_lattices(kind):
    if kind == "compact":       # CompactLattice records as input (the usual case in recipes: re-determinizing determinized lattices): made by our own program
        out = lc.parse_compact_text(_run(["--beam=1000"], "".join(lc.lattice_text(k, l) for k, l in _ref_lattices("random")).encode()).stdout.decode())
        return [(k, ("compact", v)) for k, v in out.items()]
    if kind == "random": return [("k%02d" % s, lc.random_lattice(s, frames=5 + s % 4, width=3 + s % 3, words=2 + s % 3, tids=40)) for s in range(12)]
    if kind == "ties": return [("t%02d" % s, lc.random_lattice(100 + s, frames=6, width=3, words=2, tids=40, quant=4)) for s in range(8)]
    if kind == "wide": return [("b%d" % s, lc.random_lattice(900 + s, frames=12, width=4, words=4, tids=40, p_word=0.5)) for s in range(4)]
    raise KeyError(kind)

REF_CASES = {   # name: (lattices, mode, beam, acoustic scale, extra options)
    "word_beam3": ("random", "word", 3.0, 1.0, ()), "word_wide_scaled": ("random", "word", 1000.0, 0.5, ()),
    "phone_beam3": ("random", "phone", 3.0, 1.0, ()), "phone_wide_scaled": ("random", "phone", 1000.0, 0.3, ()),
    "word_ties": ("ties", "word", 50.0, 1.0, ()), "phone_ties": ("ties", "phone", 50.0, 1.0, ()),
    "word_max_mem_retry": ("wide", "word", 8.0, 1.0, ("--max-mem=20000",)), "phone_max_mem_retry": ("wide", "phone", 1000.0, 1.0, ("--max-mem=2000",)),
    "word_minimize": ("random", "word", 1000.0, 1.0, ("--minimize=true",)), "phone_minimize": ("random", "phone", 3.0, 0.5, ("--minimize=true",)),
    "ties_minimize": ("ties", "phone", 50.0, 1.0, ("--minimize=true",)),
    "compact_input": ("compact", "word", 2.0, 1.0, ()),
    "phone_pass_only": ("random", "phone", 3.0, 1.0, ("--word-determinize=false",)), "no_pass": ("random", "phone", 3.0, 1.0, ("--word-determinize=false", "--phone-determinize=false")),
}

def write_model(td):
    from kaldi_amd import synth
    path = os.path.join(td, "final.mdl")
    synth.make_tdnn(seed=1, dim=32, num_pdfs=20).write(path, as_mdl=True, num_pdfs=20, left_context=2, right_context=2)
    return path

def _case_input(name, td):
    kind, mode, beam, scale, extra = REF_CASES[name]
    path = os.path.join(td, name + ".in.txt")
    def text(k, l):
        if not (isinstance(l, tuple) and l[0] == "compact"): return lc.lattice_text(k, l)
        c = l[1]; w = lambda g, a, t: "%r,%r,%s" % (g, a, "_".join(map(str, t))); by = {}
        for a in c["arcs"]: by.setdefault(a[0], []).append(a)
        lines = [k + " "]
        for s in [c["start"]] + sorted(set(list(by) + list(c["finals"])) - {c["start"]}):
            lines += ["%d\t%d\t%d\t%s" % (s, a[1], a[2], w(a[3], a[4], a[5])) for a in by.get(s, [])]
            if s in c["finals"]: lines.append("%d\t%s" % (s, w(*c["finals"][s])))
        return "\n".join(lines) + "\n\n" if c["start"] >= 0 else k + " \n\n"
    open(path, "w").write("".join(text(k, l) for k, l in _ref_lattices(kind)))
    return path, mode, beam, scale, list(extra)

def run_reference(name, td, mdl):
    path, mode, beam, scale, extra = _case_input(name, td)
    out = os.path.join(td, name + ".ref.txt")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([REF_EXE, mode, repr(beam), repr(scale), path, out] + ([mdl] if mode == "phone" else []) + extra, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    return open(out).read()

def run_ours(name, td, mdl):
    path, mode, beam, scale, extra = _case_input(name, td)
    exe = PROG if mode == "word" else PHONE_PROG
    r = subprocess.run([exe, "--beam=%r" % beam, "--acoustic-scale=%r" % scale] + extra + ([mdl] if mode == "phone" else []) + ["ark,t:" + path, "ark,t:-"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout

@pytest.mark.parametrize("name", sorted(REF_CASES))
def test_output_equals_the_reference_determinizer_character_for_character(name, tmp_path):
    import json
    td = str(tmp_path); model = write_model(td)
    ours = run_ours(name, td, model)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "det_ref_golden.json")))
    assert ours == gold[name]                                # recorded from the reference binary (tests/golden/make_golden_det.py)
    if os.path.exists(REF_EXE): assert ours == run_reference(name, td, model)      # and live, where oracle/_ref is present
    assert len(lc.parse_compact_text(ours)) == len(_ref_lattices(REF_CASES[name][0]))


def _convert_input(td):
    path = os.path.join(td, "conv.in.txt")
    open(path, "w").write("".join(lc.lattice_text("k%02d" % s, lc.random_lattice(700 + s, frames=7, width=2 + s % 3, words=3, p_word=0.2)) for s in range(16)))
    return path

def run_reference_convert(td):
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = os.path.join(td, "conv.ref.txt")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "bin", "ref-convert-lattice"), _convert_input(td), out], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    return open(out).read()

def test_convert_lattice_equals_the_reference_character_for_character(tmp_path):
    """ConvertLattice + Factor of the reference (fstext/lattice-utils-inl.h, fstext/factor-inl.h, compiled unmodified against the
    OpenFst stand-in): same folded chains, same state order, same text"""
    import json
    td = str(tmp_path)
    r = subprocess.run([os.path.join(ROOT, "kaldi_amd", "bin", "k3-host-tool"), "convert-lattice", "ark,t:" + _convert_input(td), "ark,t:-"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == json.load(open(os.path.join(ROOT, "tests", "golden", "det_ref_golden.json")))["convert_lattice"]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "bin", "ref-convert-lattice")): assert r.stdout == run_reference_convert(td)


def test_minimize_merges_states_and_keeps_the_language():
    """--minimize (push strings, push weights, merge equivalent states): fewer or equal arcs, the same word sequences with the same
    costs and alignments, still deterministic"""
    shrunk = 0
    for seed in range(10):
        lat = lc.random_lattice(seed, frames=5 + seed % 4, width=3 + seed % 3, words=2 + seed % 3, tids=40)
        inp = lc.lattice_text("u", lat).encode()
        a = lc.parse_compact_text(_run(["--beam=1000"], inp).stdout.decode())["u"]
        b = lc.parse_compact_text(_run(["--beam=1000", "--minimize=true"], inp).stdout.decode())["u"]
        assert len(b["arcs"]) <= len(a["arcs"]); shrunk += len(b["arcs"]) < len(a["arcs"])
        ea, eb = lc.enumerate_compact(a), lc.enumerate_compact(b)
        assert set(ea) == set(eb)
        for w in ea: assert ea[w][0][3] == eb[w][0][3] and np.allclose(ea[w][0][:3], eb[w][0][:3], atol=2e-3)
        assert len({(x[0], x[2]) for x in b["arcs"]}) == len(b["arcs"])
    assert shrunk >= 1


def test_lattice_best_path_program():
    """lattice-best-path on raw and on determinized (compact, binary) tables: the cheapest path by exhaustive enumeration, with the scales"""
    exe = PROG.replace("lattice-determinize-pruned", "lattice-best-path")
    lats = [("k%d" % s, lc.random_lattice(s, frames=5 + s % 4, width=3 + s % 3, words=2 + s % 3, tids=40)) for s in range(1, 9)]
    raw_txt = "".join(lc.lattice_text(k, l) for k, l in lats).encode()
    det_bin = _run(["--beam=1000"], raw_txt, binary_out=True).stdout
    det = lc.parse_compact_binary(det_bin)          # (a determinized lattice keeps one alignment per word sequence: its own paths are the truth for it)
    for inp, spec, truth in ((raw_txt, "ark,t:-", {k: lc.enumerate_raw(l) for k, l in lats}), (det_bin, "ark:-", {k: lc.enumerate_compact(det[k]) for k, _ in lats})):
        for lm, ac in ((1.0, 1.0), (1.0, 0.2), (0.5, 1.0)):
            r = subprocess.run([exe, "--lm-scale=%g" % lm, "--acoustic-scale=%g" % ac, spec, "ark,t:-", "ark,t:/dev/null"], input=inp, capture_output=True, timeout=60)
            assert r.returncode == 0, r.stderr.decode()
            got = {l.split()[0]: tuple(int(x) for x in l.split()[1:]) for l in r.stdout.decode().splitlines()}
            for k, l in lats:
                paths = truth[k]
                if not paths: assert k not in got; continue
                cost = lambda v: lm * v[1] + ac * v[2]
                best = min(((cost(v), w) for w, vs in paths.items() for v in vs))
                ties = [w for w, vs in paths.items() for v in vs if abs(cost(v) - best[0]) < 1e-4]
                assert got[k] in ties, (k, lm, ac, got[k], best)
    assert b"Overall cost per frame is" in r.stderr and b"Done" in r.stderr
    assert subprocess.run([exe], capture_output=True).returncode == 1
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".txt") as f:                    # --word-symbol-table: the words of every utterance on stderr, like the reference
        f.write("<eps> 0\nalpha 1\nbeta 2\ngamma 3\ndelta 4\n"); f.flush()
        r = subprocess.run([exe, "--word-symbol-table=" + f.name, "ark,t:-", "ark,t:-"], input=raw_txt, capture_output=True, timeout=60)
        assert r.returncode == 0
        words = {l.split()[0]: l.split()[1:] for l in r.stdout.decode().splitlines()}; names = {"1": "alpha", "2": "beta", "3": "gamma", "4": "delta"}
        for k, w in words.items(): assert (k + " " + " ".join(names[x] for x in w)).strip() in [l.strip() for l in r.stderr.decode().splitlines()]


@pytest.mark.skipif(not os.path.exists(REF_EXE), reason="oracle/_ref not built (needs /root/reference)")
def test_random_configurations_against_the_reference_binary(tmp_path):
    """fuzz: random lattice shapes (incl. cost grids that force exact ties), both entry points, random beam / acoustic scale / --minimize /
    --max-mem / --delta: every output identical to the reference determinizer's, character for character (a 600-lattice run of this loop
    found the one ordering detail the fixed cases had missed: --minimize runs before the wrapper's Connect)"""
    td = str(tmp_path); model = write_model(td); rng = np.random.default_rng(4321)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for rnd in range(8):
        kw = dict(frames=int(rng.integers(3, 12)), width=int(rng.integers(1, 5)), words=int(rng.integers(1, 5)), tids=40, p_word=float(rng.uniform(0.05, 0.9)), eps_arcs=bool(rng.integers(0, 2)),
                  quant=(None, 2, 4, 16)[int(rng.integers(0, 4))])
        open(f"{td}/in.txt", "w").write("".join(lc.lattice_text("r%d_%d" % (rnd, i), lc.random_lattice(int(rng.integers(0, 1 << 30)), **kw)) for i in range(10)))
        mode = ("word", "phone")[rnd % 2]; beam = float(rng.choice([0.5, 2.0, 5.0, 1000.0])); sc = float(rng.choice([1.0, 0.1, 0.5, 2.0]))
        extra = [[], ["--minimize=true"], ["--max-mem=%d" % int(rng.integers(500, 40000))], ["--minimize=true", "--delta=0.01"]][int(rng.integers(0, 4))]
        r = subprocess.run([REF_EXE, mode, repr(beam), repr(sc), f"{td}/in.txt", f"{td}/ref.txt"] + ([model] if mode == "phone" else []) + extra, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        m = subprocess.run([PROG if mode == "word" else PHONE_PROG, "--beam=%r" % beam, "--acoustic-scale=%r" % sc] + extra + ([model] if mode == "phone" else []) + [f"ark,t:{td}/in.txt", "ark,t:-"], capture_output=True, text=True)
        assert m.returncode == 0, m.stderr
        assert m.stdout == open(f"{td}/ref.txt").read(), (rnd, mode, beam, sc, extra, kw)


def test_scoring_filters_scale_penalty_prune():
    """lattice-scale | lattice-add-penalty | lattice-best-path as local/score.sh chains them, and lattice-prune: against exhaustive enumeration"""
    bindir = os.path.dirname(PROG); T = lambda n: os.path.join(bindir, n)
    lats = [("k%d" % s, lc.random_lattice(40 + s, frames=5 + s % 4, width=3 + s % 3, words=2 + s % 3, tids=40)) for s in range(8)]
    raw_txt = "".join(lc.lattice_text(k, l) for k, l in lats).encode()
    a = subprocess.run([T("lattice-scale"), "--inv-acoustic-scale=10", "ark,t:-", "ark:-"], input=raw_txt, capture_output=True); assert a.returncode == 0, a.stderr.decode()
    b = subprocess.run([T("lattice-add-penalty"), "--word-ins-penalty=0.5", "ark:-", "ark:-"], input=a.stdout, capture_output=True); assert b.returncode == 0, b.stderr.decode()
    c = subprocess.run([T("lattice-best-path"), "ark:-", "ark,t:-"], input=b.stdout, capture_output=True); assert c.returncode == 0, c.stderr.decode()
    got = {l.split()[0]: tuple(int(x) for x in l.split()[1:]) for l in c.stdout.decode().splitlines()}
    for k, l in lats:
        paths = lc.enumerate_raw(l)
        if not paths: assert k not in got; continue
        cost = lambda w, v: v[1] + v[2] / 10.0 + 0.5 * len(w)
        best = min(cost(w, v) for w, vs in paths.items() for v in vs)
        assert got[k] in [w for w, vs in paths.items() for v in vs if abs(cost(w, v) - best) < 1e-4], k
    # the penalty lands on every arc that carries a word, nowhere else: total path cost grows by 0.5 per word
    pen = lc.parse_compact_binary(b.stdout); sc = lc.parse_compact_binary(a.stdout)
    for k, _ in lats:
        ea, eb = lc.enumerate_compact(sc[k]), lc.enumerate_compact(pen[k])
        assert set(ea) == set(eb)
        for w in ea: assert abs(min(v[0] for v in eb[w]) - min(v[0] for v in ea[w]) - 0.5 * len(w)) < 1e-3
    # lattice-prune: nothing within the beam is lost, nothing is invented, costs unchanged (acoustic scale applied for the beam only)
    p = subprocess.run([T("lattice-prune"), "--beam=3", "--acoustic-scale=0.5", "--write-compact=false", "ark,t:-", "ark,t:-"], input=raw_txt, capture_output=True); assert p.returncode == 0, p.stderr.decode()
    assert b"pruned from on average" in p.stderr
    out = {}
    for rec in p.stdout.decode().split("\n\n"):
        ls_ = [x for x in rec.split("\n") if x.strip()]
        if not ls_: continue
        key = ls_[0].strip(); arcs = []; fin = {}; start = None
        for x in ls_[1:]:
            f = x.split("\t"); s = int(f[0]); start = s if start is None else start
            if len(f) <= 2: fin[s] = tuple(float(v) for v in f[1].split(",")) if len(f) == 2 else (0.0, 0.0)
            else: g, a_ = (float(v) for v in f[4].split(",")) if len(f) == 5 else (0.0, 0.0); arcs.append((s, int(f[1]), int(f[2]), int(f[3]), g, a_))
        out[key] = dict(start=start, n=1 + max([a_[1] for a_ in arcs] + [a_[0] for a_ in arcs] + list(fin) + [0]), finals=fin, arcs=arcs)
    for k, l in lats:
        raw = lc.enumerate_raw(l, 0.5)
        if not raw: continue
        best = min(v[0] for vs in raw.values() for v in vs)
        kept = lc.enumerate_raw(out[k], 0.5); kept_set = {(w, v[3]) for w, vs in kept.items() for v in vs}
        for w, vs in raw.items():
            for v in vs:
                if v[0] <= best + 3.0 - 1e-3: assert (w, v[3]) in kept_set, (k, w)
        assert kept_set <= {(w, v[3]) for w, vs in raw.items() for v in vs}
    assert subprocess.run([T("lattice-scale")], capture_output=True).returncode == 1
    assert subprocess.run([T("lattice-scale"), "--acoustic-scale=2", "--inv-acoustic-scale=3", "ark:-", "ark:-"], input=b"", capture_output=True).returncode != 0


# ---- word-level Minimum Bayes Risk decoding (kaldi_amd/host/k3_mbr.cc) against the reference's lat/sausages.cc --------------------------------------------
BIN = os.path.join(ROOT, "kaldi_amd", "bin")
def _parse_ref_mbr(text):
    out = {}; cur = None
    for line in text.splitlines():
        t = line.split()
        if not t: continue
        if t[0] == "words": cur["words"] = [int(x) for x in t[1:]]
        elif t[0] == "times": v = [float(x) for x in t[1:]]; cur["times"] = list(zip(v[0::2], v[1::2]))
        elif t[0] == "conf": cur["conf"] = [float(x) for x in t[1:]]
        elif t[0] == "risk": cur["risk"] = float(t[1])
        elif t[0] == "bins": cur["bins"] = []
        elif t[0] == "bin": cur["bins"].append(((float(t[1]), float(t[2])), [(int(e.split(":")[0]), float(e.split(":")[1])) for e in t[3:]]))
        else: cur = out.setdefault(t[0], {})
    return out

@pytest.mark.parametrize("decode_mbr", [True, False])
def test_mbr_decode_equals_the_reference_sausages(decode_mbr, tmp_path):
    """lattice-mbr-decode (k3_mbr.cc: EditDistance / AccStats / MbrDecode restated) against the reference's own lat/sausages.cc compiled unmodified over the OpenFst stand-in
    (oracle/_ref/bin/ref-mbr): MBR word sequence, Bayes risk, sausage bins (words, posteriors, times) and the one-best times on random lattices with competing words"""
    ref = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-mbr")
    if not os.path.exists(ref): pytest.skip("oracle/_ref not built")
    td = str(tmp_path); lats = {f"l{i}": lc.random_lattice(900 + i, frames=int(4 + i % 9), width=int(2 + i % 4), words=int(2 + i % 5), p_word=0.3 + 0.05 * (i % 6)) for i in range(60)}
    open(f"{td}/in.txt", "w").write("".join(lc.lattice_text(k, v) for k, v in lats.items()))
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
    r = subprocess.run([ref, f"{td}/in.txt", f"{td}/ref.txt", str(int(decode_mbr))], capture_output=True, text=True, env=env); assert r.returncode == 0, r.stderr[-2000:]
    want = _parse_ref_mbr(open(f"{td}/ref.txt").read())
    g = subprocess.run([os.path.join(BIN, "lattice-mbr-decode"), f"--decode-mbr={'true' if decode_mbr else 'false'}", f"ark,t:{td}/in.txt", f"ark,t:{td}/w.txt", f"ark,t:{td}/risk.txt", f"ark,t:{td}/saus.txt", f"ark,t:{td}/times.txt"],
                       capture_output=True, text=True); assert g.returncode == 0, g.stderr[-2000:]
    words = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in open(f"{td}/w.txt")}
    risk = {l.split()[0]: float(l.split()[1]) for l in open(f"{td}/risk.txt")}
    n_words = 0
    for line_s, line_t in zip(open(f"{td}/saus.txt"), open(f"{td}/times.txt")):
        key = line_s.split()[0]; w = want[key]
        assert words[key] == w["words"], key
        assert abs(risk[key] - w["risk"]) <= 1e-5 * max(1.0, abs(w["risk"])), (key, risk[key], w["risk"])
        bins = [[(int(b.split()[2 * i]), float(b.split()[2 * i + 1])) for i in range(len(b.split()) // 2)] for b in line_s[len(key):].replace("]", "").split("[")[1:]]
        tv = [float(x) for x in line_t[len(key):].replace(";", " ").split()]; times = list(zip(tv[0::2], tv[1::2]))
        assert len(bins) == len(w["bins"]) == len(times), key
        for (bt, be), got_b, got_t in zip(w["bins"], bins, times):
            assert [e[0] for e in got_b] == [e[0] for e in be] and np.allclose([e[1] for e in got_b], [e[1] for e in be], atol=2e-6), (key, got_b, be)
            assert np.allclose(got_t, bt, atol=1e-4), (key, got_t, bt)
        n_words += len(w["words"])
    assert n_words > 60 and len(words) == len(want) == 60


def test_lattice_postprocessor_ctm_of_a_decoded_lattice(tmp_path):
    """LatticePostprocessor::GetCTM (cudadecoder/lattice-postprocessor.cc:88-110) through libk3host: scales, word insertion penalty, MBR, times in seconds; the CTM lines are
    MergeSegmentsToCTMOutput's (two decimals)"""
    import ctypes
    from kaldi_amd import hostlib
    L = hostlib.load()
    lat = lc.random_lattice(77, frames=12, width=4, words=5, p_word=0.4)
    open(tmp_path / "in.txt", "w").write(lc.lattice_text("utt", lat)); open(tmp_path / "pp.conf", "w").write("--acoustic-scale=0.5\n--lm-scale=2.0\n--word-ins-penalty=0.25\n")
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.k3h_lattice_table_to_ctm(f"ark,t:{tmp_path}/in.txt".encode(), str(tmp_path / "pp.conf").encode(), ctypes.c_float(0.03), buf, len(buf))
    assert n > 0, hostlib.last_error()
    lines = buf.value.decode().splitlines(); assert lines and all(l.startswith("utt 0  ") and len(l.split()) == 6 for l in lines)
    # the same through the drop-in MBR program with the scales applied by hand: same words, times * 0.03
    g = subprocess.run([os.path.join(BIN, "lattice-mbr-decode"), "--acoustic-scale=0.5", "--lm-scale=2.0", "--one-best-times=true", f"ark,t:{tmp_path}/in.txt", f"ark,t:{tmp_path}/w.txt", "", "", f"ark,t:{tmp_path}/t.txt"], capture_output=True, text=True)
    assert g.returncode == 0, g.stderr[-1500:]
    # (the penalty only moves costs of word arcs: with 0.25 on a 12-frame lattice the MBR output rarely changes; compare the word COUNT and the monotone times)
    starts = [float(l.split()[3]) for l in lines]; assert starts == sorted(starts) and all(float(l.split()[4]) >= 0 for l in lines) and all(0.0 <= float(l.split()[6 - 1]) <= 1.0001 for l in lines)
