"""An INDEPENDENT check of third_party/minifst.  The decoder / determinizer / chain oracles under oracle/_ref are the reference's own
sources compiled over this stand-in for OpenFst's containers, so a misunderstanding inside the stand-in would be shared by the oracle and by the
product code that was written against it.  Here its algorithms run on random automata (tests/minifst_check.cc, compiled on the fly) and are held to
implementations that share nothing with it: scipy.sparse.csgraph for reachability / acyclicity / shortest distances, numpy.lexsort for arc order.
The second half checks the binary FST READERS against files whose bytes are put together by hand in this file from OpenFst's documented layouts
(fst/fst.h FstHeader, vector-fst.h, const-fst.h, compact-fst.h) -- bytes no writer of this repository produced.  CPU only."""
import os, struct, subprocess, numpy as np, pytest
from scipy.sparse import csr_matrix
from scipy.sparse import csgraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kaldi_amd", "bin")

@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("minifst") / "minifst_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I", os.path.join(ROOT, "third_party", "minifst"), os.path.join(ROOT, "tests", "minifst_check.cc"), "-o", out])
    return out

def random_fst(rng, n, m, acyclic=False, labels=6):
    """n states, ~m arcs; every state's final weight encodes its identity (s + 0.5) so that a renumbering can be read back; parallel arcs and ties on purpose"""
    src = rng.integers(0, n, m); dst = rng.integers(0, n, m)
    if acyclic:
        perm = rng.permutation(n); rank = np.empty(n, int); rank[perm] = np.arange(n)      # a hidden topological order
        keep = rank[src] != rank[dst]; src, dst = src[keep], dst[keep]
        swap = rank[src] > rank[dst]; src[swap], dst[swap] = dst[swap], src[swap].copy()
    il = rng.integers(0, labels, len(src)); ol = rng.integers(0, labels, len(src)); w = rng.integers(0, 33, len(src)) / 8.0      # (exact in float32: the automaton survives the text round trip bit for bit)
    order = np.argsort(src, kind="stable")      # stored order = insertion order per state
    return dict(n=n, start=int(rng.integers(0, n)), src=src[order], dst=dst[order], il=il[order], ol=ol[order], w=w[order], final={s: s + 0.5 for s in range(n)})

def to_text(f):
    lines = ["n %d %d" % (f["n"], f["start"])]
    lines += ["a %d %d %d %d %.9g" % t for t in zip(f["src"], f["dst"], f["il"], f["ol"], f["w"])]
    lines += ["f %d %.9g" % (s, w) for s, w in sorted(f["final"].items())]
    return "\n".join(lines) + "\n"

def from_text(txt):
    n = start = 0; arcs = []; final = {}
    for line in txt.splitlines():
        p = line.split()
        if p[0] == "n": n, start = int(p[1]), int(p[2])
        elif p[0] == "a": arcs.append((int(p[1]), int(p[2]), int(p[3]), int(p[4]), float(p[5])))
        elif p[0] == "f": final[int(p[1])] = float(p[2])
    return n, start, arcs, final

def run(exe, op, f):
    r = subprocess.run([exe, op], input=to_text(f), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout

def reach(n, src, dst, roots):
    """states reachable from `roots` (scipy breadth-first search over a CSR adjacency with one extra super-root)"""
    a = csr_matrix((np.ones(len(src) + len(roots)), (np.concatenate([src, np.full(len(roots), n)]), np.concatenate([dst, np.asarray(roots, int)]))), shape=(n + 1, n + 1))
    order = csgraph.breadth_first_order(a, n, directed=True, return_predecessors=False)
    seen = np.zeros(n + 1, bool); seen[order] = True
    return seen[:n]

@pytest.mark.parametrize("seed", range(12))
def test_connect_keeps_exactly_the_accessible_and_coaccessible_states_in_their_old_order(exe, seed):
    rng = np.random.default_rng(seed); n = int(rng.integers(2, 40)); f = random_fst(rng, n, int(rng.integers(1, 3 * n)))
    finals = sorted(rng.choice(n, size=int(rng.integers(0, max(1, n // 4) + 1)), replace=False).tolist()); f["final"] = {s: s + 0.5 for s in finals}
    got_n, got_start, got_arcs, got_final = from_text(run(exe, "connect", f))
    acc = reach(n, f["src"], f["dst"], [f["start"]]); co = reach(n, f["dst"], f["src"], finals) if finals else np.zeros(n, bool)
    keep = np.flatnonzero(acc & co)
    if len(keep) == 0 or not (acc & co)[f["start"]]:
        assert got_n == 0 and not got_arcs; return
    new = -np.ones(n, int); new[keep] = np.arange(len(keep))      # old relative order
    want_arcs = [(int(new[s]), int(new[d]), int(i), int(o), float(w)) for s, d, i, o, w in zip(f["src"], f["dst"], f["il"], f["ol"], f["w"]) if new[s] >= 0 and new[d] >= 0]
    assert got_n == len(keep) and got_start == new[f["start"]]
    assert got_arcs == want_arcs      # (arcs of a state keep their stored order)
    assert got_final == {int(new[s]): s + 0.5 for s in finals if new[s] >= 0}

@pytest.mark.parametrize("seed", range(12))
def test_topsort_is_a_renumbering_in_topological_order_and_refuses_cycles(exe, seed):
    rng = np.random.default_rng(100 + seed); n = int(rng.integers(2, 40)); acyclic = seed % 3 != 0
    f = random_fst(rng, n, int(rng.integers(1, 3 * n)), acyclic=acyclic)
    out = run(exe, "topsort", f)
    ncomp, _ = csgraph.connected_components(csr_matrix((np.ones(len(f["src"])), (f["src"], f["dst"])), shape=(n, n)), directed=True, connection="strong")
    has_cycle = ncomp < n or bool(np.any(f["src"] == f["dst"]))      # acyclic <=> every strongly connected component is a single state without a self-loop
    if has_cycle:
        assert out.strip() == "cyclic"; return
    got_n, got_start, got_arcs, got_final = from_text(out)
    assert got_n == n and len(got_final) == n
    old_of = {s: int(round(w - 0.5)) for s, w in got_final.items()}      # the identity every state carries in its final weight
    assert sorted(old_of.values()) == list(range(n)) and old_of[got_start] == f["start"]
    assert all(s < d for s, d, _, _, _ in got_arcs)      # topological
    want = {}      # per old state, its arcs in stored order
    for s, d, i, o, w in zip(f["src"], f["dst"], f["il"], f["ol"], f["w"]): want.setdefault(int(s), []).append((int(d), int(i), int(o), float(w)))
    got = {}
    for s, d, i, o, w in got_arcs: got.setdefault(old_of[s], []).append((old_of[d], i, o, w))
    assert got == want
    # OpenFst's TopSort numbers by reverse DFS finishing order from the start state first: the start state is state 0 whenever everything else it does not reach finishes before it
    assert got_start == min(s for s in range(n) if old_of[s] == f["start"])

@pytest.mark.parametrize("seed", range(8))
def test_arcsort_orders_on_input_then_output_label_like_openfst_ilabelcompare(exe, seed):
    rng = np.random.default_rng(200 + seed); n = int(rng.integers(1, 15)); f = random_fst(rng, n, int(rng.integers(1, 8 * n)), labels=4)
    _, _, got_arcs, _ = from_text(run(exe, "arcsort", f))
    order = np.lexsort((f["ol"], f["il"], f["src"]))      # by state, then (ilabel, olabel): fst/arcsort.h ILabelCompare compares the pair
    want_keys = [(int(f["src"][k]), int(f["il"][k]), int(f["ol"][k])) for k in order]
    assert [(s, i, o) for s, d, i, o, w in got_arcs] == want_keys      # arcs that agree in all three may come in any order (std::sort):
    canon = lambda arcs: sorted(arcs)
    assert canon(got_arcs) == canon([(int(s), int(d), int(i), int(o), float(w)) for s, d, i, o, w in zip(f["src"], f["dst"], f["il"], f["ol"], f["w"])])

def test_invert_swaps_the_labels_and_nothing_else(exe):
    rng = np.random.default_rng(7); f = random_fst(rng, 9, 30)
    _, start, got_arcs, got_final = from_text(run(exe, "invert", f))
    assert start == f["start"] and got_final == f["final"]
    assert got_arcs == [(int(s), int(d), int(o), int(i), float(w)) for s, d, i, o, w in zip(f["src"], f["dst"], f["il"], f["ol"], f["w"])]

@pytest.mark.parametrize("seed", range(8))
def test_shortest_path_has_the_cost_of_an_independent_shortest_distance(exe, seed):
    rng = np.random.default_rng(300 + seed); n = int(rng.integers(2, 30)); f = random_fst(rng, n, int(rng.integers(n, 4 * n)))
    finals = sorted(rng.choice(n, size=int(rng.integers(1, max(2, n // 3))), replace=False).tolist()); f["final"] = {s: float(rng.integers(0, 17)) / 8.0 for s in finals}
    got_n, got_start, got_arcs, got_final = from_text(run(exe, "shortestpath", f))
    # scipy: Bellman-Ford over the min-weight simple graph + a super-final state reached with the final weights
    best = {}
    for s, d, w in zip(f["src"], f["dst"], f["w"]): best[(int(s), int(d))] = min(best.get((int(s), int(d)), np.inf), float(w) + 1e-9)      # (+eps: an explicit zero is "no edge" for csgraph)
    for s, w in f["final"].items(): best[(s, n)] = w + 1e-9
    keys = list(best); a = csr_matrix(([best[k] for k in keys], ([k[0] for k in keys], [k[1] for k in keys])), shape=(n + 1, n + 1))
    dist = csgraph.bellman_ford(a, directed=True, indices=f["start"])[n]
    if not np.isfinite(dist):
        assert got_n == 0; return
    assert got_n == len(got_arcs) + 1 and len(got_final) == 1      # a linear automaton
    cost = sum(w for *_, w in got_arcs) + list(got_final.values())[0]
    assert abs(cost - dist) <= 1e-5 * (1 + len(got_arcs)), (cost, dist)
    have = {(int(s), int(d), int(i), int(o), float(np.float32(w))) for s, d, i, o, w in zip(f["src"], f["dst"], f["il"], f["ol"], f["w"])}
    assert all(any(abs(w - hw) < 1e-6 and (i, o) == (hi, ho) for _, _, hi, ho, hw in have) for _, _, i, o, w in got_arcs)      # made of arcs of the input

def path_set(n, start, arcs, final, drop_eps_pairs=True):
    """{(input label sequence without 0s, output label sequence without 0s): least total weight} over ALL successful paths of an acyclic transducer, by exhaustive walk (no
    shortest-distance algorithm, no closure: nothing in common with RmEpsilon).  Weights are multiples of 1/8: sums are exact."""
    out_arcs = {}
    for s, d, i, o, w in arcs: out_arcs.setdefault(s, []).append((d, i, o, w))
    best = {}
    def walk(s, iseq, oseq, w):
        if s in final:
            k = (iseq, oseq); t = w + final[s]
            if k not in best or t < best[k]: best[k] = t
        for d, i, o, aw in out_arcs.get(s, ()): walk(d, iseq + ((i,) if i else ()), oseq + ((o,) if o else ()), w + aw)
    if n and start >= 0: walk(start, (), (), 0.0)
    return best

@pytest.mark.parametrize("seed", range(24))
def test_rmepsilon_keeps_the_weighted_path_set_and_leaves_no_epsilon_arc(exe, seed):
    """RmEpsilon (used by the reference's word-align code compiled over minifst, which pins k3_mbr.cc) against an exhaustive enumeration of the paths before and
    after: same {(input string, output string): least weight}, no arc with both labels 0 left, and -- with connect -- every remaining state on a successful path."""
    rng = np.random.default_rng(900 + seed); n = int(rng.integers(2, 11)); f = random_fst(rng, n, int(rng.integers(n, 3 * n)), acyclic=True, labels=3)
    eps = rng.random(len(f["src"])) < 0.45; f["il"] = np.where(eps, 0, f["il"]); f["ol"] = np.where(eps, 0, f["ol"])      # real epsilon arcs (0:0) next to 0:x, x:0 and x:y arcs
    finals = sorted(rng.choice(n, size=int(rng.integers(1, max(2, n // 2) + 1)), replace=False).tolist()); f["final"] = {s: float(rng.integers(0, 17)) / 8.0 for s in finals}
    arcs_in = [(int(s), int(d), int(i), int(o), float(w)) for s, d, i, o, w in zip(f["src"], f["dst"], f["il"], f["ol"], f["w"])]
    want = path_set(n, f["start"], arcs_in, f["final"])
    for op in ("rmepsilon", "rmepsilon_noconnect"):
        got_n, got_start, got_arcs, got_final = from_text(run(exe, op, f))
        assert not any(i == 0 and o == 0 for _, _, i, o, _ in got_arcs), op
        assert path_set(got_n, got_start if got_n else -1, got_arcs, got_final) == want, op
        if op == "rmepsilon" and got_n:
            src = np.array([a[0] for a in got_arcs], int); dst = np.array([a[1] for a in got_arcs], int)
            assert reach(got_n, src, dst, [got_start]).all() and reach(got_n, dst, src, sorted(got_final)).all()      # trim
        if op == "rmepsilon_noconnect": assert got_n == n and got_start == f["start"]      # states keep their numbers
        # no two arcs of a state with the same labels and destination (OpenFst's RmEpsilon combines them with Plus)
        assert len({(s, d, i, o) for s, d, i, o, _ in got_arcs}) == len(got_arcs), op

def test_project_and_map_touch_the_labels_only(exe):
    rng = np.random.default_rng(11); f = random_fst(rng, 9, 30)
    base = [(int(s), int(d), int(i), int(o), float(w)) for s, d, i, o, w in zip(f["src"], f["dst"], f["il"], f["ol"], f["w"])]
    for op, fn in (("project_input", lambda i, o: (i, i)), ("project_output", lambda i, o: (o, o)), ("map_ilabel_plus1", lambda i, o: (i + 1, o))):
        _, start, got_arcs, got_final = from_text(run(exe, op, f))
        assert start == f["start"] and got_final == f["final"]
        assert got_arcs == [(s, d) + fn(i, o) + (w,) for s, d, i, o, w in base], op

# ------------------------------------------------------------------------------------------------------------------------------------------------------
# Binary containers, bytes by hand.  FstHeader (fst/fst.h, FstHeader::Write): int32 magic 2125659606; fst type and arc type as (int32 length, bytes);
# int32 version; int32 flags (1 = has input symbols, 2 = has output symbols, 4 = aligned); uint64 properties; int64 start; int64 num states; int64 num arcs.
MAGIC = 2125659606
def header(ftype, version, flags, start, ns, na, props=0):
    s = lambda t: struct.pack("<i", len(t)) + t
    return struct.pack("<i", MAGIC) + s(ftype) + s(b"standard") + struct.pack("<iiQqqq", version, flags, props, start, ns, na)

# the automaton every container below holds: state 1 is the start state; state 2 has no arcs and is final; state 3 is neither final nor left
STATES = [  # (final weight or None, [(ilabel, olabel, weight, nextstate)])
    (None, [(3, 3, 0.5, 2), (1, 1, 0.25, 3)]),
    (1.5, [(2, 2, 0.0, 0), (2, 2, 1.0, 2), (7, 7, 2.5, 1)]),
    (0.0, []),
    (None, []),
]
INF = float("inf")
def want_arrays():
    off = [0]; il = []; ol = []; w = []; nx = []
    for _, arcs in STATES:
        for a in arcs: il.append(a[0]); ol.append(a[1]); w.append(a[2]); nx.append(a[3])
        off.append(len(il))
    return off, il, ol, w, nx, [INF if fw is None else fw for fw, _ in STATES]

def vector_bytes():
    # VectorFst<StdArc>::Write (vector-fst.h): header (version 2), then per state: float final weight, int64 number of arcs, arcs {int32 ilabel, int32 olabel, float weight, int32 nextstate}
    b = header(b"vector", 2, 0, 1, len(STATES), sum(len(a) for _, a in STATES))
    for fw, arcs in STATES:
        b += struct.pack("<fq", INF if fw is None else fw, len(arcs))
        for a in arcs: b += struct.pack("<iifi", *a)
    return b

def const_bytes(aligned):
    # ConstFst<StdArc, uint32>::Write (const-fst.h): header (version 2; aligned files: flag 4 and zero padding to 16 bytes in front of each array), the state array
    # {float final; uint32 pos; uint32 narcs; uint32 niepsilons; uint32 noepsilons} (20 bytes each), then the arc array (16 bytes each)
    b = header(b"const", 2, 4 if aligned else 0, 1, len(STATES), sum(len(a) for _, a in STATES))
    pad = lambda x: x + b"\0" * ((-len(x)) % 16) if aligned else x
    b = pad(b); pos = 0
    for fw, arcs in STATES:
        b += struct.pack("<fIIII", INF if fw is None else fw, pos, len(arcs), sum(a[0] == 0 for a in arcs), sum(a[1] == 0 for a in arcs)); pos += len(arcs)
    b = pad(b)
    for _, arcs in STATES:
        for a in arcs: b += struct.pack("<iifi", *a)
    return b

def _check_reader(path, tmp_path):
    from kaldi_amd.fst import Fst
    tool = os.path.join(BIN, "k3-host-tool")
    off, il, ol, w, nx, fin = want_arrays()
    r = subprocess.run([tool, "fstinfo", path], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    assert r.stdout.split()[:3] == [str(len(STATES)), str(len(il)), "1"], r.stdout
    out = str(tmp_path / "copy.fst")
    assert subprocess.run([tool, "copy-fst", path, out], capture_output=True, text=True).returncode == 0
    # what the C++ reader understood (re-written as a vector FST) and -- for the vector container, the only one it knows -- the Python reader on the hand-made bytes themselves
    for g in [Fst.read_openfst(out)] + ([Fst.read_openfst(path)] if open(path, "rb").read()[8:14] == b"vector" else []):
        assert g.start == 1 and list(g.arc_offsets) == off and list(g.ilabel) == il and list(g.olabel) == ol and list(g.nextstate) == nx
        assert np.array_equal(np.asarray(g.weight, np.float32), np.asarray(w, np.float32)) and np.array_equal(np.asarray(g.final, np.float32), np.asarray(fin, np.float32))

def test_vector_fst_reader_on_bytes_written_by_hand(tmp_path):
    p = str(tmp_path / "v.fst"); open(p, "wb").write(vector_bytes()); _check_reader(p, tmp_path)

@pytest.mark.parametrize("aligned", [False, True])
def test_const_fst_reader_on_bytes_written_by_hand(aligned, tmp_path):
    p = str(tmp_path / "c.fst"); open(p, "wb").write(const_bytes(aligned)); _check_reader(p, tmp_path)

def test_reader_refuses_what_it_does_not_understand(tmp_path):
    tool = os.path.join(BIN, "k3-host-tool")
    for name, data in (("trunc", vector_bytes()[:-5]), ("magic", b"\0\0\0\0" + vector_bytes()[4:]), ("arctype", vector_bytes().replace(b"standard", b"log\0\0\0\0\0")),
                       ("symbols", header(b"vector", 2, 1, 0, 0, 0)), ("type", header(b"ngram", 2, 0, 0, 0, 0))):
        p = str(tmp_path / (name + ".fst")); open(p, "wb").write(data)
        r = subprocess.run([tool, "fstinfo", p], capture_output=True, text=True)
        assert r.returncode != 0, (name, r.stdout)

# compact_acceptor: chain::Supervision::Write (chain/chain-supervision.cc:549-611) stores the numerator FST with fst::StdCompactAcceptorFst::WriteFst -- CompactFst<StdArc,
# AcceptorCompactor<StdArc>, uint32>, type "compact_acceptor" (compact-fst.h): header (version 2, properties as computed by the writer), then the state index array --
# (num states + 1) uint32 offsets into the compact-element array -- and the elements {int32 label, float weight, int32 nextstate} (12 bytes); a FINAL state's first element is its
# final weight with label -1 and nextstate -1; header.num_arcs counts the real arcs only.
def compact_acceptor_bytes(start, states, props):
    offs = [0]; elems = b""; narcs = 0
    for fw, arcs in states:
        k = 0
        if fw is not None: elems += struct.pack("<ifi", -1, fw, -1); k += 1
        for lab, w, nx in arcs: elems += struct.pack("<ifi", lab, w, nx); k += 1; narcs += 1
        offs.append(offs[-1] + k)
    return header(b"compact_acceptor", 2, 0, start, len(states), narcs, props) + struct.pack("<%dI" % len(offs), *offs) + elems

CP = os.path.join(ROOT, "kaldi_amd", "adapter", "_build", "nnet3-chain-copy-egs")
@pytest.mark.skipif(not os.path.exists(CP), reason="kaldi_amd/adapter/_build/nnet3-chain-copy-egs is built by kaldi_amd/adapter/build.sh where /root/reference exists")
def test_compact_acceptor_container_of_chain_examples_against_bytes_written_by_hand(tmp_path):
    """the FST inside a binary NnetChainExample (Supervision::Write) must BE the hand-assembled compact_acceptor container of the same automaton, byte for byte except the
    informational properties word; and a binary archive whose container was replaced by the hand-assembled bytes reads back to the same text"""
    from kaldi_amd import synth
    from tests import chain_egs as ce
    td = str(tmp_path); B, T, P, s = 2, 5, 20, 3; lc = rc = 4; Tin = (T - 1) * s + 1 + lc + rc
    fsts = [synth.make_supervision_fst(T, P, seed=77 + i) for i in range(B)]; merged = synth.merge_supervision_fsts(fsts)
    x = np.random.default_rng(5).standard_normal((Tin * B, 8)).astype(np.float32)
    ce.write_chain_egs_text(f"{td}/a.txt", [ce.minibatch("mb0", x, B, T, s, lc, rc, P, merged_fst=merged)])
    run = lambda a, b: subprocess.run([CP, a, b], capture_output=True, text=True)
    assert run(f"ark,t:{td}/a.txt", f"ark:{td}/b.egs").returncode == 0
    blob = open(f"{td}/b.egs", "rb").read(); i = blob.find(struct.pack("<i", MAGIC)); assert i >= 0
    # the automaton as it was handed to the writer (the text archive keeps the state numbers and each state's arc order; weights are float32)
    assert run(f"ark:{td}/b.egs", f"ark,t:{td}/c.txt").returncode == 0
    hand_states = []
    for q in range(merged.num_states):
        arcs = [(int(merged.ilabel[a]), float(np.float32(merged.weight[a])), int(merged.nextstate[a])) for a in range(int(merged.arc_offsets[q]), int(merged.arc_offsets[q + 1]))]
        hand_states.append((float(np.float32(merged.final[q])) if np.isfinite(merged.final[q]) else None, arcs))
    hlen = len(header(b"compact_acceptor", 2, 0, 0, 0, 0)); props_at = hlen - 8 - 24      # the uint64 in front of start / num states / num arcs
    props = struct.unpack_from("<Q", blob, i + props_at)[0]
    hand = compact_acceptor_bytes(int(merged.start), hand_states, props)
    assert blob[i:i + len(hand)] == hand, "the container Supervision::Write produced is not the documented compact_acceptor layout of the same automaton"
    # a reader fed the hand-assembled container (spliced into the archive in place of the writer's bytes; properties zeroed: a reader must not depend on them)
    hand0 = compact_acceptor_bytes(int(merged.start), hand_states, 0)
    open(f"{td}/h.egs", "wb").write(blob[:i] + hand0 + blob[i + len(hand):])
    assert run(f"ark:{td}/h.egs", f"ark,t:{td}/d.txt").returncode == 0
    assert open(f"{td}/d.txt").read() == open(f"{td}/c.txt").read()
