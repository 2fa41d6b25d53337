"""The decoding tail against the reference's own sources, end to end (CPU): the reference's LatticeFasterDecoder followed by the
reference's DeterminizeLatticePhonePrunedWrapper (both compiled unmodified against the OpenFst stand-in, oracle/_ref/bin) versus the
chain the GPU path implements -- the decoder's order-independent two-pass definition (oracle mode 1, which the HIP decoder equals bit
for bit in tests/test_decoder_gpu.py) followed by kaldi_amd/host's determinizer.

The two raw lattices are not identical by design (the reference creates a few extra tokens/arcs while its cutoff is still
loose; which ones depends on its hash-table order).  What must hold, and is checked here on every case of tests/decoder_cases.py:
  * the same best path: words, transition-ids, graph and acoustic cost;
  * every word sequence of our determinized lattice is in the reference's determinized lattice; the reference's cost for it is the
    same or, for a few sequences far from the best path, lower (its extra arcs offer a cheaper alignment) -- never higher.
Skipped where oracle/_ref is absent."""
import os, subprocess, numpy as np, pytest
from tests import decoder_cases as dc, lattice_cases as lc
from oracle import lattice_oracle as lo, ref_decoder as rd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DET = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-lattice-determinize")
OUR_DET = os.path.join(ROOT, "kaldi_amd", "bin", "lattice-determinize-phone-pruned")
pytestmark = pytest.mark.skipif(not (rd.available() and os.path.exists(REF_DET)), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def mdl(tmp_path_factory):
    import __graft_entry__ as ge
    ge.build()
    from kaldi_amd import synth
    path = str(tmp_path_factory.mktemp("chain") / "final.mdl")
    synth.make_tdnn(seed=1, dim=32, num_pdfs=dc.N, ).write(path, as_mdl=True, num_pdfs=dc.N, left_context=2, right_context=2)
    return path


def _trim(n, start, finals, arcs):
    """fst::Connect on a dict lattice (the decoders' wrappers trim before they determinize)"""
    out, inn = {}, {}
    for a in arcs: out.setdefault(a[0], []).append(a[1]); inn.setdefault(a[1], []).append(a[0])
    def reach(seeds, adj):
        seen = set(seeds); st = list(seeds)
        while st:
            s = st.pop()
            for d in adj.get(s, []):
                if d not in seen: seen.add(d); st.append(d)
        return seen
    keep = reach([start], out) & reach(list(finals), inn)
    order = sorted(keep); new = {s: i for i, s in enumerate(order)}
    return dict(start=new[start], n=len(order), finals={new[s]: w for s, w in finals.items() if s in keep},
                arcs=[(new[a[0]], new[a[1]]) + tuple(a[2:]) for a in arcs if a[0] in keep and a[1] in keep])


@pytest.mark.parametrize("name", sorted(n for n in dc.CASES if len(dc.CASES[n]) == 4))      # (the 6024-pdf bench case needs its own model: covered in DESIGN 2.2 by hand)
def test_reference_decoder_and_determinizer_vs_our_chain(name, mdl, tmp_path):
    f, t2p, ll, kw = dc.make(name)
    cfg = lo.Config(**kw); beam = float(kw["lattice_beam"]); td = str(tmp_path)
    # reference chain
    r = rd.decode(f, ll, t2p, cfg)
    ref_raw = _trim(r["frame"].size, r["start"], {int(s): (float(g), float(a)) for s, (g, a) in enumerate(zip(r["final_graph"], r["final_ac"])) if np.isfinite(g)},
                    [(int(s), int(d), int(i), int(o), float(g), float(a)) for s, d, i, o, g, a in zip(r["src"], r["dst"], r["ilabel"], r["olabel"], r["graph"], r["ac"])])
    open(f"{td}/ref_raw.txt", "w").write(lc.lattice_text("u", ref_raw))
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([REF_DET, "phone", repr(beam), "1.0", f"{td}/ref_raw.txt", f"{td}/ref_det.txt", mdl], capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stderr
    ref_det = lc.parse_compact_text(open(f"{td}/ref_det.txt").read())["u"]
    # our chain: two-pass decoder definition (= the HIP decoder) + host determinizer
    raw = lo.decode(f, ll, t2p, cfg, 1)[0].connect()
    ours_raw = dict(start=raw.start_index(), n=raw.num_states, finals={int(s): (float(raw.st_final[s]), 0.0) for s in np.nonzero(np.isfinite(raw.st_final))[0]},
                    arcs=[(int(s), int(d), int(i), int(o), float(g), float(a)) for s, d, i, o, g, a in zip(raw.arc_src, raw.arc_dst, raw.arc_ilabel, raw.arc_olabel, raw.arc_graph, raw.arc_ac)])
    open(f"{td}/our_raw.txt", "w").write(lc.lattice_text("u", ours_raw))
    p = subprocess.run([OUR_DET, "--beam=%r" % beam, mdl, f"ark,t:{td}/our_raw.txt", f"ark,t:{td}/our_det.txt"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    our_det = lc.parse_compact_text(open(f"{td}/our_det.txt").read())["u"]
    A, B = lc.enumerate_compact(ref_det, limit=2_000_000), lc.enumerate_compact(our_det, limit=2_000_000)
    assert len(B) > 0 and all(len(v) == 1 for v in A.values()) and all(len(v) == 1 for v in B.values())
    # same best path
    wa, wb = min(A, key=lambda w: A[w][0][0]), min(B, key=lambda w: B[w][0][0])
    assert wa == wb and A[wa][0][3] == B[wb][0][3] and np.allclose(A[wa][0][:3], B[wb][0][:3], atol=2e-3), (wa, wb)
    # containment: our raw lattice is a sub-lattice of the reference's (same costs arc for arc), so every word sequence we keep is in
    # the reference's lattice and the reference's cost for it can only be lower or equal (its extra arcs may offer a cheaper alignment)
    # (with a binding --max-active / --min-active the two token sets differ before the limit is applied, the limit then cuts at different
    # costs, and neither lattice contains the other: only the best path is asserted for those cases)
    limits_bind = "max_active" in kw or "min_active" in kw
    same = cheaper = missing = 0
    for w, v in B.items():
        if w not in A: missing += 1; continue
        if not limits_bind: assert A[w][0][0] <= v[0][0] + 2e-3, (name, w, A[w][0][0], v[0][0])
        if abs(A[w][0][0] - v[0][0]) <= 2e-3: same += 1
        else: cheaper += 1
    extra = [A[w][0][0] - A[wa][0][0] for w in A if w not in B]
    print(name, "word sequences: reference %d, ours %d (%d same cost, %d different cost, %d not in the reference's); the reference's extra ones are %s above the best path"
          % (len(A), len(B), same, cheaper, missing, ("%.2f .. %.2f" % (min(extra), max(extra))) if extra else "none"))
    if not limits_bind: assert missing == 0
