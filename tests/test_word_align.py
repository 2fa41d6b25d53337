"""WordAlignLattice (lat/word-align-lattice.cc) in the host tail: the restatement in kaldi_amd/host/k3_mbr.cc against the reference's own source compiled unmodified
over the OpenFst stand-in (oracle/_ref/bin/ref-word-align = WordAlignLattice followed by MinimumBayesRisk, the two steps of LatticePostprocessor::GetCTM when a
word-boundary file is configured, cudadecoder/lattice-postprocessor.cc:55-110).  Lattices are built from a small lexicon over word-position-dependent phones
(silence / begin / internal / end / singleton), with competing words, optional silences, both self-loop orders (--reorder), zero and non-zero silence / partial-word
labels, word labels early or late inside their word, and broken inputs (an utterance cut inside a word, the wrong --reorder option).  The aligned lattice must be the
reference's ARC FOR ARC (same states, same order, weights to 9 digits, same transition-id strings), and so must the MBR words / times / confidences.  CPU only."""
import json, os, subprocess, numpy as np, pytest
from tests import lattice_cases as lc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "bin", "ref-word-align")
TOOL = os.path.join(ROOT, "kaldi_amd", "bin", "k3-host-tool")
GOLD = os.path.join(ROOT, "tests", "golden", "word_align_golden.json")

SIL, BEG, END, INT, ONE = [1, 2], [3, 4, 5, 6], [7, 8, 9, 10], [11, 12, 13, 14], [15, 16, 17, 18]
def write_model(td):
    """20 phones with a one-state topology: transition-ids 2p - 1 (self-loop) and 2p (leaves the phone: TransitionModel::IsFinal)"""
    from kaldi_amd import synth
    path = os.path.join(td, "final.mdl")
    synth.make_tdnn(seed=1, dim=32, num_pdfs=20).write(path, as_mdl=True, num_pdfs=20, left_context=2, right_context=2)
    wb = os.path.join(td, "word_boundary.int")
    with open(wb, "w") as f:
        for kind, ps in (("nonword", SIL), ("begin", BEG), ("end", END), ("internal", INT), ("singleton", ONE)):
            for p in ps: f.write("%d %s\n" % (p, kind))
    return path, wb

def phone_tids(p, dur, reorder):
    loop, leave = 2 * p - 1, 2 * p
    return [leave] + [loop] * (dur - 1) if reorder else [loop] * (dur - 1) + [leave]

def make_lattice(seed, reorder=True, segments=4, alts=3, cut=False, label_pos="first", p_sil=0.5):
    """a word lattice: boundary nodes 0 .. segments; between consecutive nodes `alts` alternatives of EQUAL duration (state times must be unique), each an optional silence
    followed by a word -- one singleton phone, or begin (internal)* end -- expanded into frame-level arcs (ilabel = transition-id, the word's label on its first or last arc).
    cut: the last segment stops inside a word (a forced-out utterance)."""
    rng = np.random.default_rng(seed); arcs = []; n = segments + 1; nxt = [n]
    def new_state(): nxt[0] += 1; return nxt[0] - 1
    lexicon = {w: ([int(rng.choice(ONE))] if rng.uniform() < 0.3 else [int(rng.choice(BEG))] + [int(rng.choice(INT)) for _ in range(int(rng.integers(0, 3)))] + [int(rng.choice(END))]) for w in range(1, 9)}
    for seg in range(segments):
        dur = int(rng.integers(6, 12)); seqs = []      # (at most 5 units -- silence + 4 phones -- always fit)
        for _ in range(alts):
            w = int(rng.integers(1, 9)); phones = list(lexicon[w]); sil = rng.uniform() < p_sil
            units = ([("sil", int(rng.choice(SIL)))] if sil else []) + [("w", p) for p in phones]
            d = np.ones(len(units), int)
            for _ in range(dur - len(units)): d[int(rng.integers(0, len(units)))] += 1
            seq = []; first_word_arc = None      # (transition-id, word label) per frame
            for (kind, p), k in zip(units, d):
                if kind == "w" and first_word_arc is None: first_word_arc = len(seq)
                seq += [(x, 0) for x in phone_tids(p, int(k), reorder)]
            at = first_word_arc if label_pos == "first" else len(seq) - 1
            seq[at] = (seq[at][0], w); seqs.append(seq)
        if cut and seg == segments - 1:      # a forced-out utterance: every alternative stops a frame or two before its word ends
            m = max(1, dur - int(rng.integers(1, 3))); seqs = [q[:m] for q in seqs]
        for seq in seqs:
            src = seg
            for i, (tid, lab) in enumerate(seq):
                dst = seg + 1 if i == len(seq) - 1 else new_state()
                arcs.append((src, dst, tid, lab, float(np.round(rng.uniform(0, 1), 3)), float(np.round(rng.uniform(0, 2), 3)))); src = dst
    return dict(start=0, n=nxt[0], finals={segments: (float(np.round(rng.uniform(0, 1), 3)), 0.0)}, arcs=arcs)

CASES = {   # name: (lattice kwargs, seeds, reorder option handed to the aligner, silence label, partial-word label, max-expand)
    "reorder": (dict(reorder=True), range(10), 1, 0, 0, 0),
    "no_reorder": (dict(reorder=False), range(10, 18), 0, 0, 0, 0),
    "labels": (dict(reorder=True, p_sil=0.8), range(20, 28), 1, 901, 902, 0),
    "label_on_last_arc": (dict(reorder=True, label_pos="last"), range(30, 36), 1, 0, 0, 0),
    "forced_out": (dict(reorder=True, cut=True), range(40, 48), 1, 0, 903, 0),
    "wrong_reorder_option": (dict(reorder=False, segments=2), range(50, 54), 1, 0, 0, 0),
    "max_expand": (dict(reorder=True, segments=24, alts=8), range(60, 62), 1, 0, 0, 0.001),      # (max-states = 1000 + max-expand * states: these lattices need more)
}

def _input(name, td):
    kw, seeds, reorder, sil, part, mx = CASES[name]
    path = os.path.join(td, name + ".in.txt")
    open(path, "w").write("".join(lc.lattice_text("%s%02d" % (name[:3], s), make_lattice(s, **kw)) for s in seeds))
    return path, [str(reorder), str(sil), str(part), repr(mx)]

def run_ours(name, td, mdl, wb):
    path, opts = _input(name, td); out = os.path.join(td, name + ".ours.txt")
    r = subprocess.run([TOOL, "word-align", mdl, wb, "ark,t:" + path, out] + opts, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(out).read(), r.stderr

def run_reference(name, td, mdl, wb):
    path, opts = _input(name, td); out = os.path.join(td, name + ".ref.txt")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([REF, mdl, wb, path, out] + opts, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(out).read()

@pytest.mark.parametrize("name", sorted(CASES))
def test_word_aligned_lattice_and_ctm_equal_the_reference(name, tmp_path):
    td = str(tmp_path); mdl, wb = write_model(td)
    ours, log = run_ours(name, td, mdl, wb)
    assert ours == json.load(open(GOLD))[name]      # recorded from the reference binary (tests/golden/make_golden_word_align.py)
    if os.path.exists(REF): assert ours == run_reference(name, td, mdl, wb)      # and live, where oracle/_ref is present
    recs = ours.split("end\n")[:-1]; assert len(recs) == len(CASES[name][1])
    oks = [int(r.split("\nok ")[1][0]) for r in recs]
    if name in ("reorder", "no_reorder", "labels", "label_on_last_arc"):
        assert all(oks)
        for r in recs:      # every arc of an aligned lattice is one whole word or one silence: its transition-ids start a phone and end by leaving one
            for line in r.splitlines():
                if line.startswith("a "):
                    tids = [int(x) for x in line.split()[6].split("_")] if len(line.split()) > 6 else []
                    assert tids, line
    if name in ("forced_out", "wrong_reorder_option", "max_expand"): assert not all(oks) and "WARNING" in log

def test_word_times_are_word_boundaries(tmp_path):
    """what the alignment is for: on a lattice with ONE path the CTM times must be the words' true frame spans (silences dropped), whatever arcs the words were spread over"""
    td = str(tmp_path); mdl, wb = write_model(td)
    lat = make_lattice(7, reorder=True, segments=5, alts=1, p_sil=0.6)
    open(f"{td}/one.txt", "w").write(lc.lattice_text("one", lat))
    r = subprocess.run([TOOL, "word-align", mdl, wb, f"ark,t:{td}/one.txt", f"{td}/o.txt", "1", "0", "0"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    out = open(f"{td}/o.txt").read(); words = [int(x) for x in out.split("\nwords")[1].split("\n")[0].split()]; tv = [float(x) for x in out.split("\ntimes")[1].split("\n")[0].split()]
    # truth from the construction: walk the single path, group frames by phone, words = maximal runs begin..end / singleton
    by_src = {a[0]: a for a in lat["arcs"]}; s = 0; frames = []
    while s in by_src: a = by_src[s]; frames.append((a[2], a[3])); s = a[1]
    phones = [(t + 1) // 2 for t, _ in frames]; spans = []; i = 0
    while i < len(frames):
        p = phones[i]; j = i
        if p in SIL:
            while j < len(frames) and phones[j] == p and not (j > i and frames[j][0] == 2 * p): j += 1
            i = max(j, i + 1); continue
        # a word: up to the end of its END / ONE phone
        j = i
        while True:
            q = phones[j]; k = j + 1
            while k < len(frames) and phones[k] == q and frames[k][0] == 2 * q - 1: k += 1      # (reorder: the leaving transition first, then self-loops)
            j = k
            if q in END or q in ONE: break
        spans.append((i, j)); i = j
    assert len(words) == len(spans) and list(zip(tv[0::2], tv[1::2])) == [(float(b), float(e)) for b, e in spans], (words, tv, spans)

def test_postprocessor_config_with_word_boundary_file(tmp_path):
    """--word-boundary-rxfilename in a lattice post-processor config (cudadecoder/lattice-postprocessor.h:35-71) through libk3host's CTM entry point"""
    import ctypes
    from kaldi_amd import hostlib
    L = hostlib.load(); td = str(tmp_path); mdl, wb = write_model(td)
    lat = make_lattice(3, reorder=True, segments=4, alts=2)
    open(f"{td}/in.txt", "w").write(lc.lattice_text("utt", lat)); open(f"{td}/pp.conf", "w").write(f"--word-boundary-rxfilename={wb}\n--acoustic-scale=1.0\n")
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.k3h_lattice_table_to_ctm_model(f"ark,t:{td}/in.txt".encode(), f"{td}/pp.conf".encode(), ctypes.c_float(0.03), mdl.encode(), buf, len(buf))
    assert n > 0, hostlib.last_error()
    lines = buf.value.decode().splitlines(); assert lines and all(l.startswith("utt 0  ") for l in lines)
    # without the model the same config is refused loudly (the reference asserts SetTransitionModel was called)
    assert L.k3h_lattice_table_to_ctm(f"ark,t:{td}/in.txt".encode(), f"{td}/pp.conf".encode(), ctypes.c_float(0.03), buf, len(buf)) < 0 and "SetTransitionInformation" in hostlib.last_error()
