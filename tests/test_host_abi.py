"""CPU tests of the host-tail C ABI (include/k3host.h -> kaldi_amd/lib/libk3host.so) and its Python front-end (kaldi_amd/lattice.py):
symbols, agreement with the command-line programs built on the same code (which are pinned to the reference's determinizer source in
tests/test_lattice_det.py), error behaviour."""
import ctypes, os, re, subprocess, numpy as np, pytest
from tests import lattice_cases as lc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kaldi_amd", "bin")

@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()

def _raw(lat):
    """dict lattice of tests/lattice_cases -> kaldi_amd.lattice.RawLattice (state = index, frame 0 only for the start state)"""
    from kaldi_amd.lattice import RawLattice
    n = lat["n"]; fin = np.full(n, np.inf, np.float32)
    for s, (g, a) in lat["finals"].items(): fin[s] = g
    a = lat["arcs"]; col = lambda i, dt: np.array([x[i] for x in a], dt)
    frame = np.ones(n, np.int32); frame[lat["start"]] = 0
    return RawLattice(frame, np.arange(n, dtype=np.int32), fin, col(0, np.int32), col(1, np.int32), col(2, np.int32), col(3, np.int32), col(4, np.float32), col(5, np.float32), lat["start"])

def test_library_exports_all_declared_symbols():
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "k3host.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(k3h_[a-z0-9_]+)\s*\(", src)))
    from kaldi_amd import hostlib
    L = ctypes.CDLL(hostlib.LIB_PATH)
    assert len(names) >= 12 and not [n for n in names if not hasattr(L, n)]

def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from kaldi_amd import hostlib
    monkeypatch.setattr(hostlib, "LIB_PATH", str(tmp_path / "nope.so")); monkeypatch.setattr(hostlib, "_lib", None)
    with pytest.raises(hostlib.K3HostError): hostlib.load()

@pytest.mark.parametrize("mode", ["word", "phone", "phone_minimize"])
def test_python_determinization_equals_the_programs(mode, tmp_path):
    from kaldi_amd import lattice as kl, synth
    td = str(tmp_path); mdl = f"{td}/final.mdl"
    synth.make_tdnn(seed=1, dim=32, num_pdfs=20).write(mdl, as_mdl=True, num_pdfs=20, left_context=2, right_context=2)
    trans = kl.TransitionInformation(mdl); assert trans.NumTransitionIds() == 40
    lats = [lc.random_lattice(s, frames=5 + s % 4, width=3 + s % 3, words=2 + s % 3, tids=40) for s in range(10)]
    open(f"{td}/in.txt", "w").write("".join(lc.lattice_text("k%d" % i, l) for i, l in enumerate(lats)))
    exe = os.path.join(BIN, "lattice-determinize-pruned" if mode == "word" else "lattice-determinize-phone-pruned")
    extra = ["--minimize=true"] if mode == "phone_minimize" else []
    r = subprocess.run([exe, "--beam=4", "--acoustic-scale=0.5"] + extra + ([mdl] if mode != "word" else []) + [f"ark,t:{td}/in.txt", f"ark,t:{td}/cli.txt"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cli = lc.parse_compact_text(open(f"{td}/cli.txt").read())
    for i, l in enumerate(lats):
        raw = _raw(l); raw.arc_ac = (raw.arc_ac.astype(np.float64) * np.float32(0.5)).astype(np.float32)          # the programs scale, determinize, scale back
        c, ok = kl.DeterminizeLatticePhonePruned(raw, 4.0, trans if mode != "word" else None, minimize=(mode == "phone_minimize"))
        assert ok
        c.ScaleAcoustic(1.0 / np.float32(0.5)); c.Write("k%d" % i, f"ark,t:{td}/py.txt")
        mine = lc.parse_compact_text(open(f"{td}/py.txt").read())["k%d" % i]
        # the programs sort the result topologically before writing; compare up to that renumbering
        assert lc.enumerate_compact(mine) == lc.enumerate_compact(cli["k%d" % i])
        assert len(mine["arcs"]) == len(cli["k%d" % i]["arcs"]) == c.num_arcs
        bp = c.best_path()
        if bp is not None:
            words = min(lc.enumerate_compact(mine).items(), key=lambda kv: kv[1][0][0])[0]
            assert tuple(bp[1]) == words

def test_convert_lattice_and_errors(tmp_path):
    from kaldi_amd import lattice as kl, hostlib
    td = str(tmp_path); lat = lc.random_lattice(703, frames=7, width=3, words=3, p_word=0.2)
    raw = _raw(lat).connect()
    c = kl.ConvertLattice(raw); c.Write("u", f"ark,t:{td}/py.txt")
    open(f"{td}/in.txt", "w").write(lc.lattice_text("u", lat))
    assert subprocess.run([os.path.join(BIN, "k3-host-tool"), "convert-lattice", f"ark,t:{td}/in.txt", f"ark,t:{td}/cli.txt"], capture_output=True).returncode == 0
    assert open(f"{td}/py.txt").read() == open(f"{td}/cli.txt").read()
    c.Write("u", f"ark:{td}/py.ark"); assert list(lc.parse_compact_binary(open(f"{td}/py.ark", "rb").read())) == ["u"]
    with pytest.raises(hostlib.K3HostError): kl.TransitionInformation(f"{td}/missing.mdl")
    bad = _raw(dict(start=0, n=2, finals={1: (0.0, 0.0)}, arcs=[(0, 1, 1, 1, 1.0, 1.0), (1, 0, 0, 0, 1.0, 0.0)]))
    with pytest.raises(hostlib.K3HostError, match="Topological sorting"): kl.DeterminizeLatticePhonePruned(bad, 5.0)
    with pytest.raises(hostlib.K3HostError): kl.DeterminizeLatticePhonePruned(_raw(lat), -1.0)


def test_determinization_is_thread_safe(tmp_path):
    """the reference calls its determinizer from pool threads (one lattice each); the C ABI must allow the same: 8 Python threads (ctypes
    releases the GIL) determinize different lattices concurrently with one shared transition map, results equal the serial ones"""
    import threading
    from kaldi_amd import lattice as kl, synth
    mdl = str(tmp_path / "final.mdl"); synth.make_tdnn(seed=1, dim=32, num_pdfs=20).write(mdl, as_mdl=True, num_pdfs=20, left_context=2, right_context=2)
    trans = kl.TransitionInformation(mdl)
    raws = [_raw(lc.random_lattice(300 + s, frames=20, width=4, words=3, tids=40)) for s in range(16)]
    def sig(c): return (c.num_states, c.num_arcs, c.arc_label.tolist(), c.arc_graph.tolist(), c.strings.tolist())
    serial = [sig(kl.DeterminizeLatticePhonePruned(r, 5.0, trans)[0]) for r in raws]
    out = [None] * len(raws); errs = []
    def work(i):
        try:
            for _ in range(3): out[i] = sig(kl.DeterminizeLatticePhonePruned(raws[i], 5.0, trans)[0])
        except Exception as e: errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(raws))]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs and out == serial
