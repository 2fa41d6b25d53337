"""Oracle fixtures of the next scope row (online i-vector extraction): tests/golden/ivector holds a small extractor made
with the reference's own tools and the i-vectors the reference's ivector-extract-online2 (OnlineIvectorFeature) computes with it
(tests/golden/make_golden_ivector.py).  No GPU code consumes them yet; this test keeps the fixture honest: the committed model files,
fed to the reference binary again, reproduce the committed i-vectors."""
import os, subprocess, numpy as np, pytest
from oracle import kaldi_io as kio
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "golden", "ivector"); EXE = os.path.join(ROOT, "oracle", "_ref", "bin", "ivector-extract-online2")

def test_fixture_is_complete():
    g = np.load(os.path.join(DIR, "ivector_golden.npz"))
    utts = sorted(k[5:] for k in g.files if k.startswith("feat_"))
    assert utts == ["utt0", "utt1", "utt2", "utt3"]
    for u in utts:
        T = g["feat_" + u].shape[0]
        assert g["iv_default_" + u].shape == ((T + 9) // 10, 16) and g["iv_repeat_" + u].shape == (T, 16)      # one i-vector per --ivector-period=10 frames
        assert np.isfinite(g["iv_default_" + u]).all()
    for f in ("final.dubm", "final.ie", "final.mat", "global_cmvn.stats", "splice.conf", "online_cmvn.conf", "ivector_extractor.conf"): assert os.path.exists(os.path.join(DIR, f))

@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref not built (needs /root/reference)")
def test_reference_binary_reproduces_the_golden_ivectors(tmp_path):
    g = np.load(os.path.join(DIR, "ivector_golden.npz")); td = str(tmp_path)
    feats = {k[5:]: g[k] for k in g.files if k.startswith("feat_")}
    kio.write_ark(f"{td}/feats.ark", feats)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
    r = subprocess.run([EXE, "--config=ivector_extractor.conf", "ark:spk2utt", f"ark:{td}/feats.ark", f"ark:{td}/iv.ark"], capture_output=True, text=True, env=env, cwd=DIR)
    assert r.returncode == 0, r.stderr[-2000:]
    iv = kio.read_ark(f"{td}/iv.ark")
    for u in feats: assert np.array_equal(iv[u], g["iv_default_" + u]), u


def _numbers(text):
    import re
    return np.array([float(x) for x in re.findall(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?", text)])

def test_host_readers_of_the_model_files_match_the_reference_dumps():
    """kaldi_amd/host's readers of the binary DiagGmm / IvectorExtractor files against the text dumps the REFERENCE's gmm-global-copy and
    ivector-extractor-copy made of the same models (final.*.txt in the fixture; text prints ~7 significant digits)"""
    import __graft_entry__ as ge
    ge.build()
    tool = os.path.join(ROOT, "kaldi_amd", "bin", "k3-host-tool")
    r = subprocess.run([tool, "gmm-dump", os.path.join(DIR, "final.dubm")], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    mine = {l.split()[0]: _numbers(l.split(" ", 1)[1]) if " " in l else np.zeros(0) for l in r.stdout.splitlines()[1:]}
    ref = open(os.path.join(DIR, "final.dubm.txt")).read()
    sect = lambda a, b: _numbers(ref.split(a, 1)[1].split(b, 1)[0])
    for name, a, b in (("gconsts", "<GCONSTS>", "<WEIGHTS>"), ("weights", "<WEIGHTS>", "<MEANS_INVVARS>"), ("means_invvars", "<MEANS_INVVARS>", "<INV_VARS>"), ("inv_vars", "<INV_VARS>", "</DiagGMM>")):
        assert mine[name].shape == sect(a, b).shape and np.allclose(mine[name], sect(a, b), rtol=2e-6, atol=1e-9), name
    assert r.stdout.splitlines()[0] == "num_gauss 32 dim 20"
    r = subprocess.run([tool, "ie-dump", os.path.join(DIR, "final.ie")], capture_output=True, text=True); assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines()[0].startswith("num_gauss 32 feat_dim 20 ivector_dim 16 w 0x0 prior_offset 100")
    mine = {l.split()[0]: _numbers(l.split(" ", 1)[1]) if " " in l else np.zeros(0) for l in r.stdout.splitlines()[1:]}
    ref = open(os.path.join(DIR, "final.ie.txt")).read()
    assert np.allclose(mine["w_vec"], _numbers(ref.split("<w_vec>", 1)[1].split("<M>", 1)[0]), rtol=2e-6)
    M = _numbers(ref.split("<M>", 1)[1].split("<SigmaInv>", 1)[0])[1:]                     # the first number is the count of matrices
    assert mine["M"].shape == M.shape == (32 * 20 * 16,) and np.allclose(mine["M"], M, rtol=2e-6, atol=1e-9)
    S = _numbers(ref.split("<SigmaInv>", 1)[1].split("<IvectorOffset>", 1)[0])
    assert mine["sigma_inv"].shape == S.shape == (32 * 20 * 21 // 2,) and np.allclose(mine["sigma_inv"], S, rtol=2e-6, atol=1e-9)
    # error behaviour: a text-format model is refused with a message, a truncated file too
    assert subprocess.run([tool, "gmm-dump", os.path.join(DIR, "final.dubm.txt")], capture_output=True).returncode != 0


def test_numpy_restatement_matches_the_reference_ivectors():
    """oracle/ivector_oracle.py (the algorithm written out: posteriors, statistics, conjugate-gradient solution, schedule) against the
    reference binary's output.  Tolerance 5e-5 absolute (measured 8e-6) on i-vector entries of magnitude ~0.2: the reference sums float32 feature
    statistics through BLAS in a different order, and 15 CG iterations amplify the last bits."""
    from oracle import ivector_oracle as io
    g = np.load(os.path.join(DIR, "ivector_golden.npz"))
    ubm, ie = io.read_models(os.path.join(DIR, "final.dubm.txt"), os.path.join(DIR, "final.ie.txt"))
    lda = io.re.sub(r"[\[\]]", " ", open(os.path.join(DIR, "final.mat")).read()); lda = np.array(lda.split(), np.float64).reshape(20, -1).astype(np.float32)
    st = np.array(io.re.sub(r"[\[\]]", " ", open(os.path.join(DIR, "global_cmvn.stats")).read()).split(), np.float64).reshape(2, -1)
    for u in ("utt0", "utt1", "utt2", "utt3"):
        mine = io.extract_online(g["feat_" + u], ubm, ie, lda, st, max_count=100.0)
        ref = g["iv_default_" + u]
        assert mine.shape == ref.shape
        assert np.abs(mine - ref).max() <= 5e-5, (u, np.abs(mine - ref).max())          # measured: <= 8e-6
    rep = io.extract_online(g["feat_utt3"], ubm, ie, lda, st, max_count=100.0, repeat=True)
    assert np.abs(rep - g["iv_repeat_utt3"]).max() <= 5e-5
