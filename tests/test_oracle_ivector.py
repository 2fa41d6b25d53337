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
