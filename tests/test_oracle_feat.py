"""Pins oracle/feat_oracle.c (the CPU restatement) against (a) the reference's HTK golden vectors and
(b) outputs of the reference's own binaries (tests/golden/feat_golden.npz).  CPU only."""
import numpy as np, pytest
from oracle import feat_oracle as fo
from tests import feat_cases as fc

def _mk(kind, kw):
    return fo.mfcc_opts(**kw) if kind == "mfcc" else fo.fbank_opts(**kw)

@pytest.mark.parametrize("idx", [1, 2, 3, 4])
def test_oracle_fbank_vs_htk(feat_golden, idx):
    kw, tol = fc.HTK_FBANK[idx]
    got = fo.compute_features(feat_golden["wav"].astype(np.float32), fo.fbank_opts(**kw))
    ref = feat_golden[f"htk_fbank_{idx}"]
    assert got.shape == ref.shape
    d = np.abs(got[10:-10] - ref[10:-10])
    if idx == 3: d = d[:, :20]   # feature-fbank-test.cc:334 "We know the last couple of filterbanks differ"
    assert d.max() <= tol

@pytest.mark.parametrize("idx", [1, 2, 3, 4, 5, 6])
def test_oracle_mfcc_vs_htk(feat_golden, idx):
    kw = dict(fc.HTK_MFCC[idx])
    got = fo.compute_features(feat_golden["wav"].astype(np.float32), fo.mfcc_opts(**kw))
    ref = feat_golden[f"htk_mfcc_{idx}"][:, :13]   # HTK file = statics + deltas; statics only here
    assert got.shape == ref.shape
    assert np.abs(got[10:-10] - ref[10:-10]).max() <= 1e-3     # reference test allows 1.0 (feature-mfcc-test.cc:164); we are at ~4e-4

@pytest.mark.parametrize("name", sorted(fc.REF_CASES))
def test_oracle_vs_reference_binary(feat_golden, name):
    kind, kw, wkey = fc.REF_CASES[name]
    got = fo.compute_features(feat_golden[wkey].astype(np.float32), _mk(kind, kw))
    ref = feat_golden["ref_" + name]
    assert got.shape == ref.shape
    tol = 5e-5 if kind == 'fbank' else 2e-4   # MFCC values reach ~150 (ulp 1.5e-5) and sum 40 terms
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()

@pytest.mark.parametrize("nv", [0, 1])
def test_oracle_cmvn_vs_reference_binary(feat_golden, nv):
    got = fo.cmvn_offline(feat_golden["ref_fbank_default40"], norm_vars=bool(nv))
    ref = feat_golden[f"ref_cmvn_normvars{nv}"]
    assert np.abs(got - ref).max() <= 2e-6

def test_oracle_num_frames_edges():
    o = fo.fbank_opts()
    assert fo.num_frames(399, o) == 0 and fo.num_frames(400, o) == 1 and fo.num_frames(559, o) == 1 and fo.num_frames(560, o) == 2
    o2 = fo.fbank_opts(snip_edges=0)
    assert fo.num_frames(160000, o2) == 1000 and fo.num_frames(79, o2) == 0 and fo.num_frames(80, o2) == 1
    assert fo.compute_features(np.zeros(100, np.float32), o).shape == (0, 23)

# ---- online CMVN (feat/online-feature.cc OnlineCmvn) vs the reference's apply-cmvn-online (tests/golden/cmvn_online_golden.npz)
from tests import cmvn_cases as cc

@pytest.mark.parametrize("name", sorted(cc.CASES))
def test_oracle_cmvn_online_vs_reference_binary(cmvn_online_golden, name):
    g = cmvn_online_golden
    for utt, feats, gstats, spk, kw in cc.runs(g, name):
        got = fo.cmvn_online(feats, gstats, speaker_stats=spk, **kw)
        ref = g[f"ref_{name}_{utt}"]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-6, (name, utt, np.abs(got - ref).max())
        if not kw.get("norm_vars"): assert np.array_equal(got, ref)          # mean-only: the double-precision recursion is reproduced exactly

def test_oracle_cmvn_online_rejects_what_the_reference_rejects(cmvn_online_golden):
    g = cmvn_online_golden
    with pytest.raises(ValueError): fo.cmvn_online(g["feats_a"], g["global"], norm_means=False, norm_vars=True)
    bad = g["global"].copy(); bad[0, -1] = 0.0
    with pytest.raises(ValueError): fo.cmvn_online(g["feats_a"], bad)
    with pytest.raises(ValueError): fo.cmvn_online(g["feats_a"], g["global"], cmn_window=10, speaker_frames=20, global_frames=5)
