"""Pins oracle/feat_oracle.c (the CPU restatement) against (a) the reference's HTK golden vectors and
(b) outputs of the reference's own binaries (tests/golden/feat_golden.npz).  CPU only."""
import os, numpy as np, pytest
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


def test_random_option_sets_against_the_reference_binaries(tmp_path):
    """fuzz (live only): all window types, frame lengths / shifts, pre-emphasis, dc removal, snip-edges, bin counts and band edges, the energy
    options, htk-compat, log / power switches, cepstra and lifter drawn at random; oracle vs the reference's compute-{fbank,mfcc}-feats
    (a 60-configuration run of this loop: worst difference 8e-6 of the value range)"""
    import os, subprocess
    from oracle import feat_oracle as fo, kaldi_io as kio
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); bindir = os.path.join(root, "oracle", "_ref", "bin")
    if not os.path.exists(os.path.join(bindir, "compute-fbank-feats")): pytest.skip("oracle/_ref not built (needs /root/reference)")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(root, "oracle", "_ref", "mkl"), MKL_THREADING_LAYER="SEQUENTIAL")
    rng = np.random.default_rng(6); td = str(tmp_path); B = lambda v: "true" if v else "false"; checked = 0
    for it in range(14):
        kind = ("fbank", "mfcc")[it % 2]
        wav = np.clip(np.rint(rng.normal(0, 3000, int(rng.integers(500, 20000)))), -32768, 32767).astype(np.int16)
        kio.write_wav(f"{td}/u.wav", wav); open(f"{td}/u.scp", "w").write(f"u {td}/u.wav\n")
        kw = dict(dither=0.0, window_type=str(rng.choice(["hamming", "hanning", "povey", "rectangular", "blackman", "sine"])), frame_length_ms=float(rng.choice([25.0, 20.0, 30.0, 10.0])),
                  frame_shift_ms=float(rng.choice([10.0, 5.0, 12.5])), preemph_coeff=float(rng.choice([0.97, 0.0, 0.5])), remove_dc_offset=int(rng.integers(0, 2)), snip_edges=int(rng.integers(0, 2)),
                  num_bins=int(rng.choice([23, 40, 64, 80])), low_freq=float(rng.choice([20.0, 0.0, 125.0])), high_freq=float(rng.choice([0.0, -400.0, 7600.0])), use_energy=int(rng.integers(0, 2)),
                  raw_energy=int(rng.integers(0, 2)), energy_floor=float(rng.choice([0.0, 1.0])), htk_compat=int(rng.integers(0, 2)))
        flags = ["--dither=0", "--window-type=" + kw["window_type"], "--frame-length=%g" % kw["frame_length_ms"], "--frame-shift=%g" % kw["frame_shift_ms"], "--preemphasis-coefficient=%g" % kw["preemph_coeff"],
                 "--remove-dc-offset=" + B(kw["remove_dc_offset"]), "--snip-edges=" + B(kw["snip_edges"]), "--num-mel-bins=%d" % kw["num_bins"], "--low-freq=%g" % kw["low_freq"], "--high-freq=%g" % kw["high_freq"],
                 "--use-energy=" + B(kw["use_energy"]), "--raw-energy=" + B(kw["raw_energy"]), "--energy-floor=%g" % kw["energy_floor"], "--htk-compat=" + B(kw["htk_compat"])]
        if kind == "fbank":
            kw.update(use_log_fbank=int(rng.integers(0, 2)), use_power=int(rng.integers(0, 2))); flags += ["--use-log-fbank=" + B(kw["use_log_fbank"]), "--use-power=" + B(kw["use_power"])]
            opts = fo.fbank_opts(**kw)
        else:
            nc = min(int(rng.choice([13, 20, kw["num_bins"]])), kw["num_bins"]); kw.update(num_ceps=nc, cepstral_lifter=float(rng.choice([22.0, 0.0])))
            flags += ["--num-ceps=%d" % nc, "--cepstral-lifter=%g" % kw["cepstral_lifter"]]; opts = fo.mfcc_opts(**kw)
        r = subprocess.run([f"{bindir}/compute-{kind}-feats"] + flags + [f"scp:{td}/u.scp", f"ark:{td}/o.ark"], capture_output=True, text=True, env=env)
        if r.returncode != 0: continue                      # an utterance too short for one frame etc.: nothing to compare
        ref = kio.read_ark(f"{td}/o.ark").get("u")
        if ref is None: continue
        mine = fo.compute_features(wav.astype(np.float32), opts)
        assert mine.shape == ref.shape, (kind, kw)
        fin = np.isfinite(ref); assert np.array_equal(np.isfinite(mine), fin)
        assert np.abs(mine[fin] - ref[fin]).max() <= 3e-5 * max(1.0, np.abs(ref[fin]).max()), (kind, kw, np.abs(mine[fin] - ref[fin]).max())
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize("name", sorted(fc.FFTSIZE_CASES))
def test_oracle_vs_reference_binary_on_256_and_1024_point_windows(name):
    """8 kHz / 25 ms (256-sample padded window), 32 kHz / 25 ms and 16 kHz / 40-50 ms (1024): tests/golden/make_golden_feat_fftsizes.py"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "feat_fftsizes_golden.npz"))
    kind, kw, rate, nsamp, seed = fc.FFTSIZE_CASES[name]
    got = fo.compute_features(g["wav_" + name].astype(np.float32), _mk(kind, kw)); ref = g["ref_" + name]
    assert got.shape == ref.shape
    tol = 5e-5 if kind == "fbank" else 2e-4
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()

@pytest.mark.parametrize("name", sorted(fc.REF_CASES))
def test_f64path_yardstick_is_pinned_three_ways(feat_golden, name):
    """the GPU kernel's primary gate is its distance to oracle.feat_oracle.compute_features_f64path ("the exact value of the reference's formulas"), so
    that yardstick must not be self-referential.  It is (a) the SAME source as the float32 restatement (oracle/feat_oracle_path.inc compiled with REAL = double),
    whose float32 instantiation is pinned to the HTK vectors and to the reference binaries above; here additionally (b) within table rounding of the committed
    all-float64 evaluation tests/golden/feat_truth64.npz (tables in double as well: another code path of the restatement) and (c) no further from the reference
    BINARY's committed output than the float32 rounding noise measured when the fixture was made (fbank <= 1.1e-4, lifted cepstra <= 4e-4)."""
    kind, kw, wkey = fc.REF_CASES[name]
    o = fo.mfcc_opts(**kw) if kind == "mfcc" else fo.fbank_opts(**kw)
    w = feat_golden[wkey].astype(np.float32)
    exact = fo.compute_features_f64path(w, o)
    t64 = np.load(os.path.join(os.path.dirname(__file__), "golden", "feat_truth64.npz"))["truth64_" + name]
    assert exact.shape == t64.shape == feat_golden["ref_" + name].shape
    assert np.abs(exact - t64).max() <= (3e-5 if kind == "fbank" else 2e-4), np.abs(exact - t64).max()      # float32 vs float64 TABLES only
    assert np.abs(exact - feat_golden["ref_" + name]).max() <= (1.1e-4 if kind == "fbank" else 4e-4)
    assert np.abs(exact - fo.compute_features(w, o)).max() <= (1.1e-4 if kind == "fbank" else 4.5e-4)
