"""OfflineFeatureTpl::ComputeFeatures' resampling branch (feat/feature-common-inl.h:29-57 -> ResampleWaveform, feat/resample.cc:363-372).  CPU: the numpy restatement of
LinearResample followed by the pinned feature oracle against the REFERENCE binaries' output with --allow-downsample / --allow-upsample (tests/golden/feat_resample_golden.npz,
made by tests/golden/make_golden_feat_resample.py).  GPU: k3_resample_batch against the restatement sample by sample, and the drop-in program with the same flags against the golden features."""
import os, subprocess, numpy as np, pytest
from tests import feat_cases as fc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "feat_resample_golden.npz"))

def _opts(fo, kind, kw): return (fo.fbank_opts if kind == "fbank" else fo.mfcc_opts)(**kw)

def _close(got, ref, kind):
    """north_star's 1e-4 (3e-4 for MFCC, tests/test_feat_gpu.py) -- except in mel bins that hold only what the resampler's low-pass lets through above its cutoff (upsampling:
    the band above the old Nyquist is ~1e-6 of the passband's power, 9+ nats below it): there the log amplifies the float32 rounding of the resampler's dot products, whose
    summation order inside the reference (cblas_sdot) is not defined; those bins are held to 1e-3."""
    d = np.abs(got - ref)
    if kind != "fbank": return got.shape == ref.shape and d.max() <= 3e-4
    tol = np.where(ref.mean(0) > ref.mean(0).max() - 9.0, 1e-4, 1e-3)
    return got.shape == ref.shape and bool((d.max(0) <= tol).all())

@pytest.mark.parametrize("name", list(fc.RESAMPLE_CASES))
def test_oracle_resampling_then_features_equal_the_reference_binary(name):
    from oracle import feat_oracle as fo
    kind, kw, rate, nsamp, seed = fc.RESAMPLE_CASES[name]
    wav = G["wav_" + name].astype(np.float32); assert wav.size == nsamp
    res = fo.resample_waveform(wav, rate, kw["samp_freq"])
    got = fo.compute_features(res, _opts(fo, kind, kw)); ref = G["ref_" + name]
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert _close(got, ref, kind), np.abs(got - ref).max(0)

def test_number_of_output_samples_and_length_edge_cases():
    from oracle import feat_oracle as fo
    assert fo.resample_waveform(np.zeros(0, np.float32), 16000, 8000).size == 0
    assert fo.resample_waveform(np.ones(1, np.float32), 16000, 8000).size == 1          # GetNumOutputSamples: the interval [0, 1/16000) holds output sample 0
    assert fo.resample_waveform(np.ones(160, np.float32), 16000, 8000).size == 80 and fo.resample_waveform(np.ones(161, np.float32), 16000, 8000).size == 81
    assert fo.resample_waveform(np.ones(441, np.float32), 44100, 16000).size == 160
    x = fo.resample_waveform(np.full(4000, 1000.0, np.float32), 16000, 8000); assert np.abs(x[100:-100] - 1000.0).max() < 15.0      # DC passes the low-pass (ripple of the six-zero window)

@pytest.mark.gpu
@pytest.mark.parametrize("name", list(fc.RESAMPLE_CASES))
def test_hip_resampling_equals_the_restatement(name):
    import torch
    from kaldi_amd import feat
    from oracle import feat_oracle as fo
    kind, kw, rate, nsamp, seed = fc.RESAMPLE_CASES[name]
    wav = G["wav_" + name].astype(np.float32); extra = np.random.default_rng(seed + 1).normal(0, 3000, 777).astype(np.float32)
    both = torch.from_numpy(np.concatenate([wav, extra, wav[:1], wav[:0]])).cuda()           # a ragged batch: the file, a second one, one sample, none
    out, nl = feat.ResampleWaveform(both, [wav.size, extra.size, 1, 0], rate, kw["samp_freq"]); out = out.cpu().numpy()
    off = np.concatenate([[0], np.cumsum(nl)])
    for u, src in enumerate((wav, extra, wav[:1], wav[:0])):
        want = fo.resample_waveform(src, rate, kw["samp_freq"]); got = out[off[u]:off[u + 1]]
        assert got.shape == want.shape and (want.size == 0 or np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())), (name, u)      # (float32 sums; the kernel may contract multiply-adds)

@pytest.mark.gpu
@pytest.mark.parametrize("name", list(fc.RESAMPLE_CASES))
def test_feature_program_resamples_like_the_reference_binary(name, tmp_path):
    from oracle import kaldi_io as kio
    kind, kw, rate, nsamp, seed = fc.RESAMPLE_CASES[name]
    exe = os.path.join(ROOT, "kaldi_amd", "bin", f"compute-{kind}-feats-cuda")
    if not os.path.exists(exe): pytest.fail(f"{exe} is missing: run __graft_entry__.build()")
    kio.write_wav(str(tmp_path / "a.wav"), G["wav_" + name], rate=rate); open(tmp_path / "a.scp", "w").write(f"u {tmp_path}/a.wav\n")
    flags = [f"--sample-frequency={kw['samp_freq']}", "--dither=0"] + ([f"--num-mel-bins={kw['num_bins']}"] if "num_bins" in kw else []) + (["--snip-edges=false"] if kw.get("snip_edges", 1) == 0 else [])
    r = subprocess.run([exe] + flags + [f"scp:{tmp_path}/a.scp", f"ark:{tmp_path}/o.ark"], capture_output=True, text=True)      # without the flag: the file is skipped with the reference's message
    assert r.returncode == 1 and "sample Frequency mismatch" in r.stderr, r.stderr[-800:]
    r = subprocess.run([exe, "--allow-downsample=true", "--allow-upsample=true"] + flags + [f"scp:{tmp_path}/a.scp", f"ark:{tmp_path}/o.ark"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    got = kio.read_ark(str(tmp_path / "o.ark"))["u"]; ref = G["ref_" + name]
    assert _close(got, ref, kind), np.abs(got - ref).max(0)
