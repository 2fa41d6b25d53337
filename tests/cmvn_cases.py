"""Online-CMVN cases shared by the oracle (CPU) and HIP (GPU) tests; the option sets are the command lines of
tests/golden/make_golden_cmvn_online.py."""
import numpy as np

CASES = {   # name -> (kwargs, feature set, spk2utt?)
    "default": (dict(), "ab", False),
    "w100": (dict(cmn_window=100, speaker_frames=60, global_frames=25), "ab", False),
    "w100_vars": (dict(cmn_window=100, speaker_frames=60, global_frames=25, norm_vars=True), "ab", False),
    "w100_spk": (dict(cmn_window=100, speaker_frames=60, global_frames=25), "ab", True),
    "w100_spk_vars_skip": (dict(cmn_window=100, speaker_frames=100, global_frames=10, norm_vars=True, skip_dims=(0, 5)), "ab", True),
    "nomeans": (dict(norm_means=False), "ab", False),
    "fbank40": (dict(cmn_window=50, speaker_frames=50, global_frames=20), "c", False),
}

def speaker_stats_after(feats):
    """OnlineCmvn::GetState (feat/online-feature.cc:470-486): sums, sums of squares (of the double values) and count over the utterance."""
    x = feats.astype(np.float64); dim = x.shape[1]; st = np.zeros((2, dim + 1))
    for t in range(x.shape[0]): st[0, :dim] += x[t]; st[1, :dim] += x[t] * x[t]; st[0, dim] += 1.0
    return st

def runs(g, name):
    """yields (utt, feats, global_stats, speaker_stats or None, kwargs) in the order apply-cmvn-online processes them"""
    kw, fs, spk = CASES[name]
    if fs == "c":
        yield "utt_c", g["feats_c"], g["global40"], None, kw
        return
    yield "utt_a", g["feats_a"], g["global"], None, kw
    yield "utt_b", g["feats_b"], g["global"], (speaker_stats_after(g["feats_a"]) if spk else None), kw
