"""ctypes front-end of oracle/feat_oracle.c (TEST INFRASTRUCTURE; see that file's header)."""
import ctypes, os, numpy as np
from . import build as _build
HERE = os.path.dirname(os.path.abspath(__file__))

WINDOW_TYPES = {"hanning": 0, "sine": 1, "hamming": 2, "povey": 3, "rectangular": 4, "blackman": 5}

class FeatOpts(ctypes.Structure):
    """Field order == k3o_feat_opts == k3_feat_opts (include/k3hip.h)."""
    _fields_ = [("samp_freq", ctypes.c_float), ("frame_shift_ms", ctypes.c_float), ("frame_length_ms", ctypes.c_float),
                ("dither", ctypes.c_float), ("preemph_coeff", ctypes.c_float), ("blackman_coeff", ctypes.c_float),
                ("remove_dc_offset", ctypes.c_int32), ("round_to_power_of_two", ctypes.c_int32), ("snip_edges", ctypes.c_int32),
                ("window_type", ctypes.c_int32), ("num_bins", ctypes.c_int32),
                ("low_freq", ctypes.c_float), ("high_freq", ctypes.c_float), ("vtln_low", ctypes.c_float), ("vtln_high", ctypes.c_float),
                ("htk_mode", ctypes.c_int32), ("use_energy", ctypes.c_int32), ("energy_floor", ctypes.c_float),
                ("raw_energy", ctypes.c_int32), ("htk_compat", ctypes.c_int32), ("use_log_fbank", ctypes.c_int32), ("use_power", ctypes.c_int32),
                ("num_ceps", ctypes.c_int32), ("cepstral_lifter", ctypes.c_float), ("feature_type", ctypes.c_int32), ("vtln_warp", ctypes.c_float)]

def fbank_opts(**kw):
    """FbankOptions defaults: feat/feature-fbank.h:44-61, feature-window.h:53-66, mel-computations.h:56-58."""
    o = FeatOpts(16000.0, 10.0, 25.0, 1.0, 0.97, 0.42, 1, 1, 1, 3, 23, 20.0, 0.0, 100.0, -500.0, 0, 0, 0.0, 1, 0, 1, 1, 13, 22.0, 0, 1.0)
    for k, v in kw.items():
        setattr(o, k, WINDOW_TYPES[v] if k == "window_type" and isinstance(v, str) else v)
    return o

def mfcc_opts(**kw):
    """MfccOptions defaults: feat/feature-mfcc.h:40-60."""
    o = fbank_opts(use_energy=1, feature_type=1)
    for k, v in kw.items():
        setattr(o, k, WINDOW_TYPES[v] if k == "window_type" and isinstance(v, str) else v)
    return o

_lib = None
def lib():
    global _lib
    if _lib is None:
        _build.build(with_ref=False)
        _lib = ctypes.CDLL(os.path.join(HERE, "libk3oracle_feat.so"))
        _lib.k3o_num_frames.restype = ctypes.c_int32
        _lib.k3o_num_frames.argtypes = [ctypes.c_int64, ctypes.POINTER(FeatOpts)]
        _lib.k3o_feat_dim.argtypes = [ctypes.POINTER(FeatOpts)]
        _lib.k3o_compute_features.argtypes = [ctypes.POINTER(FeatOpts), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        _lib.k3o_compute_features_f64path.argtypes = [ctypes.POINTER(FeatOpts), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        _lib.k3o_cmvn_offline.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        _lib.k3o_cmvn_online.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 7 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
    return _lib

def num_frames(nsamp, opts):
    return lib().k3o_num_frames(int(nsamp), ctypes.byref(opts))

def compute_features(wave, opts):
    wave = np.ascontiguousarray(wave, dtype=np.float32)
    T = num_frames(len(wave), opts); dim = lib().k3o_feat_dim(ctypes.byref(opts))
    out = np.zeros((max(T, 0), dim), dtype=np.float32)
    if T > 0:
        r = lib().k3o_compute_features(ctypes.byref(opts), wave.ctypes.data, len(wave), out.ctypes.data)
        assert r == T, r
    return out

def compute_features_f64path(wave, opts):
    """The same formulas over the same float32 tables (window, mel weights, DCT, lifter, pre-emphasis coefficient as the reference's float32 code makes them) with
    every operation on the samples in FLOAT64 and exact twiddles: the exact value of what the reference computes (returned as float64).  The yardstick of the
    truth-distance gates: the reference's float32 binary is up to 1.1e-4 away from it on log-mel values, the GPU kernel (float64 data path) ~1e-6."""
    wave = np.ascontiguousarray(wave, dtype=np.float32)
    T = num_frames(len(wave), opts); dim = lib().k3o_feat_dim(ctypes.byref(opts))
    out = np.zeros((max(T, 0), dim), dtype=np.float64)
    if T > 0:
        r = lib().k3o_compute_features_f64path(ctypes.byref(opts), wave.ctypes.data, len(wave), out.ctypes.data)
        assert r == T, r
    return out

def cmvn_offline(feats, norm_vars=False):
    f = np.array(feats, dtype=np.float32, order='C', copy=True)
    r = lib().k3o_cmvn_offline(f.ctypes.data, f.shape[0], f.shape[1], int(norm_vars)); assert r == 0
    return f

def cmvn_online(feats, global_stats, speaker_stats=None, cmn_window=600, speaker_frames=600, global_frames=200, norm_means=True, norm_vars=False, skip_dims=()):
    """apply-cmvn-online on one utterance (feat/online-feature.cc:361-468).  Stats are [2 x (dim+1)] float64."""
    f = np.ascontiguousarray(feats, dtype=np.float32); out = np.empty_like(f)
    g = np.ascontiguousarray(global_stats, dtype=np.float64); assert g.shape == (2, f.shape[1] + 1)
    sp = None if speaker_stats is None else np.ascontiguousarray(speaker_stats, dtype=np.float64)
    sk = np.ascontiguousarray(list(skip_dims), dtype=np.int32)
    r = lib().k3o_cmvn_online(f.ctypes.data, out.ctypes.data, f.shape[0], f.shape[1], cmn_window, speaker_frames, global_frames, int(norm_means), int(norm_vars),
                              g.ctypes.data, None if sp is None else sp.ctypes.data, sk.ctypes.data if len(sk) else None, len(sk))
    if r != 0: raise ValueError("online CMVN: the reference raises an error for these stats/options")
    return out


def resample_waveform(wave, rate_in, rate_out):
    """ResampleWaveform (feat/resample.cc:363-372) = LinearResample (:33-60 constructor, :61-103 GetNumOutputSamples, :105-128 SetIndexesAndWeights, :142-205 Resample with
    flush = true, :232-246 FilterFunc): cutoff 0.99 * 0.5 * min(rates) as float32, six zero crossings; FilterFunc takes and returns float32 and evaluates in float64.
    Test infrastructure (numpy); the dot products are summed in float32 in tap order."""
    import math
    wave = np.asarray(wave, np.float32); rate_in = int(rate_in); rate_out = int(rate_out); num_zeros = 6
    cutoff = np.float32(0.99 * 0.5 * np.float32(min(rate_in, rate_out)))
    base = math.gcd(rate_in, rate_out); in_unit = rate_in // base; out_unit = rate_out // base
    window_width = num_zeros / (2.0 * float(cutoff))
    def filter_func(t):                 # t: float32
        t = float(np.float32(t))
        window = np.float32(0.5 * (1 + math.cos(6.283185307179586476925286766559005 * float(cutoff) / num_zeros * t))) if abs(t) < num_zeros / (2.0 * float(cutoff)) else np.float32(0.0)
        filt = np.float32(math.sin(6.283185307179586476925286766559005 * float(cutoff) * t) / (math.pi * t)) if t != 0 else np.float32(2) * cutoff
        return np.float32(filt * window)
    first, weights = [], []
    for i in range(out_unit):
        output_t = i / float(rate_out); lo = math.ceil((output_t - window_width) * rate_in); hi = math.floor((output_t + window_width) * rate_in)
        first.append(lo); weights.append(np.array([filter_func(np.float32((lo + j) / float(rate_in) - output_t)) / np.float32(rate_in) for j in range(hi - lo + 1)], np.float32))
    tick = rate_in // base * rate_out; interval = len(wave) * (tick // rate_in); per_out = tick // rate_out
    if interval <= 0: return np.zeros(0, np.float32)
    last = interval // per_out
    if last * per_out == interval: last -= 1
    out = np.zeros(last + 1, np.float32); n = len(wave)
    for s in range(last + 1):
        unit, ph = divmod(s, out_unit); f = first[ph] + unit * in_unit; w = weights[ph]
        a = max(0, -f); b = min(len(w), n - f)
        if b > a:
            acc = np.float32(0.0)
            for x in (w[a:b] * wave[f + a:f + b]): acc = np.float32(acc + x)
            out[s] = acc
    return out
