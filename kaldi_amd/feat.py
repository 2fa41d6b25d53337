"""Python plumbing over the feature C ABI (k3_feat_*): what the C++ adapter
kaldi::CudaSpectralFeatures / OnlineCudaFeaturePipeline (kaldi_amd/host) calls, exposed to tests and
bench.py.  torch is used for device buffers and the current stream only."""
import ctypes, numpy as np, torch
from . import lib as _l

def fbank_options(**kw):
    """FbankOptions defaults (feat/feature-fbank.h:44-61, feature-window.h:53-66, mel-computations.h:56-58)."""
    o = _l.FeatOpts(16000.0, 10.0, 25.0, 1.0, 0.97, 0.42, 1, 1, 1, 3, 23, 20.0, 0.0, 100.0, -500.0, 0, 0, 0.0, 1, 0, 1, 1, 13, 22.0, 0, 1.0)
    for k, v in kw.items():
        setattr(o, k, _l.WINDOW_TYPES[v] if k == "window_type" and isinstance(v, str) else v)
    return o

def mfcc_options(**kw):
    """MfccOptions defaults (feat/feature-mfcc.h:40-60): 23 bins, 13 ceps, use_energy, lifter 22."""
    o = fbank_options(use_energy=1, feature_type=1)
    for k, v in kw.items():
        setattr(o, k, _l.WINDOW_TYPES[v] if k == "window_type" and isinstance(v, str) else v)
    return o

def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

class SpectralFeatures:
    """Batched equivalent of CudaSpectralFeatures(opts) (cudafeat/feature-spectral-cuda.h): ComputeFeatures over
    a ragged batch of utterances held in one device buffer."""
    def __init__(self, opts):
        self._L = _l.load()
        self._h = ctypes.c_void_p()
        self.opts = opts
        _l.check(self._L.k3_feat_plan_create(ctypes.byref(opts), ctypes.byref(self._h)))
        self.dim = self._L.k3_feat_dim(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._L.k3_feat_plan_destroy(self._h); self._h.value = None
        except Exception:      # interpreter shutdown
            pass

    def Dim(self):
        return self.dim

    def NumFrames(self, nsamp):
        return self._L.k3_feat_num_frames(self._h, int(nsamp))

    def offsets(self, lengths, device):
        """host-side bookkeeping: (wave_offsets, frame_offsets) int64 device tensors + total_frames."""
        wo = [0]; fo = [0]
        for n in lengths:
            wo.append(wo[-1] + int(n)); fo.append(fo[-1] + self.NumFrames(n))
        return (torch.tensor(wo, dtype=torch.int64, device=device), torch.tensor(fo, dtype=torch.int64, device=device), fo[-1], fo)

    def ComputeFeatures(self, waves, wave_offsets, frame_offsets, total_frames, out=None):
        """waves: float32 (CuVector<BaseFloat>, the reference's interface) or int16 (PCM16 as read from a wav file: same values, half the
        bytes) [sum nsamp] on the GPU; returns float32 [total_frames, dim]."""
        assert waves.is_cuda and waves.dtype in (torch.float32, torch.int16) and waves.is_contiguous()
        if out is None:
            out = torch.empty((total_frames, self.dim), dtype=torch.float32, device=waves.device)
        fn = self._L.k3_feat_compute_batch if waves.dtype == torch.float32 else self._L.k3_feat_compute_batch_pcm16
        _l.check(fn(self._h, waves.data_ptr(), wave_offsets.data_ptr(), frame_offsets.data_ptr(),
                                               wave_offsets.numel() - 1, int(total_frames), out.data_ptr(), out.stride(0), _stream()))
        return out

def ResampleWaveform(waves, lengths, rate_in, rate_out):
    """ResampleWaveform (feat/resample.cc:363-372) on a batch: `waves` = float32 CUDA tensor with the utterances back to back (lengths[u] samples each, at rate_in Hz).
    Returns (resampled tensor, new lengths) -- what OfflineFeatureTpl::ComputeFeatures (feat/feature-common-inl.h:29-57) feeds its frame loop under --allow-downsample / --allow-upsample."""
    L = _l.load(); assert waves.is_cuda and waves.dtype == torch.float32 and waves.is_contiguous()
    io = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64); assert io[-1] == waves.numel()
    nl = [int(L.k3_resample_num_samples(int(rate_in), int(rate_out), int(n))) for n in lengths]; oo = np.concatenate([[0], np.cumsum(nl)]).astype(np.int64)
    out = torch.empty(int(oo[-1]), dtype=torch.float32, device=waves.device)
    _l.check(L.k3_resample_batch(int(rate_in), int(rate_out), waves.data_ptr(), io.ctypes.data, len(lengths), out.data_ptr(), oo.ctypes.data, _stream()))
    return out, nl

def ApplyCmvnOffline(feats, frame_offsets, norm_vars=False, stats=None):
    """compute-cmvn-stats | apply-cmvn per utterance, in place (transform/cmvn.cc:30-115)."""
    assert feats.is_cuda and feats.dtype == torch.float32
    L = _l.load()
    _l.check(L.k3_cmvn_offline_batch(feats.data_ptr(), feats.stride(0), feats.shape[1], frame_offsets.data_ptr(),
                                     frame_offsets.numel() - 1, int(norm_vars), stats.data_ptr() if stats is not None else None, _stream()))
    return feats


def ApplyCmvnOnline(feats, frame_offsets, global_stats, speaker_stats=None, cmn_window=600, speaker_frames=600, global_frames=200,
                    norm_means=True, norm_vars=False, skip_dims=(), out=None):
    """apply-cmvn-online on a batch of whole utterances (OnlineCmvn::GetFrame per frame, feat/online-feature.cc:361-468; GPU reference
    CudaOnlineCmvn::ComputeFeatures).  feats float32 [rows, dim] on the GPU; global_stats float64 [2, dim+1]; speaker_stats (optional)
    float64 [U, 2, dim+1].  Returns a new matrix (the window needs the raw frames, so not in place)."""
    assert feats.is_cuda and feats.dtype == torch.float32
    dim = feats.shape[1]; U = frame_offsets.numel() - 1
    g = torch.as_tensor(global_stats, dtype=torch.float64).to(feats.device).contiguous(); assert tuple(g.shape) == (2, dim + 1)
    sp = None
    if speaker_stats is not None:
        sp = torch.as_tensor(speaker_stats, dtype=torch.float64).to(feats.device).contiguous(); assert tuple(sp.shape) == (U, 2, dim + 1)
    if out is None: out = torch.empty_like(feats)
    o = _l.OnlineCmvnOpts(int(cmn_window), int(speaker_frames), int(global_frames), int(bool(norm_means)), int(bool(norm_vars)))
    import ctypes as _ct
    sk = (_ct.c_int32 * max(1, len(skip_dims)))(*[int(d) for d in skip_dims])
    _l.check(_l.load().k3_cmvn_online_batch(feats.data_ptr(), feats.stride(0), out.data_ptr(), out.stride(0), dim, frame_offsets.data_ptr(), U, _ct.byref(o),
                                            g.data_ptr(), sp.data_ptr() if sp is not None else None, _ct.cast(sk, _ct.c_void_p) if len(skip_dims) else None, len(skip_dims), _stream()))
    return out
