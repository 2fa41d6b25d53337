"""Online i-vector extraction of whole utterances on the GPU (k3_ivector_* of include/k3hip.h): the Python face of what
BatchedIvectorExtractorCuda (cudafeat/feature-online-batched-ivector-cuda.h:30-61) is to the reference's pipeline, with the arithmetic of
OnlineIvectorFeature (online2/online-ivector-feature.cc).  The model files are read by libk3host (k3h_ivector_config_read =
OnlineIvectorExtractionInfo::Init); torch is used for device buffers and the current stream only."""
import ctypes, torch
import numpy as np
from . import lib as _l
from . import hostlib as _h


class OnlineIvectorExtractionInfo:
    """OnlineIvectorExtractionInfo(config) (online2/online-ivector-feature.cc:29-98): the parsed --ivector-extraction-config and the files it names.
    Relative paths inside the config are resolved against the process's working directory, like the reference."""
    INTS = ("feat_dim", "lda_rows", "lda_cols", "num_gauss", "ivector_dim", "left_context", "right_context", "ivector_period", "num_gselect", "num_cg_iters",
            "cmn_window", "speaker_frames", "global_frames", "normalize_mean", "normalize_variance", "online_cmvn_iextractor")
    REALS = ("min_post", "posterior_scale", "max_count", "prior_offset", "max_remembered_frames")

    def __init__(self, config_rxfilename):
        self._H = _h.load(); self._c = ctypes.c_void_p()
        _h.check(self._H.k3h_ivector_config_read(str(config_rxfilename).encode(), ctypes.byref(self._c)))
        ints = (ctypes.c_int32 * 16)(); reals = (ctypes.c_double * 5)(); self._ptr = [ctypes.c_void_p() for _ in range(7)]
        _h.check(self._H.k3h_ivector_config_get(self._c, ints, reals, *[ctypes.byref(p) for p in self._ptr]))
        for k, v in zip(self.INTS, ints): setattr(self, k, int(v))
        for k, v in zip(self.REALS, reals): setattr(self, k, float(v))

    def model(self):
        m = _l.IvectorModel(self.feat_dim, self.lda_rows, self.lda_cols, self.num_gauss, self.ivector_dim, *[p.value for p in self._ptr], self.prior_offset)
        return m

    def opts(self, **overrides):
        o = _l.IvectorOpts(); _l.load().k3_ivector_opts_default(ctypes.byref(o))
        for k in ("left_context", "right_context", "num_gselect", "min_post", "posterior_scale", "max_count", "ivector_period", "num_cg_iters", "online_cmvn_iextractor"): setattr(o, k, getattr(self, k))
        for k in ("cmn_window", "speaker_frames", "global_frames", "normalize_mean", "normalize_variance"): setattr(o.cmvn, k, getattr(self, k))
        for k, v in overrides.items(): setattr(o, k, v)
        return o

    def __del__(self):
        try:
            if getattr(self, "_c", None) and self._c.value: self._H.k3h_ivector_config_free(self._c); self._c.value = None
        except Exception:
            pass


class BatchedIvectorExtractor:
    """GetIvectors over a ragged batch of whole utterances: feats [sum T_u, feat_dim] float32 on the GPU, utterance u = rows
    frame_offsets[u]..[u+1] -> i-vectors [sum ceil(T_u / period), ivector_dim] (one row per --ivector-period frames) and their row offsets."""
    def __init__(self, info, **overrides):
        self._L = _l.load(); self._h = ctypes.c_void_p(); self.info = info
        m = info.model(); o = info.opts(**overrides) if hasattr(info, "opts") else info._opts
        _l.check(self._L.k3_ivector_create(ctypes.byref(m), ctypes.byref(o), ctypes.byref(self._h)))
        self.left_context, self.right_context = int(o.left_context), int(o.right_context)      # of the LDA splice: an i-vector for frame t needs the features up to t + right_context
        i = _l.IvectorInfo(); _l.check(self._L.k3_ivector_get_info(self._h, ctypes.byref(i)))
        self.feat_dim, self.lda_dim, self.num_gauss, self.ivector_dim, self.ivector_period = i.feat_dim, i.lda_dim, i.num_gauss, i.ivector_dim, i.ivector_period

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value: self._L.k3_ivector_destroy(self._h); self._h.value = None
        except Exception:
            pass

    @classmethod
    def FromArrays(cls, lda, global_cmvn_stats, gconsts, means_invvars, inv_vars, M, sigma_inv_packed, prior_offset, **opts):
        """the same from host arrays (the layout of k3_ivector_model): lda [D, F*(l+r+1) (+1)], stats [2, F+1], UBM [G], [G, D], [G, D], M [G, D, R],
        Sigma^-1 [G, D(D+1)/2] packed lower triangles; opts = fields of k3_ivector_opts (cmvn_* for the CMVN options)"""
        class _Info: pass
        i = _Info(); c = lambda a, t: np.ascontiguousarray(np.asarray(a), dtype=t)
        i._keep = [c(lda, np.float32), c(global_cmvn_stats, np.float64), c(gconsts, np.float64), c(means_invvars, np.float64), c(inv_vars, np.float64), c(M, np.float64), c(sigma_inv_packed, np.float64)]
        G, D, R = i._keep[5].shape
        mdl = _l.IvectorModel(i._keep[1].shape[1] - 1, i._keep[0].shape[0], i._keep[0].shape[1], G, R, *[a.ctypes.data for a in i._keep], float(prior_offset))
        i.model = lambda: mdl
        o = _l.IvectorOpts(); _l.load().k3_ivector_opts_default(ctypes.byref(o))
        for k, v in opts.items():
            if k.startswith("cmvn_"): setattr(o.cmvn, k[5:], v)
            else: setattr(o, k, v)
        i._opts = o
        return cls(i)

    def FeatDim(self): return self.feat_dim
    def LdaDim(self): return self.lda_dim
    def IvectorDim(self): return self.ivector_dim
    def NumGauss(self): return self.num_gauss

    def GetIvectors(self, feats, frame_offsets, cmvn_speaker_stats=None, stats_in=None, return_stats=False, accumulate_tail=False, frame_weights=None):
        """-> (ivectors, row_offsets[, stats]).  cmvn_speaker_stats [U, 2, feat_dim+1] / stats_in [U, StatsSize()] (float64, host or GPU): the adaptation state
        the speaker's earlier utterances left (OnlineIvectorExtractorAdaptationState); return_stats: also the i-vector statistics after each utterance -- at its last estimate, or
        with accumulate_tail over all of its frames (what ivector-extract-online2 --repeat=true hands on).  frame_weights [total frames] (float32, host or GPU): the weight of
        each frame in the statistics (silence weighting, ivector-extract-online2 --frame-weights-rspecifier): 0 drops the frame, w scales its posteriors and prunes them at min(0.99, min_post/|w|)."""
        self._L.k3_ivector_set_accumulate_tail.restype = None; self._L.k3_ivector_set_accumulate_tail(self._h, ctypes.c_int32(1 if accumulate_tail else 0))
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.dim() == 2 and feats.stride(1) == 1
        fo = np.ascontiguousarray(np.asarray(frame_offsets, dtype=np.int64)); U = fo.size - 1
        ro = np.zeros(U + 1, np.int64)
        n = self._L.k3_ivector_num_rows(self._h, U, fo.ctypes.data, ro.ctypes.data)
        if n < 0: raise _l.K3Error("k3_ivector_num_rows: bad argument")
        out = torch.empty((n, self.ivector_dim), dtype=torch.float32, device=feats.device)
        dev = lambda a: None if a is None else torch.as_tensor(np.asarray(a, np.float64) if not torch.is_tensor(a) else a, dtype=torch.float64).to(feats.device).contiguous()
        cm, si = dev(cmvn_speaker_stats), dev(stats_in)
        so = torch.empty((U, self.StatsSize()), dtype=torch.float64, device=feats.device) if return_stats else None
        ptr = lambda t: None if t is None else t.data_ptr()
        fw = None if frame_weights is None else torch.as_tensor(frame_weights, dtype=torch.float32).to(feats.device).contiguous()
        if fw is not None and fw.numel() != int(fo[-1]): raise ValueError("frame_weights: one weight per frame")
        _l.check(self._L.k3_ivector_extract_batch_weighted(self._h, feats.data_ptr(), feats.stride(0), fo.ctypes.data, U, ptr(fw), out.data_ptr(), out.stride(0), ptr(cm), ptr(si), ptr(so),
                                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return (out, ro, so) if return_stats else (out, ro)

    def StatsSize(self): return int(self._L.k3_ivector_stats_size(self._h))


class IvectorStream:
    """One audio stream's extractor state (k3_ivector_stream_*): AcceptFrames(feats, finished) takes the new feature rows and returns the i-vectors they complete -- rows of the whole
    utterance's GetIvectors, bit for bit, every frame processed once -- Latest() the estimate the reference's online decodable would hand the network now (zeros before the first)."""
    def __init__(self, extractor):
        self.ex = extractor; self._L = extractor._L; self._h = ctypes.c_void_p()
        _l.check(self._L.k3_ivector_stream_create(extractor._h, ctypes.byref(self._h)))
        self.latest = None

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value: self._L.k3_ivector_stream_destroy(self._h); self._h.value = None
        except Exception:
            pass

    def Reset(self): _l.check(self._L.k3_ivector_stream_reset(self._h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))); self.latest = None
    def NumRows(self): return int(self._L.k3_ivector_stream_num_rows(self._h))

    def AcceptFrames(self, feats, finished):
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.dim() == 2 and (feats.shape[0] == 0 or feats.stride(1) == 1)
        n = int(feats.shape[0]); cap = (n + self.ex.right_context) // self.ex.ivector_period + 2      # (the end of the stream releases the frames that waited for their right context)
        rows = torch.empty((cap, self.ex.ivector_dim), dtype=torch.float32, device=feats.device); got = ctypes.c_int32(0)
        if self.latest is None: self.latest = torch.zeros(self.ex.ivector_dim, dtype=torch.float32, device=feats.device)
        _l.check(self._L.k3_ivector_stream_accept(self._h, feats.data_ptr() if n else None, feats.stride(0) if n else self.ex.feat_dim, n, int(bool(finished)), rows.data_ptr(), rows.stride(0), cap, ctypes.byref(got),
                                                  self.latest.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return rows[:got.value]

    def Latest(self): return self.latest


def AcceptFramesBatch(streams, feats, frame_offsets, finished):
    """k3_ivector_stream_accept_batch: stream i takes rows frame_offsets[i] .. [i + 1] of feats (GPU, float32); -> [len(streams), ivector_dim], every stream's latest estimate.  One launch
    per stage for the whole batch; per stream the same results as IvectorStream.AcceptFrames."""
    ex = streams[0].ex; n = len(streams); fo = np.ascontiguousarray(np.asarray(frame_offsets, dtype=np.int64)); fin = np.ascontiguousarray(np.asarray(finished, dtype=np.int32))
    assert fo.size == n + 1 and fin.size == n and feats.is_cuda and feats.dtype == torch.float32 and feats.dim() == 2 and (feats.shape[0] == 0 or feats.stride(1) == 1)
    hs = (ctypes.c_void_p * n)(*[s._h.value for s in streams]); out = torch.empty((n, ex.ivector_dim), dtype=torch.float32, device=feats.device)
    _l.check(ex._L.k3_ivector_stream_accept_batch(hs, n, feats.data_ptr() if feats.shape[0] else None, feats.stride(0) if feats.shape[0] else ex.feat_dim, fo.ctypes.data, fin.ctypes.data, out.data_ptr(), out.stride(0),
                                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for i, s in enumerate(streams): s.latest = out[i]
    return out
