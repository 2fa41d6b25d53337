"""Python plumbing over the nnet3 C ABI (k3_nnet_*): what the C++ adapter (BatchedStaticNnet3 / NnetComputer
front-end in kaldi_amd/host) calls, exposed to tests and bench.py.  torch = device buffers + stream only."""
import ctypes, numpy as np, torch
from . import lib as _l

class Nnet:
    """A loaded nnet3 model (raw nnet or final.mdl), fused for the MI355X TDNN/TDNN-F path."""
    def __init__(self, path):
        self._L = _l.load(); self._h = ctypes.c_void_p()
        _l.check(self._L.k3_nnet_load(str(path).encode(), ctypes.byref(self._h)))
        self.info = _l.NnetInfo(); _l.check(self._L.k3_nnet_get_info(self._h, ctypes.byref(self.info)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._L.k3_nnet_destroy(self._h); self._h.value = None
        except Exception:      # interpreter shutdown
            pass

    def priors(self):
        p = np.zeros(self.info.output_dim, np.float32)
        _l.check(self._L.k3_nnet_get_priors(self._h, p.ctypes.data)); return p

class NnetBatch:
    """One planned ragged batch (cf. BatchedStaticNnet3::RunBatch, cudadecoder/batched-static-nnet3.cc:293-365)."""
    def __init__(self, nnet, num_frames, frame_subsampling_factor=1, log_priors=None, acoustic_scale=1.0, ivector_rows=None, online_ivector_period=0, frames_per_chunk=50):
        """ivector_rows (models with an i-vector input): None = one i-vector per utterance (--ivectors); else the number of --online-ivectors rows of
        every utterance, with online_ivector_period and frames_per_chunk as nnet3-compute / nnet3-latgen-faster take them"""
        self.nnet = nnet; self._L = nnet._L; self._h = ctypes.c_void_p()
        nf = np.ascontiguousarray(num_frames, dtype=np.int32)
        lp = None if log_priors is None else np.ascontiguousarray(log_priors, dtype=np.float32)
        if nnet.info.ivector_dim > 0:
            rows = None if ivector_rows is None else np.ascontiguousarray(ivector_rows, dtype=np.int32)
            _l.check(self._L.k3_nnet_batch_create_ivector(nnet._h, len(nf), nf.ctypes.data, int(frame_subsampling_factor), None if lp is None else lp.ctypes.data, float(acoustic_scale),
                                                          int(frames_per_chunk), 0 if rows is None else int(online_ivector_period), None if rows is None else rows.ctypes.data, ctypes.byref(self._h)))
            self.total_ivector_rows = self._L.k3_nnet_batch_ivector_rows(self._h)
        else:
            _l.check(self._L.k3_nnet_batch_create(nnet._h, len(nf), nf.ctypes.data, int(frame_subsampling_factor),
                                                  None if lp is None else lp.ctypes.data, float(acoustic_scale), ctypes.byref(self._h)))
        off = np.zeros(len(nf) + 1, np.int64)
        self.total_out_rows = self._L.k3_nnet_batch_output_rows(self._h, off.ctypes.data)
        self.out_offsets = off
        self.flops = self._L.k3_nnet_batch_flops(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._L.k3_nnet_batch_destroy(self._h); self._h.value = None
        except Exception:      # interpreter shutdown
            pass

    def set_precision(self, mode):
        """0 = the FP32 matrix core with the reference's summation order (default, the parity path); 1 = split-bf16 (exploratory: six bf16 matrix-core products per product, operands split by the loader); 2 = the same with the
        activations' three bf16 planes written by the producing epilogue (bit-identical to 1)"""
        _l.check(self._L.k3_nnet_batch_set_precision(self._h, int(mode)))
    def forward(self, feats, out=None, ivectors=None):
        """feats: float32 [sum T_u, >= input_dim] on the GPU -> float32 [total_out_rows, output_dim]; ivectors (models with an i-vector input):
        float32 [total_ivector_rows, ivector_dim] on the GPU, the utterances' rows back to back."""
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.stride(1) == 1
        if out is None:
            out = torch.empty((self.total_out_rows, self.nnet.info.output_dim), dtype=torch.float32, device=feats.device)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if ivectors is None:
            _l.check(self._L.k3_nnet_forward(self._h, feats.data_ptr(), feats.stride(0), out.data_ptr(), out.stride(0), st))
        else:
            assert ivectors.is_cuda and ivectors.dtype == torch.float32 and ivectors.stride(1) == 1 and ivectors.shape[0] == self.total_ivector_rows
            _l.check(self._L.k3_nnet_forward_ivector(self._h, feats.data_ptr(), feats.stride(0), ivectors.data_ptr(), ivectors.stride(0), out.data_ptr(), out.stride(0), st))
        return out


class BatchedStaticNnet3:
    """Streaming driver of the nnet3 forward, the role of cudadecoder/batched-static-nnet3.{h,cc} (BatchedStaticNnet3::RunBatch
    :293-365): the network is planned ONCE for `max_batch_size` slots of `frames_per_chunk` input frames plus the model's left and
    right input context; between calls the last context frames of every channel are stashed on the GPU; the first chunk of a channel
    sees frame 0 replicated to the left, the last chunk flushes the frames still waiting for right context with the last frame
    replicated (DecodableNnetSimple's edge handling, nnet3/nnet-am-decodable-simple.cc:154-163).  Like the reference it recomputes
    the context frames of every chunk instead of keeping per-layer state, so every output row is the same arithmetic on the same
    inputs as in a whole-utterance forward: the chunked log-likelihoods are bit-identical to NnetBatch.forward's.

    RunBatch(channels, chunks, is_first_chunk, is_last_chunk) -> list of [n_out_i, output_dim] GPU tensors (views into an internal
    buffer, valid until the next call): the output frames of each slot's channel that became computable, in time order."""
    def __init__(self, nnet, max_batch_size, nchannels=None, frames_per_chunk=150, frame_subsampling_factor=3, log_priors=None, acoustic_scale=1.0,
                 device="cuda:0"):
        from .cumatrix import CuMatrix
        self._CuMatrix = CuMatrix
        s = self.s = int(frame_subsampling_factor); C = self.C = int(frames_per_chunk)
        if C <= 0 or C % s: raise ValueError("frames_per_chunk must be a positive multiple of the frame-subsampling factor")
        self.nnet, self.B = nnet, int(max_batch_size); self.nch = int(nchannels or max_batch_size)
        if self.nch < self.B: raise ValueError("nchannels must be >= max_batch_size")
        self.Lc = -(-nnet.info.left_context // s) * s; self.Rc = int(nnet.info.right_context)
        self.P = self.Lc + C + self.Rc; self.rows_per_slot = -(-self.P // s)
        self.dim = nnet.info.input_dim; self.dev = torch.device(device)
        self.ivector_dim = int(nnet.info.ivector_dim)      # models with the recipes' i-vector input: one i-vector per slot and pass (RunBatch(..., ivectors=))
        self.batch = NnetBatch(nnet, [self.P] * self.B, s, log_priors, acoustic_scale)
        self.iv = torch.zeros((self.B, max(1, self.ivector_dim)), dtype=torch.float32, device=self.dev) if self.ivector_dim else None
        self.S = self.Lc + self.Rc + C + 2 * s             # frames a channel can have waiting between calls (generous)
        self.stash = [torch.zeros((self.nch * self.S, self.dim), dtype=torch.float32, device=self.dev) for _ in range(2)]
        self.cur = 0
        self.inp = torch.zeros((self.B * self.P, self.dim), dtype=torch.float32, device=self.dev)
        self.out = torch.empty((self.batch.total_out_rows, nnet.info.output_dim), dtype=torch.float32, device=self.dev)
        self.t_next = np.zeros(self.nch, np.int64); self.n_seen = np.zeros(self.nch, np.int64); self.stash_lo = np.zeros(self.nch, np.int64)

    def GetNOutputFramesPerChunk(self): return self.C // self.s
    def GetTotalNnet3RightContext(self): return self.Rc

    def _pass(self, channels, new, n_new, last, ivectors=None):
        """one planned forward over the slots: returns per-slot (row0, count) in self.out"""
        B, P, S, s = len(channels), self.P, self.S, self.s
        idx_st = np.full(self.B * P, -1, np.int32); idx_nw = np.full(self.B * P, -1, np.int32)
        upd_st = np.full(self.nch * S, -1, np.int32); upd_nw = np.full(self.nch * S, -1, np.int32)
        keep = np.ones(self.nch, bool); keep[list(channels)] = False
        for ch in np.nonzero(keep)[0]:                    # untouched channels keep their stash
            n = int(self.n_seen[ch] - self.stash_lo[ch]); upd_st[ch * S: ch * S + n] = np.arange(ch * S, ch * S + n, dtype=np.int32)
        res = []; new_off = np.concatenate([[0], np.cumsum(n_new)])
        for i, ch in enumerate(channels):
            avail = int(self.n_seen[ch] + n_new[i]); tn = int(self.t_next[ch])
            if last[i]: count = max(0, -(-(avail - tn) // s))
            else: count = max(0, (avail - 1 - self.Rc - tn) // s + 1) if avail - 1 - self.Rc - tn >= 0 else 0
            count = min(count, self.C // s)
            if avail > 0:
                tau = np.clip(np.arange(tn - self.Lc, tn - self.Lc + P), 0, avail - 1)
                from_new = tau >= self.n_seen[ch]
                idx_nw[i * P:(i + 1) * P][from_new] = (new_off[i] + tau[from_new] - self.n_seen[ch]).astype(np.int32)
                idx_st[i * P:(i + 1) * P][~from_new] = (ch * S + tau[~from_new] - self.stash_lo[ch]).astype(np.int32)
            res.append((i * self.rows_per_slot + self.Lc // s, count))
            tn2 = tn + count * s; lo2 = max(0, tn2 - self.Lc)
            tau = np.arange(lo2, avail); assert len(tau) <= S, "internal: stash capacity"
            from_new = tau >= self.n_seen[ch]
            upd_nw[ch * S: ch * S + len(tau)][from_new] = (new_off[i] + tau[from_new] - self.n_seen[ch]).astype(np.int32)
            upd_st[ch * S: ch * S + len(tau)][~from_new] = (ch * S + tau[~from_new] - self.stash_lo[ch]).astype(np.int32)
            self.t_next[ch], self.n_seen[ch], self.stash_lo[ch] = tn2, avail, lo2
        A, Bn = self.stash[self.cur], self.stash[self.cur ^ 1]
        d = lambda a: torch.from_numpy(a).to(self.dev)
        M = self._CuMatrix
        M(self.inp).CopyRows(M(A), d(idx_st))
        if new is not None and new.shape[0] > 0: M(self.inp).AddRows(1.0, M(new), d(idx_nw))
        M(Bn).CopyRows(M(A), d(upd_st))
        if new is not None and new.shape[0] > 0: M(Bn).AddRows(1.0, M(new), d(upd_nw))
        self.cur ^= 1
        if self.ivector_dim:
            if ivectors is None or ivectors.shape != (B, self.ivector_dim): raise ValueError("this model has an i-vector input: pass ivectors [len(channels), %d]" % self.ivector_dim)
            self.iv.zero_(); self.iv[:B] = ivectors.to(self.dev, torch.float32)
            self.batch.forward(self.inp, out=self.out, ivectors=self.iv)
        else: self.batch.forward(self.inp, out=self.out)
        return res

    def RunBatch(self, channels, chunks, is_first_chunk, is_last_chunk, ivectors=None):
        """ivectors (models with an i-vector input): [len(channels), ivector_dim], the i-vector every listed channel's chunk is evaluated with -- DecodableNnetLoopedOnlineBase::AdvanceChunk
        (nnet3/decodable-online-looped.cc:166-205) hands the network ONE i-vector per chunk, the extractor's latest"""
        if len(channels) > self.B or len(set(channels)) != len(channels): raise ValueError("at most one chunk per channel and max_batch_size slots")
        n_new = [int(c.shape[0]) for c in chunks]
        if any(n > self.C for n in n_new): raise ValueError("a chunk has more than frames_per_chunk frames")
        for ch, first in zip(channels, is_first_chunk):
            if first: self.t_next[ch] = self.n_seen[ch] = self.stash_lo[ch] = 0
        new = torch.cat([c.to(self.dev, torch.float32) for c in chunks], 0) if chunks else None
        res = self._pass(list(channels), new, n_new, list(is_last_chunk), ivectors)
        outs = [[self.out[r0:r0 + n].clone()] if n else [] for r0, n in res]
        # end of stream: the frames that were waiting for right context may need more passes than one chunk's worth of output rows
        pending = [i for i, ch in enumerate(channels) if is_last_chunk[i] and self.t_next[ch] < self.n_seen[ch]]
        while pending:
            chs = [channels[i] for i in pending]
            res2 = self._pass(chs, None, [0] * len(chs), [True] * len(chs), None if ivectors is None else ivectors[pending])
            for i, (r0, n) in zip(pending, res2):
                if n: outs[i].append(self.out[r0:r0 + n].clone())
            pending = [i for i in pending if self.t_next[channels[i]] < self.n_seen[channels[i]]]
        od = self.nnet.info.output_dim
        return [torch.cat(o, 0) if o else torch.empty((0, od), dtype=torch.float32, device=self.dev) for o in outs]


class StreamNnet3:
    """Stateful streaming forward (k3_nnet_stream_*, round 5): the role of BatchedStaticNnet3 WITHOUT re-evaluating a chunk's context -- every node keeps its last rows per
    channel, a pass consumes frames_per_chunk new frames per channel, every row of every node is computed once per stream; outputs bit-identical to NnetBatch.forward over the
    whole utterance.  RunBatch has BatchedStaticNnet3's signature; frames are buffered per channel until a whole chunk (or the end of the stream) is there, so a call returns
    the output rows that became computable (possibly none), in time order."""
    def __init__(self, nnet, nchannels, frames_per_chunk=51, frame_subsampling_factor=3, log_priors=None, acoustic_scale=1.0, device="cuda:0"):
        self._L = _l.load(); self.nnet = nnet; self.nch = int(nchannels); self.C = int(frames_per_chunk); self.s = int(frame_subsampling_factor); self.dev = torch.device(device)
        lp = None if log_priors is None else np.ascontiguousarray(log_priors, np.float32)
        h = ctypes.c_void_p()
        _l.check(self._L.k3_nnet_stream_create(nnet._h, self.nch, self.C, self.s, None if lp is None else lp.ctypes.data, float(acoustic_scale), ctypes.byref(h)))
        self._h = h; info = _l.NnetStreamInfo(); _l.check(self._L.k3_nnet_stream_get_info(self._h, ctypes.byref(info))); self.info = info
        self.N_out = info.output_rows_per_pass; self.first_out = info.first_output_time; self.dim = nnet.info.input_dim; self.odim = nnet.info.output_dim
        self.out = torch.empty((self.N_out * self.nch, self.odim), dtype=torch.float32, device=self.dev)
        self.passes = np.zeros(self.nch, np.int64); self.n_total = np.zeros(self.nch, np.int64); self.ended = np.zeros(self.nch, bool)
        self.pend = [torch.empty((0, self.dim), dtype=torch.float32, device=self.dev) for _ in range(self.nch)]; self.needs_seed = np.zeros(self.nch, bool)
    def __del__(self):
        try:
            if getattr(self, "_h", None): self._L.k3_nnet_stream_destroy(self._h); self._h = None
        except Exception: pass
    def GetNOutputFramesPerChunk(self): return self.C // self.s
    def _wants_pass(self, ch):
        if self.pend[ch].shape[0] >= self.C: return True
        if not self.ended[ch]: return False
        if self.pend[ch].shape[0] > 0: return True
        last_out = (int(self.n_total[ch]) - 1) // self.s * self.s      # outputs exist for t = 0, s, 2s, ... < number of frames
        hi_out = int(self.passes[ch]) * self.C - 1 - self.info.right_context
        return self.n_total[ch] > 0 and hi_out < last_out
    def _pass(self, chs):
        st = torch.cuda.current_stream(self.dev).cuda_stream
        start = np.zeros(self.nch, np.int64); cnt = np.full(self.nch, -1, np.int32); parts = []; at = 0
        for ch in chs:      # (any order: a channel's rows are addressed by (start, count))
            n = min(self.C, int(self.pend[ch].shape[0])); start[ch] = at; cnt[ch] = n
            if n: parts.append(self.pend[ch][:n]); self.pend[ch] = self.pend[ch][n:]; at += n
        new = torch.cat(parts, 0).contiguous() if parts else None
        _l.check(self._L.k3_nnet_stream_forward(self._h, None if new is None else new.data_ptr(), self.dim if new is None else new.stride(0), start.ctypes.data, cnt.ctypes.data, self.out.data_ptr(), self.out.stride(0), st))
        res = {}
        for ch in chs:
            t = self.first_out + int(self.passes[ch]) * self.C + np.arange(self.N_out) * self.s
            ok = t >= 0
            if self.ended[ch]: ok &= t < self.n_total[ch]
            k = np.nonzero(ok)[0]
            res[ch] = self.out[torch.from_numpy(k * self.nch + ch).to(self.dev)] if len(k) else None
            self.passes[ch] += 1
        return res
    def RunBatch(self, channels, chunks, is_first_chunk, is_last_chunk, ivectors=None):
        if ivectors is not None: raise ValueError("StreamNnet3: models with an i-vector input use BatchedStaticNnet3")
        if len(set(channels)) != len(channels): raise ValueError("at most one chunk per channel")
        st = torch.cuda.current_stream(self.dev).cuda_stream
        for ch, f in zip(channels, is_first_chunk):
            if f: self.passes[ch] = 0; self.n_total[ch] = 0; self.ended[ch] = False; self.pend[ch] = self.pend[ch][:0]; self.needs_seed[ch] = True
        for ch, c, last in zip(channels, chunks, is_last_chunk):
            if self.ended[ch]: raise ValueError("StreamNnet3: frames for a stream that has ended")
            self.pend[ch] = torch.cat([self.pend[ch], c.to(self.dev, torch.float32)], 0); self.n_total[ch] += int(c.shape[0]); self.ended[ch] = bool(last)
        fresh = [ch for ch in channels if self.needs_seed[ch] and self.pend[ch].shape[0] > 0]      # a stream is seeded with its first frame as soon as that exists
        if fresh:
            ids = np.ascontiguousarray(fresh, np.int32); f0 = torch.stack([self.pend[ch][0] for ch in fresh], 0).contiguous()
            _l.check(self._L.k3_nnet_stream_reset(self._h, ids.ctypes.data, len(ids), f0.data_ptr(), f0.stride(0), st))
            for ch in fresh: self.needs_seed[ch] = False
        outs = {ch: [] for ch in channels}
        while True:
            run = [ch for ch in channels if self._wants_pass(ch)]
            if not run: break
            for ch, o in self._pass(run).items():
                if o is not None: outs[ch].append(o)
        return [torch.cat(outs[ch], 0) if outs[ch] else torch.empty((0, self.odim), dtype=torch.float32, device=self.dev) for ch in channels]
