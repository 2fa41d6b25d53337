"""Python plumbing over the decoder C ABI (k3_fst_*, k3_decoder_*): what the C++ adapters CudaFst / CudaDecoder
(kaldi_amd/host) call, exposed to tests and bench.py.  torch = device buffers + stream only."""
import ctypes, numpy as np, torch
from . import lib as _l
from .lattice import RawLattice

def decoder_config(**kw):
    """k3_decoder_config with LatticeFasterDecoderConfig defaults; override by keyword (beam=15.0, max_active=10000, ...)."""
    c = _l.DecoderConfig(); _l.load().k3_decoder_config_default(ctypes.byref(c))
    for k, v in kw.items():
        assert hasattr(c, k), k
        setattr(c, k, v)
    return c

class CudaFst:
    """The decoding graph resident in HBM (cf. cuda_decoder::CudaFst, cudadecoder/cuda-fst.h:62-149).
    fst: kaldi_amd.fst.Fst; tid2pdf: int32 map transition-id -> pdf-id (ApplyTransitionModelOnIlabels, cuda-fst.cc:166-175)."""
    def __init__(self, fst, tid2pdf):
        self._L = _l.load(); self._h = ctypes.c_void_p()
        t2p = np.ascontiguousarray(tid2pdf, np.int32)
        self.start = fst.start; self.num_states = fst.num_states; self.num_arcs = fst.num_arcs
        _l.check(self._L.k3_fst_create(fst.num_states, fst.start, fst.arc_offsets.ctypes.data, fst.ilabel.ctypes.data, fst.olabel.ctypes.data,
                                       fst.weight.ctypes.data, fst.nextstate.ctypes.data, fst.final.ctypes.data, t2p.ctypes.data, t2p.size, ctypes.byref(self._h)))

    @classmethod
    def empty(cls, num_states, num_arcs, start):
        """receiver side of the graph broadcast: an image of the right shape, filled by a collective"""
        self = cls.__new__(cls); self._L = _l.load(); self._h = ctypes.c_void_p()
        self.start = start; self.num_states = num_states; self.num_arcs = num_arcs
        _l.check(self._L.k3_fst_create_empty(num_states, num_arcs, start, ctypes.byref(self._h)))
        return self

    @classmethod
    def adopt(cls, handle):
        """wrap a k3_fst* the C ABI handed out (the receiving side of k3_fst_bcast); this object owns it"""
        self = cls.__new__(cls); self._L = _l.load(); self._h = ctypes.c_void_p(handle)
        self.start = int(self._L.k3_fst_start(self._h)); self.num_states = int(self._L.k3_fst_num_states(self._h)); self.num_arcs = int(self._L.k3_fst_num_arcs(self._h))
        return self

    def export_image(self, tensor):
        """copy the graph image into a uint8 CUDA tensor (the collective's buffer)"""
        assert tensor.is_cuda and tensor.dtype == torch.uint8 and tensor.numel() >= self.image()[1]
        _l.check(self._L.k3_fst_export_image(self._h, tensor.data_ptr()))

    def import_image(self, tensor):
        assert tensor.is_cuda and tensor.dtype == torch.uint8 and tensor.numel() >= self.image()[1]
        _l.check(self._L.k3_fst_import_image(self._h, tensor.data_ptr()))

    def image(self):
        """(device pointer, bytes) of the packed read-only graph image"""
        ptr, n = ctypes.c_void_p(), ctypes.c_int64()
        _l.check(self._L.k3_fst_image(self._h, ctypes.byref(ptr), ctypes.byref(n)))
        return ptr.value, n.value

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._L.k3_fst_destroy(self._h); self._h.value = None
        except Exception:      # interpreter shutdown
            pass

class RawLatticeBatch:
    """The lattices of one batch as handed over by k3_decoder_get_raw_lattices: flat arrays + per-utterance offsets.  Behaves like a
    list of RawLattice (len, indexing, iteration); the objects are numpy views created on access."""
    def __init__(self, so, ao, si, sf, ai, af, start):
        self.state_offsets, self.arc_offsets, self._si, self._sf, self._ai, self._af, self._start = so, ao, si, sf, ai, af, start
    def __len__(self): return len(self.state_offsets) - 1
    def __getitem__(self, u):
        if isinstance(u, slice): return [self[i] for i in range(*u.indices(len(self)))]
        if u < 0: u += len(self)
        if not 0 <= u < len(self): raise IndexError(u)
        s0, s1, a0, a1 = self.state_offsets[u], self.state_offsets[u + 1], self.arc_offsets[u], self.arc_offsets[u + 1]
        si, sf, ai, af = self._si, self._sf, self._ai, self._af
        return RawLattice(si[0][s0:s1], si[1][s0:s1], sf[1][s0:s1], ai[0][a0:a1], ai[1][a0:a1], ai[2][a0:a1], ai[3][a0:a1], af[0][a0:a1], af[1][a0:a1],
                          self._start, st_cost=sf[0][s0:s1])
    def __iter__(self): return (self[u] for u in range(len(self)))


class CudaDecoder:
    """Batched lattice decoder (cf. cuda_decoder::CudaDecoder, cudadecoder/cuda-decoder.h:224-345): nlanes utterances per call."""
    INFO = ("lat_states", "lat_arcs", "status", "reached_final", "tokens", "links", "max_frame_tokens", "emitting_arcs", "eps_arcs", "frames")
    def __init__(self, fst, config, nlanes, num_pdfs):
        self._L = _l.load(); self._h = ctypes.c_void_p(); self.fst = fst; self.config = config; self.nlanes = nlanes
        _l.check(self._L.k3_decoder_create(fst._h, ctypes.byref(config), nlanes, num_pdfs, ctypes.byref(self._h)))
        self._n = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._L.k3_decoder_destroy(self._h); self._h.value = None
        except Exception:      # interpreter shutdown
            pass

    def DecodeBatch(self, loglikes, row_offsets):
        """loglikes: float32 [rows x >= num_pdfs] on the GPU; utterance u = rows row_offsets[u]..row_offsets[u+1] (host ints).
        Asynchronous on the current stream."""
        assert loglikes.is_cuda and loglikes.dtype == torch.float32 and loglikes.stride(1) == 1
        ro = np.ascontiguousarray(row_offsets, np.int64); self._n = ro.size - 1; self._n_sel = None
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _l.check(self._L.k3_decoder_decode_batch(self._h, self._n, loglikes.data_ptr(), loglikes.stride(0), ro.ctypes.data, st))

    def InitDecoding(self, num_utts, max_total_frames):
        self._n = num_utts
        _l.check(self._L.k3_decoder_init_decoding(self._h, num_utts, int(max_total_frames), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def AdvanceDecoding(self, loglikes, row_offsets):
        """the next chunk of frames for every lane: lane u gets rows row_offsets[u]..row_offsets[u+1] of `loglikes` (may be empty)"""
        assert loglikes.is_cuda and loglikes.dtype == torch.float32 and loglikes.stride(1) == 1
        ro = np.ascontiguousarray(row_offsets, np.int64); assert ro.size - 1 == self._n
        _l.check(self._L.k3_decoder_advance_decoding(self._h, self._n, loglikes.data_ptr(), loglikes.stride(0), ro.ctypes.data, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def FinalizeDecoding(self):
        self._n_sel = None
        _l.check(self._L.k3_decoder_finalize_decoding(self._h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def InitChannels(self, channels):
        """restart the listed lanes of the current group (CudaDecoder::InitDecoding(channels)); the others keep decoding"""
        ch = np.ascontiguousarray(channels, np.int32)
        _l.check(self._L.k3_decoder_init_channels(self._h, ch.ctypes.data, ch.size, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def FinalizeChannels(self, channels):
        """FinalizeDecoding on the listed lanes only; LatticeInfo / GetRawLattices then return these lanes, in this order"""
        ch = np.ascontiguousarray(channels, np.int32); self._n_sel = int(ch.size)
        _l.check(self._L.k3_decoder_finalize_channels(self._h, ch.ctypes.data, ch.size, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def NumFramesDecoded(self, utt): return self._L.k3_decoder_num_frames_decoded(self._h, utt)

    def LatticeInfo(self, check=True):
        """int64 [num_utts x 10], columns = CudaDecoder.INFO (synchronises)"""
        info = np.zeros((getattr(self, "_n_sel", None) or self._n, 10), np.int64)
        rc = self._L.k3_decoder_lattice_info(self._h, info.ctypes.data)
        if check: _l.check(rc)
        return info

    def GetBestPath(self, channels=None, use_final_probs=True):
        """one-best path per channel from the tokens the lanes hold now (CudaDecoder::GetBestPath; with use_final_probs=False in the middle of an
        utterance: the traceback of GetPartialHypothesis).  Returns a list of dicts: ilabels (transition-ids, eps arcs included as 0), olabels
        (words, eps removed), graph, acoustic (sums), final_cost, relative_cost (FinalRelativeCost), reached_final."""
        ch = np.ascontiguousarray(range(self._n) if channels is None else channels, np.int32); n = ch.size
        cap = int(sum(4 * max(1, self.NumFramesDecoded(int(c))) + 64 for c in ch))
        off = np.zeros(n + 1, np.int64); il = np.zeros(cap, np.int32); ol = np.zeros(cap, np.int32); g = np.zeros(cap, np.float32); ac = np.zeros(cap, np.float32)
        fc = np.zeros(n, np.float32); rc = np.zeros(n, np.float32); rf = np.zeros(n, np.int32)
        _l.check(self._L.k3_decoder_get_best_path(self._h, ch.ctypes.data, n, int(bool(use_final_probs)), off.ctypes.data, cap, il.ctypes.data, ol.ctypes.data, g.ctypes.data, ac.ctypes.data,
                                                  fc.ctypes.data, rc.ctypes.data, rf.ctypes.data))
        out = []
        for u in range(n):
            a, b = off[u], off[u + 1]
            out.append(dict(ilabels=[int(x) for x in il[a:b] if x], olabels=[int(x) for x in ol[a:b] if x], graph=float(g[a:b].astype(np.float64).sum()) + float(fc[u]), acoustic=float(ac[a:b].astype(np.float64).sum()),
                            final_cost=float(fc[u]), relative_cost=float(rc[u]), reached_final=bool(rf[u]), num_arcs=int(b - a)))
        return out

    def OrderSensitiveEvents(self):
        """int64 per finalised utterance: literal_order -> forward links that exist only because next_cutoff was still loose when
        their arc was examined; default mode -> emitting arcs below the pre-pass bound but not below the final bound (an upper bound)"""
        ev = np.zeros(getattr(self, "_n_sel", None) or self._n, np.int64)
        _l.check(self._L.k3_decoder_order_sensitive_events(self._h, ev.ctypes.data)); return ev

    def PoolGrowths(self):
        """int32 per finalised utterance: how often its lane has moved to bigger token / link pools (lane_tokens_cap / lane_links_cap are a reservation, like the
        reference's ntokens_pre_allocated; the growth happens inside the token-passing kernel, from the decoder's spare arena)"""
        g = np.zeros(getattr(self, "_n_sel", None) or self._n, np.int32)
        _l.check(self._L.k3_decoder_pool_growths(self._h, g.ctypes.data)); return g

    def GetRawLattices(self, copy=False):
        """sequence of RawLattice, one per utterance of the last batch (GetRawLattice, not yet Connect()-ed).  All lattices arrive in
        ten flat host arrays (one D2H copy each); the per-utterance RawLattice objects are views made on access."""
        info = self.LatticeInfo()
        so = np.concatenate([[0], np.cumsum(info[:, 0])]); ao = np.concatenate([[0], np.cumsum(info[:, 1])])
        NS, NA = int(so[-1]), int(ao[-1])
        # ten arrays back to back in one page-locked host buffer: k3_decoder_get_raw_lattices then needs a single D2H copy.  Two
        # buffers alternate, so the lattices of a batch stay valid until the call after the next one (copy=True detaches them).
        n32 = 4 * NS + 6 * NA
        slot = self._lat_slot = 1 - getattr(self, "_lat_slot", 1)
        bufs = self.__dict__.setdefault("_lat_bufs", [None, None])
        if bufs[slot] is None or bufs[slot].numel() < max(n32, 1):
            bufs[slot] = torch.empty(max(n32 + n32 // 4, 1024), dtype=torch.int32, pin_memory=True)
        flat = bufs[slot].numpy()[:max(n32, 1)]
        if copy: flat = np.empty(max(n32, 1), np.int32)
        cuts = np.cumsum([0] + [NS] * 4 + [NA] * 6)
        parts = [flat[cuts[i]:cuts[i + 1]] for i in range(10)]
        si = [parts[0], parts[1]]; sf = [parts[2].view(np.float32), parts[3].view(np.float32)]
        ai = [parts[4], parts[5], parts[6], parts[7]]; af = [parts[8].view(np.float32), parts[9].view(np.float32)]
        _l.check(self._L.k3_decoder_get_raw_lattices(self._h, si[0].ctypes.data, si[1].ctypes.data, sf[0].ctypes.data, sf[1].ctypes.data,
                                                     ai[0].ctypes.data, ai[1].ctypes.data, ai[2].ctypes.data, ai[3].ctypes.data, af[0].ctypes.data, af[1].ctypes.data))
        return RawLatticeBatch(so, ao, si, sf, ai, af, self.fst.start)

    def FramePathCounts(self):
        """literal_order = 1: frames since the last call by path -- on the LDS-resident path (k3_decoder_fast.h), given up there and redone, on the general path (redone ones included) --
        and why the LDS path gave frames up (k3_decoder_phase_cycles of the shipped library; reading resets the counters)"""
        c = np.zeros(16, np.int64); _l.check(self._L.k3_decoder_phase_cycles(self._h, c.ctypes.data))
        names = ("tokens", "table", "hash", "labels", "worklist", "eps_links", "degree", "closure", "queue", "stack", "mismatch")
        return dict(lds_path=int(c[12]), given_up=int(c[13]), general_path=int(c[14]), give_up_reasons={n: int(v) for n, v in zip(names, c[:11]) if v},
                    cycles_lds_path=int(c[15]), cycles_general_path=int(c[11]))      # shader cycles summed over the lanes (general: a given-up attempt on the LDS path included)

    def SetProfiling(self, on=True):
        _l.check(self._L.k3_decoder_set_profiling(self._h, int(on)))

    def StreamWaitTokenPassing(self, stream):
        """`stream` (torch.cuda.Stream) waits on the device for this decoder's latest token-passing launch -- the last reader of the log-likelihoods it was given; what a pipelined
        caller puts in front of the kernels that refill that buffer (the pruning kernel behind the launch works on the lane pools only)"""
        _l.check(self._L.k3_decoder_stream_wait_token_passing(self._h, ctypes.c_void_p(stream.cuda_stream)))

    def KernelTimes(self):
        """(token-passing kernel ms, lattice-pruning kernel ms) of the last batch, HIP events on the launch stream"""
        ms = np.zeros(2, np.float32); _l.check(self._L.k3_decoder_kernel_times(self._h, ms.ctypes.data)); return float(ms[0]), float(ms[1])

    def algorithmic_bytes(self, info=None):
        """32 B per emitting arc traversed + 28 B per epsilon arc traversed + 16 B per token per frame"""
        info = self.LatticeInfo() if info is None else info
        return 32.0 * info[:, 7].sum() + 28.0 * info[:, 8].sum() + 16.0 * info[:, 4].sum()

    def FrameStats(self, utt, num_frames=None):
        nd = self.NumFramesDecoded(utt)
        assert num_frames is None or num_frames == nd, (num_frames, nd)      # the C side writes NumFramesDecoded(utt) elements
        num_frames = nd
        nt = np.zeros(num_frames, np.int32); f = [np.zeros(num_frames, np.float32) for _ in range(4)]
        _l.check(self._L.k3_decoder_frame_stats(self._h, utt, nt.ctypes.data, f[0].ctypes.data, f[1].ctypes.data, f[2].ctypes.data, f[3].ctypes.data))
        return dict(ntoks=nt, cur_cutoff=f[0], adaptive_beam=f[1], next_cutoff=f[2], cost_offset=f[3])
