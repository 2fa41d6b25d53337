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
    def __init__(self, nnet, num_frames, frame_subsampling_factor=1, log_priors=None, acoustic_scale=1.0):
        self.nnet = nnet; self._L = nnet._L; self._h = ctypes.c_void_p()
        nf = np.ascontiguousarray(num_frames, dtype=np.int32)
        lp = None if log_priors is None else np.ascontiguousarray(log_priors, dtype=np.float32)
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

    def forward(self, feats, out=None):
        """feats: float32 [sum T_u, >= input_dim] on the GPU -> float32 [total_out_rows, output_dim]."""
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.stride(1) == 1
        if out is None:
            out = torch.empty((self.total_out_rows, self.nnet.info.output_dim), dtype=torch.float32, device=feats.device)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _l.check(self._L.k3_nnet_forward(self._h, feats.data_ptr(), feats.stride(0), out.data_ptr(), out.stride(0), st))
        return out
