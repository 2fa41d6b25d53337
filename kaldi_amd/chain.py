"""kaldi_amd.chain -- host-side mirror of the reference's chain-training classes for the part that is built: the LF-MMI denominator.
`DenominatorGraph(fst, num_pdfs)` = chain::DenominatorGraph (chain/chain-den-graph.h:60-143), `DenominatorComputation(opts, den_graph,
num_sequences, nnet_output)` with `Forward()` / `Backward(deriv_weight, nnet_output_deriv)` = chain::DenominatorComputation
(chain/chain-denominator.h:203-318).  ctypes plumbing over libk3hip.so (include/k3hip.h: k3_chain_den_*); torch only owns the device memory."""
import ctypes
import numpy as np
import torch
from . import lib as _l

class ChainTrainingOptions:
    """chain::ChainTrainingOptions (chain/chain-training.h:45-104); apply_out_of_range_penalty replaces the reference's coin flip (chain-training.cc:273)"""
    def __init__(self, leaky_hmm_coefficient=1.0e-05, l2_regularize=0.0, out_of_range_regularize=0.01, apply_out_of_range_penalty=False):
        self.leaky_hmm_coefficient = float(leaky_hmm_coefficient); self.l2_regularize = float(l2_regularize); self.out_of_range_regularize = float(out_of_range_regularize)
        self.apply_out_of_range_penalty = bool(apply_out_of_range_penalty)
    def _c(self): return _l.ChainTrainingOpts(self.l2_regularize, self.out_of_range_regularize, self.leaky_hmm_coefficient, int(self.apply_out_of_range_penalty))

class DenominatorGraph:
    def __init__(self, fst, num_pdfs):
        """fst: kaldi_amd.fst.Fst whose ilabels are pdf-id + 1 (the output of the reference's chain-make-den-fst)"""
        L = _l.load()
        off = np.ascontiguousarray(fst.arc_offsets, np.int64); il = np.ascontiguousarray(fst.ilabel, np.int32); nx = np.ascontiguousarray(fst.nextstate, np.int32)
        w = np.ascontiguousarray(fst.weight, np.float32); fin = np.ascontiguousarray(fst.final, np.float32)
        h = ctypes.c_void_p()
        _l.check(L.k3_chain_den_create(int(fst.num_states), int(fst.start), int(num_pdfs), off.ctypes.data, il.ctypes.data, nx.ctypes.data, w.ctypes.data, fin.ctypes.data, ctypes.byref(h)))
        self._h = h; self.num_states = int(fst.num_states); self.num_pdfs = int(num_pdfs)
    def NumStates(self): return self.num_states
    def NumPdfs(self): return self.num_pdfs
    def InitialProbs(self):
        p = np.zeros(self.num_states, np.float32); _l.check(_l.load().k3_chain_den_initial_probs(self._h, p.ctypes.data)); return p
    def __del__(self):
        if getattr(self, "_h", None) and _l is not None: _l.load().k3_chain_den_destroy(self._h); self._h = None

class DenominatorComputation:
    """nnet_output: [frames_per_sequence * num_sequences, num_pdfs] float32 on the GPU, row t * num_sequences + s."""
    def __init__(self, opts, den_graph, num_sequences, nnet_output):
        assert nnet_output.dtype == torch.float32 and nnet_output.is_cuda and nnet_output.shape[1] == den_graph.num_pdfs and nnet_output.stride(1) == 1
        assert nnet_output.shape[0] % num_sequences == 0
        self.opts, self.g, self.B, self.out = opts, den_graph, int(num_sequences), nnet_output
        self.T = nnet_output.shape[0] // self.B; self._objf = None; self._ok = True
    def _run(self, deriv_weight, deriv):
        objf = ctypes.c_float(0.0); ok = ctypes.c_int32(1)
        _l.check(_l.load().k3_chain_den_forward_backward(self.g._h, self.out.data_ptr(), self.out.stride(0), self.B, self.T, self.opts.leaky_hmm_coefficient, float(deriv_weight),
                                                         deriv.data_ptr() if deriv is not None else None, deriv.stride(0) if deriv is not None else 0, ctypes.byref(objf), ctypes.byref(ok),
                                                         torch.cuda.current_stream().cuda_stream))
        return objf.value, bool(ok.value)
    def Forward(self):
        """total log-probability of the minibatch over the denominator graph"""
        self._objf, _ = self._run(0.0, None); return self._objf
    def Backward(self, deriv_weight, nnet_output_deriv):
        """nnet_output_deriv += deriv_weight * occupation probabilities; returns False when the minibatch should be abandoned (alpha-beta check).
        (The forward pass is repeated inside the same launch: the kernel keeps no state between calls.)"""
        assert nnet_output_deriv.shape == self.out.shape and nnet_output_deriv.stride(1) == 1 and nnet_output_deriv.is_cuda
        self._objf, ok = self._run(deriv_weight, nnet_output_deriv); return ok


class Supervision:
    """chain::Supervision (chain/chain-supervision.h:226-330) for a minibatch, with the sequences' FSTs kept apart: fsts = one kaldi_amd.fst.Fst per
    sequence (start state 0, labels pdf-id + 1, states sorted by path length, every path frames_per_sequence arcs long)."""
    def __init__(self, fsts, frames_per_sequence, label_dim, weight=1.0, e2e=False):
        """e2e: the FSTs are Supervision::e2e_fsts (end-to-end / flat-start training: self-loops, several final states; chain-supervision.h:255-282) -> GenericNumeratorComputation"""
        L = _l.load(); self.e2e = bool(e2e); self.num_sequences = len(fsts); self.frames_per_sequence = int(frames_per_sequence); self.label_dim = int(label_dim); self.weight = float(weight)
        so = np.concatenate([[0], np.cumsum([f.num_states for f in fsts])]).astype(np.int32)
        ao = np.concatenate([[0]] + [np.asarray(f.arc_offsets[1:], np.int64) + b for f, b in zip(fsts, np.concatenate([[0], np.cumsum([int(f.arc_offsets[-1]) for f in fsts])[:-1]]))]).astype(np.int64)
        il = np.concatenate([f.ilabel for f in fsts]).astype(np.int32); nx = np.concatenate([f.nextstate for f in fsts]).astype(np.int32)
        w = np.concatenate([f.weight for f in fsts]).astype(np.float32); fin = np.concatenate([f.final for f in fsts]).astype(np.float32)
        h = ctypes.c_void_p()
        _l.check((L.k3_chain_supervision_create_e2e if self.e2e else L.k3_chain_supervision_create)(self.num_sequences, self.frames_per_sequence, self.label_dim, self.weight, so.ctypes.data, ao.ctypes.data, il.ctypes.data, nx.ctypes.data, w.ctypes.data, fin.ctypes.data, ctypes.byref(h)))
        self._h = h
    def __del__(self):
        if getattr(self, "_h", None) and _l is not None: _l.load().k3_chain_supervision_destroy(self._h); self._h = None

class NumeratorComputation:
    """chain::NumeratorComputation (chain/chain-numerator.h:63-146): Forward() = weight * log-prob of the supervision; Backward(deriv) adds weight * occupation probabilities"""
    def __init__(self, supervision, nnet_output):
        assert nnet_output.shape == (supervision.num_sequences * supervision.frames_per_sequence, supervision.label_dim) and nnet_output.is_cuda and nnet_output.stride(1) == 1
        self.sup, self.out = supervision, nnet_output
    def _run(self, deriv):
        v = ctypes.c_float(0.0)
        _l.check(_l.load().k3_chain_numerator(self.sup._h, self.out.data_ptr(), self.out.stride(0), deriv.data_ptr() if deriv is not None else None, deriv.stride(0) if deriv is not None else 0, ctypes.byref(v), torch.cuda.current_stream().cuda_stream))
        return v.value
    def Forward(self): return self._run(None)
    def Backward(self, nnet_output_deriv): self._run(nnet_output_deriv)

def ComputeChainObjfAndDeriv(opts, den_graph, supervision, nnet_output, nnet_output_deriv=None, xent_output_deriv=None):
    """chain::ComputeChainObjfAndDeriv (chain/chain-training.cc:242-337); returns (objf, l2_term, weight); the derivative matrices (optional) are overwritten"""
    objf, l2, wt = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0); c = opts._c()
    p = lambda m: (m.data_ptr(), m.stride(0)) if m is not None else (None, 0)
    _l.check(_l.load().k3_chain_objf_and_deriv(den_graph._h, supervision._h, ctypes.byref(c), nnet_output.data_ptr(), nnet_output.stride(0), *p(nnet_output_deriv), *p(xent_output_deriv),
                                               ctypes.byref(objf), ctypes.byref(l2), ctypes.byref(wt), torch.cuda.current_stream().cuda_stream))
    return objf.value, l2.value, wt.value
