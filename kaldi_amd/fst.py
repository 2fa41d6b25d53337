"""Host-side WFST container for the decoding graph (HCLG): a generic CSR over numpy arrays, the neutral form
between a graph source (OpenFst binary file, or the synthetic generator in synth.py) and the device CSR that
k3_fst_create() builds (cf. CudaFst, cudadecoder/cuda-fst.{h,cc}: OpenFst -> CSR + upload)."""
import numpy as np

class Fst:
    """arcs of state s = [arc_offsets[s], arc_offsets[s+1]) in FST order; ilabel 0 = epsilon (non-emitting);
    final[s] = final cost (tropical weight), +inf when s is not final."""
    def __init__(self, start, arc_offsets, ilabel, olabel, weight, nextstate, final):
        self.start = int(start)
        self.arc_offsets = np.ascontiguousarray(arc_offsets, np.int32)
        self.ilabel = np.ascontiguousarray(ilabel, np.int32); self.olabel = np.ascontiguousarray(olabel, np.int32)
        self.weight = np.ascontiguousarray(weight, np.float32); self.nextstate = np.ascontiguousarray(nextstate, np.int32)
        self.final = np.ascontiguousarray(final, np.float32)
        assert self.arc_offsets.size == self.final.size + 1 and self.arc_offsets[-1] == self.ilabel.size
        assert 0 <= self.start < self.num_states

    @property
    def num_states(self): return self.final.size
    @property
    def num_arcs(self): return self.ilabel.size

    @staticmethod
    def from_arcs(num_states, start, src, ilabel, olabel, weight, dst, final):
        """build from an unsorted arc list (stable in the given order within a state)."""
        src = np.asarray(src, np.int64)
        order = np.argsort(src, kind="stable")
        off = np.zeros(num_states + 1, np.int64); np.add.at(off, src + 1, 1); off = np.cumsum(off)
        return Fst(start, off, np.asarray(ilabel)[order], np.asarray(olabel)[order], np.asarray(weight)[order], np.asarray(dst)[order], final)

    def stats(self):
        e = int((self.ilabel != 0).sum()); deg = np.diff(self.arc_offsets)
        return dict(states=self.num_states, arcs=self.num_arcs, emitting=e, epsilon=self.num_arcs - e, max_degree=int(deg.max()),
                    mean_degree=float(deg.mean()), finals=int(np.isfinite(self.final).sum()), olabel_arcs=int((self.olabel != 0).sum()))
