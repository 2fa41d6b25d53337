"""Host-side WFST container for the decoding graph (HCLG): a generic CSR over numpy arrays, the neutral form
between a graph source (OpenFst binary file, or the synthetic generator in synth.py) and the device CSR that
k3_fst_create() builds (cf. CudaFst, cudadecoder/cuda-fst.{h,cc}: OpenFst -> CSR + upload)."""
import struct
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

    # ---- OpenFst binary container (what fstcompile / Kaldi's mkgraph write: fst::VectorFst<StdArc>::Write).  Layout restated from
    # the OpenFst 1.8 sources' documented format (FstHeader: magic 2125659606, fst type, arc type, version, flags, properties,
    # start, numstates, numarcs; vector body: per state final weight, arc count, arcs {ilabel, olabel, weight, nextstate}).
    # OpenFst is not available in this environment, so reader and writer are checked against each other only.
    MAGIC = 2125659606
    def write_openfst(self, path):
        with open(path, "wb") as f:
            def wstr(s_): f.write(struct.pack("<i", len(s_)) + s_.encode())
            f.write(struct.pack("<i", Fst.MAGIC)); wstr("vector"); wstr("standard")
            f.write(struct.pack("<iiQqqq", 2, 0, 0x0000000000000003, self.start, self.num_states, self.num_arcs))   # version 2, no symbols, props: expanded|mutable
            rec = np.zeros(self.num_arcs, dtype=[("il", "<i4"), ("ol", "<i4"), ("w", "<f4"), ("n", "<i4")])
            rec["il"], rec["ol"], rec["w"], rec["n"] = self.ilabel, self.olabel, self.weight, self.nextstate
            raw = rec.tobytes(); off = self.arc_offsets
            for s_ in range(self.num_states):
                f.write(struct.pack("<fq", float(self.final[s_]), int(off[s_ + 1] - off[s_]))); f.write(raw[16 * off[s_]:16 * off[s_ + 1]])

    @staticmethod
    def read_openfst(path):
        b = open(path, "rb").read(); pos = 0
        def rd(fmt):
            nonlocal pos
            v = struct.unpack_from(fmt, b, pos); pos += struct.calcsize(fmt); return v
        def rstr():
            nonlocal pos
            n, = rd("<i"); s_ = b[pos:pos + n].decode(); pos += n; return s_
        magic, = rd("<i"); assert magic == Fst.MAGIC, "not an OpenFst binary file"
        ftype, atype = rstr(), rstr(); assert ftype == "vector" and atype == "standard", (ftype, atype)
        version, flags, props, start, ns, na = rd("<iiQqqq")
        assert flags == 0, "symbol tables inside the FST file are not supported"
        off = [0]; il, ol, w, n, fin = [], [], [], [], []
        for _ in range(ns):
            fc, cnt = rd("<fq"); fin.append(fc)
            rec = np.frombuffer(b, dtype=[("il", "<i4"), ("ol", "<i4"), ("w", "<f4"), ("n", "<i4")], count=cnt, offset=pos); pos += 16 * cnt
            il.append(rec["il"]); ol.append(rec["ol"]); w.append(rec["w"]); n.append(rec["n"]); off.append(off[-1] + cnt)
        cat = lambda x, dt: np.concatenate(x) if x else np.zeros(0, dt)
        return Fst(start, off, cat(il, np.int32), cat(ol, np.int32), cat(w, np.float32), cat(n, np.int32), np.array(fin, np.float32))
