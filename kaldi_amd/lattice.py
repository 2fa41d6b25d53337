"""Raw (state-level) lattices as flat arrays: what LatticeFasterDecoder::GetRawLattice produces
(decoder/lattice-faster-decoder.cc:114-197) with each lattice state identified by its token (frame, HCLG state).
Host-side logic shared by the decoder front-end and the tests: Connect (trim), canonical form for comparison,
best path, and Kaldi text-format output."""
import numpy as np

class RawLattice:
    """states: st_frame, st_state (HCLG state), st_final (final cost, +inf = not final);
    arcs: arc_src, arc_dst (indices into the state arrays), arc_ilabel, arc_olabel, arc_graph, arc_ac
    (LatticeWeight(graph_cost, acoustic_cost)).  start = the state with (frame 0, fst start state)."""
    def __init__(self, st_frame, st_state, st_final, arc_src, arc_dst, arc_ilabel, arc_olabel, arc_graph, arc_ac, start_state, st_cost=None):
        self.st_frame = np.asarray(st_frame, np.int32); self.st_state = np.asarray(st_state, np.int32); self.st_final = np.asarray(st_final, np.float32)
        self.arc_src = np.asarray(arc_src, np.int32); self.arc_dst = np.asarray(arc_dst, np.int32)
        self.arc_ilabel = np.asarray(arc_ilabel, np.int32); self.arc_olabel = np.asarray(arc_olabel, np.int32)
        self.arc_graph = np.asarray(arc_graph, np.float32); self.arc_ac = np.asarray(arc_ac, np.float32)
        self.start_state = int(start_state); self.st_cost = None if st_cost is None else np.asarray(st_cost, np.float32)

    @property
    def num_states(self): return self.st_frame.size
    @property
    def num_arcs(self): return self.arc_src.size

    def keys(self):
        return (self.st_frame.astype(np.int64) << 32) | self.st_state.astype(np.int64)

    def start_index(self):
        i = np.nonzero((self.st_frame == 0) & (self.st_state == self.start_state))[0]
        return int(i[0]) if i.size else -1

    def connect(self):
        """fst::Connect as applied in DecodeUtteranceLatticeFaster (decoder/decoder-wrappers.cc:353): keep states that
        are accessible from the start state and co-accessible to a final state."""
        from scipy.sparse import csr_matrix
        from scipy.sparse.csgraph import breadth_first_order
        n = self.num_states
        if n == 0: return self
        s0 = self.start_index()
        if s0 < 0: return self._subset(np.zeros(n, bool))
        fwd = csr_matrix((np.ones(self.num_arcs, np.int8), (self.arc_src, self.arc_dst)), shape=(n, n))
        acc = np.zeros(n, bool); acc[breadth_first_order(fwd, s0, directed=True, return_predecessors=False)] = True
        fin = np.nonzero(np.isfinite(self.st_final))[0]
        rev = csr_matrix((np.ones(self.num_arcs + fin.size, np.int8), (np.concatenate([self.arc_dst, np.full(fin.size, n)]), np.concatenate([self.arc_src, fin]))), shape=(n + 1, n + 1))
        co = np.zeros(n + 1, bool); co[breadth_first_order(rev, n, directed=True, return_predecessors=False)] = True
        return self._subset(acc & co[:n])

    def _subset(self, keep):
        new = np.cumsum(keep) - 1
        ka = keep[self.arc_src] & keep[self.arc_dst] if self.num_arcs else np.zeros(0, bool)
        return RawLattice(self.st_frame[keep], self.st_state[keep], self.st_final[keep], new[self.arc_src[ka]], new[self.arc_dst[ka]],
                          self.arc_ilabel[ka], self.arc_olabel[ka], self.arc_graph[ka], self.arc_ac[ka], self.start_state,
                          None if self.st_cost is None else self.st_cost[keep])

    def canonical(self):
        """order-free form: states sorted by (frame, HCLG state) with their final-cost bits; arcs as rows
        (src key, dst key, ilabel, olabel, graph-cost bits, acoustic-cost bits) sorted lexicographically."""
        k = self.keys(); so = np.argsort(k, kind="stable")
        states = np.stack([k[so], (self.st_final[so] + np.float32(0)).view(np.int32).astype(np.int64)], axis=1)   # + 0: -0.0 -> +0.0
        rows = np.stack([k[self.arc_src], k[self.arc_dst], self.arc_ilabel.astype(np.int64), self.arc_olabel.astype(np.int64),
                         (self.arc_graph + np.float32(0)).view(np.int32).astype(np.int64), (self.arc_ac + np.float32(0)).view(np.int32).astype(np.int64)], axis=1) if self.num_arcs else np.zeros((0, 6), np.int64)
        ao = np.lexsort(rows.T[::-1]) if rows.shape[0] else np.zeros(0, np.int64)
        return states, rows[ao]

    def diff(self, other):
        """'' when the canonical forms are identical, else a short description of the first differences."""
        sa, aa = self.canonical(); sb, ab = other.canonical()
        msg = []
        if sa.shape != sb.shape or not np.array_equal(sa, sb):
            A = set(map(tuple, sa.tolist())); B = set(map(tuple, sb.tolist()))
            msg.append(f"states: {sa.shape[0]} vs {sb.shape[0]}; only-left {sorted(A - B)[:5]} only-right {sorted(B - A)[:5]}")
        if aa.shape != ab.shape or not np.array_equal(aa, ab):
            A = set(map(tuple, aa.tolist())); B = set(map(tuple, ab.tolist()))
            msg.append(f"arcs: {aa.shape[0]} vs {ab.shape[0]}; only-left {sorted(A - B)[:5]} only-right {sorted(B - A)[:5]}")
        return "; ".join(msg)

    def best_path(self):
        """ShortestPath over the lattice in the tropical sense on graph+acoustic (what GetBestPath feeds to
        GetLinearSymbolSequence, decoder-wrappers.cc:322-331): returns (ilabels, olabels without 0, graph, acoustic).
        The lattice is a DAG ordered by frame; epsilon arcs stay inside a frame, so relax frame by frame to a fixpoint."""
        n = self.num_states; s0 = self.start_index()
        if n == 0 or s0 < 0: return None
        tot = (self.arc_graph.astype(np.float64) + self.arc_ac.astype(np.float64))
        best = np.full(n, np.inf); best[s0] = 0.0; back = np.full(n, -1, np.int64)
        order = np.argsort(self.st_frame[self.arc_src], kind="stable")
        changed = True
        while changed:          # few sweeps: arcs sorted by source frame, only epsilon chains need more than one
            changed = False
            for a in order:
                s, d = self.arc_src[a], self.arc_dst[a]
                if best[s] + tot[a] < best[d]: best[d] = best[s] + tot[a]; back[d] = a; changed = True
        fin = np.nonzero(np.isfinite(self.st_final))[0]
        if fin.size == 0: return None
        end = fin[np.argmin(best[fin] + self.st_final[fin])]
        if not np.isfinite(best[end]): return None
        il, ol, g, ac = [], [], float(self.st_final[end]), 0.0
        s = end
        while s != s0:
            a = back[s]; il.append(int(self.arc_ilabel[a])); ol.append(int(self.arc_olabel[a])); g += float(self.arc_graph[a]); ac += float(self.arc_ac[a]); s = self.arc_src[a]
        return [i for i in il[::-1] if i], [o for o in ol[::-1] if o], g, ac


# ---- the host tail: determinization to a CompactLattice (include/k3host.h, kaldi_amd/host/k3_lattice.cc) -------------------------------
class TransitionInformation:
    """the transition-id -> phone / self-loop / phone-start map of a model file, which the phone-level determinization pass needs
    (kaldi::TransitionInformation as DeterminizeLatticeInsertPhones uses it)"""
    def __init__(self, model_rxfilename):
        import ctypes
        from . import hostlib as hl
        self._L = hl.load(); self._h = ctypes.c_void_p()
        hl.check(self._L.k3h_transitions_read(str(model_rxfilename).encode(), ctypes.byref(self._h)))
    def NumTransitionIds(self): return int(self._L.k3h_transitions_num_ids(self._h))
    def __del__(self):
        if getattr(self, "_h", None) and self._h.value: self._L.k3h_transitions_free(self._h); self._h.value = None


class CompactLattice:
    """kaldi::CompactLattice (lat/kaldi-lattice.h:46) as flat arrays: an acceptor over word labels; every arc and every final weight
    carries (graph cost, acoustic cost) and the transition-ids of the best alignment for that stretch (strings[off[k]:off[k+1]])."""
    def __init__(self, handle, lib):
        import ctypes
        self._L, self._h = lib, handle
        ns, na, nl = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        from . import hostlib as hl
        hl.check(lib.k3h_clat_sizes(handle, ctypes.byref(ns), ctypes.byref(na), ctypes.byref(nl)))
        ns, na, nl = ns.value, na.value, nl.value
        self.is_final = np.zeros(ns, np.uint8); self.final_graph = np.zeros(ns, np.float32); self.final_ac = np.zeros(ns, np.float32); self.final_str_off = np.zeros(ns + 1, np.int64)
        self.arc_src, self.arc_dst, self.arc_label = (np.zeros(na, np.int32) for _ in range(3)); self.arc_graph = np.zeros(na, np.float32); self.arc_ac = np.zeros(na, np.float32)
        self.arc_str_off = np.zeros(na + 1, np.int64); self.strings = np.zeros(max(nl, 1), np.int32)
        start = ctypes.c_int32()
        hl.check(lib.k3h_clat_get(handle, ctypes.byref(start), *[x.ctypes.data for x in (self.is_final, self.final_graph, self.final_ac, self.final_str_off, self.arc_src, self.arc_dst,
                                                                                            self.arc_label, self.arc_graph, self.arc_ac, self.arc_str_off, self.strings)]))
        self.start = start.value; self.strings = self.strings[:nl]
    @property
    def num_states(self): return self.is_final.size
    @property
    def num_arcs(self): return self.arc_src.size
    def arc_string(self, k): return self.strings[self.arc_str_off[k]:self.arc_str_off[k + 1]]
    def final_string(self, s): return self.strings[self.final_str_off[s]:self.final_str_off[s + 1]]
    def Write(self, key, wspecifier):
        """one record of a CompactLattice table: "ark:file" (binary) or "ark,t:file" (text)"""
        from . import hostlib as hl
        hl.check(self._L.k3h_clat_write(self._h, str(key).encode(), str(wspecifier).encode()))
    def ScaleAcoustic(self, scale):
        from . import hostlib as hl
        hl.check(self._L.k3h_clat_scale_acoustic(self._h, float(scale)))
        return CompactLattice._refresh(self)
    def _refresh(self):
        new = CompactLattice(self._h, self._L); self.__dict__.update({k: v for k, v in new.__dict__.items() if k not in ("_h", "_L")}); new._h = None
        return self
    def best_path(self):
        """(transition-ids, words, graph cost, acoustic cost) of the cheapest path"""
        n = self.num_states
        if n == 0 or self.start < 0: return None
        order = np.argsort(self.arc_src, kind="stable")
        best = np.full(n, np.inf); back = np.full(n, -1, np.int64); best[self.start] = 0.0
        for _ in range(2 if np.all(self.arc_dst > self.arc_src) else n):
            for a in order:
                c = best[self.arc_src[a]] + float(self.arc_graph[a]) + float(self.arc_ac[a])
                if c < best[self.arc_dst[a]]: best[self.arc_dst[a]] = c; back[self.arc_dst[a]] = a
        fin = np.nonzero(self.is_final)[0]
        if fin.size == 0 or not np.isfinite(best[fin]).any(): return None
        tot = best[fin] + self.final_graph[fin] + self.final_ac[fin]; s = int(fin[int(np.argmin(tot))])
        tids = list(self.final_string(s)); words = []; g = float(self.final_graph[s]); ac = float(self.final_ac[s])
        while s != self.start:
            a = int(back[s]); tids = list(self.arc_string(a)) + tids; words.insert(0, int(self.arc_label[a])); g += float(self.arc_graph[a]); ac += float(self.arc_ac[a]); s = int(self.arc_src[a])
        return [int(t) for t in tids], [w for w in words if w], g, ac
    def __del__(self):
        if getattr(self, "_h", None): self._L.k3h_clat_free(self._h); self._h = None


def _lattice_args(raw):
    import ctypes
    arrs = [np.ascontiguousarray(raw.st_final, np.float32), np.ascontiguousarray(raw.arc_src, np.int32), np.ascontiguousarray(raw.arc_dst, np.int32), np.ascontiguousarray(raw.arc_ilabel, np.int32),
            np.ascontiguousarray(raw.arc_olabel, np.int32), np.ascontiguousarray(raw.arc_graph, np.float32), np.ascontiguousarray(raw.arc_ac, np.float32)]
    start = raw.start_index()
    return arrs, [raw.num_states, max(start, 0), arrs[0].ctypes.data, raw.num_arcs] + [a.ctypes.data for a in arrs[1:]]


def DeterminizeLatticePhonePruned(raw, beam, trans=None, delta=None, max_mem=None, phone_determinize=True, word_determinize=True, minimize=False):
    """fst::DeterminizeLatticePhonePrunedWrapper on a RawLattice (trim it first, like the decoders: raw.connect()); with trans=None the
    word-level pass alone (fst::DeterminizeLatticePruned).  Returns (CompactLattice, reached_the_beam)."""
    import ctypes
    from . import hostlib as hl
    L = hl.load(); o = hl.DetOpts(); L.k3h_det_opts_default(ctypes.byref(o))
    if delta is not None: o.delta = delta
    if max_mem is not None: o.max_mem = max_mem
    o.phone_determinize, o.word_determinize, o.minimize = int(phone_determinize), int(word_determinize), int(minimize)
    keep, args = _lattice_args(raw); h = ctypes.c_void_p(); ok = ctypes.c_int32()
    if raw.num_states and raw.start_index() < 0: raise hl.K3HostError("the lattice has no start state (frame 0, graph start state)")
    hl.check(L.k3h_determinize_lattice(trans._h if trans is not None else None, *args, float(beam), ctypes.byref(o), ctypes.byref(h), ctypes.byref(ok)))
    return CompactLattice(h, L), bool(ok.value)


def ConvertLattice(raw):
    """fst::ConvertLattice(Lattice -> CompactLattice): no determinization, linear chains folded into single arcs"""
    import ctypes
    from . import hostlib as hl
    L = hl.load(); keep, args = _lattice_args(raw); h = ctypes.c_void_p()
    hl.check(L.k3h_convert_lattice(*args, ctypes.byref(h)))
    return CompactLattice(h, L)
