"""Order-free identity of a raw lattice whose states have no names: forward and backward structural signatures.
The reference decoder's tokens do not remember their graph state, so its GetRawLattice output can only be compared with the
oracle's up to a renaming of the states.  sig_b(s) hashes the final weight of s and the sorted (arc payload, sig_b(target)) of its
arcs; sig_f(s) hashes the sorted (arc payload, sig_f(source)) of the arcs into s.  Two lattices with the same multiset of
(frame, sig_f, sig_b) per state and of (sig_f(src), payload, sig_b(dst)) per arc have the same paths with the same float bits and
the same sharing of states between them."""
import hashlib
import numpy as np


def _sigs(n, start, final_bits, src, dst, labels):
    out = [[] for _ in range(n)]; inn = [[] for _ in range(n)]; indeg = [0] * n
    for a, (s, d) in enumerate(zip(src, dst)): out[s].append(a); inn[d].append(a); indeg[d] += 1
    order = []; ready = [s for s in range(n) if indeg[s] == 0]
    while ready:
        s = ready.pop(); order.append(s)
        for a in out[s]:
            indeg[dst[a]] -= 1
            if indeg[dst[a]] == 0: ready.append(dst[a])
    assert len(order) == n, "the lattice has a cycle"
    H = lambda x: hashlib.blake2b(repr(x).encode(), digest_size=12).hexdigest()
    bwd = [None] * n; fwd = [None] * n
    for s in reversed(order): bwd[s] = H((final_bits[s], sorted((labels[a], bwd[dst[a]]) for a in out[s])))
    for s in order: fwd[s] = H((s == start, sorted((labels[a], fwd[src[a]]) for a in inn[s])))
    return fwd, bwd


def _bits(x): return (np.asarray(x, np.float32) + np.float32(0)).view(np.int32).tolist()      # + 0: -0.0 -> +0.0


def canonical(frame, start, final_graph, final_ac, src, dst, ilabel, olabel, graph, ac):
    lab = list(zip(np.asarray(ilabel).tolist(), np.asarray(olabel).tolist(), _bits(graph), _bits(ac)))
    fb = list(zip(_bits(final_graph), _bits(final_ac)))
    src, dst = np.asarray(src).tolist(), np.asarray(dst).tolist()
    f, b = _sigs(len(fb), int(start), fb, src, dst, lab)
    return sorted(zip(np.asarray(frame).tolist(), f, b)), sorted((f[s], l, b[d]) for s, d, l in zip(src, dst, lab))


def canonical_of_reference(r):
    """r: dict returned by oracle.ref_decoder.decode"""
    return canonical(r["frame"], r["start"], r["final_graph"], r["final_ac"], r["src"], r["dst"], r["ilabel"], r["olabel"], r["graph"], r["ac"])


def canonical_of_raw(lat):
    """lat: kaldi_amd.lattice.RawLattice (final weights are (cost, 0); non-final states (inf, inf) like LatticeWeight::Zero)"""
    fa = np.where(np.isfinite(lat.st_final), np.float32(0), np.float32(np.inf)).astype(np.float32)
    return canonical(lat.st_frame, lat.start_index(), lat.st_final, fa, lat.arc_src, lat.arc_dst, lat.arc_ilabel, lat.arc_olabel, lat.arc_graph, lat.arc_ac)


def digest(canon):
    return hashlib.blake2b(repr(canon).encode(), digest_size=16).hexdigest()
