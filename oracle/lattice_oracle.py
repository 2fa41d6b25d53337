"""oracle/lattice_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
ctypes front-end of oracle/lattice_faster_oracle.cc (the restated LatticeFasterDecoder; pinned to the reference's own decoder source, see its
header).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))

class _Fst(ctypes.Structure):
    _fields_ = [("num_states", ctypes.c_int32), ("start", ctypes.c_int32), ("arc_offsets", ctypes.c_void_p), ("ilabel", ctypes.c_void_p),
                ("olabel", ctypes.c_void_p), ("nextstate", ctypes.c_void_p), ("weight", ctypes.c_void_p), ("final_cost", ctypes.c_void_p)]

class Config(ctypes.Structure):
    """LatticeFasterDecoderConfig (decoder/lattice-faster-decoder.h:37-107) with its defaults."""
    _fields_ = [("beam", ctypes.c_float), ("max_active", ctypes.c_int32), ("min_active", ctypes.c_int32), ("lattice_beam", ctypes.c_float),
                ("prune_interval", ctypes.c_int32), ("beam_delta", ctypes.c_float), ("hash_ratio", ctypes.c_float), ("prune_scale", ctypes.c_float)]
    def __init__(self, beam=16.0, max_active=2**31 - 1, min_active=200, lattice_beam=10.0, prune_interval=25, beam_delta=0.5, hash_ratio=2.0, prune_scale=0.1):
        super().__init__(beam, max_active, min_active, lattice_beam, prune_interval, beam_delta, hash_ratio, prune_scale)

_lib = None
def _load():
    global _lib
    if _lib is None:
        from . import build as ob
        ob.build(with_ref=False)
        L = ctypes.CDLL(os.path.join(HERE, "libk3oracle_dec.so"))
        L.k3o_lfd_decode.restype = ctypes.c_void_p
        L.k3o_lfd_decode.argtypes = [ctypes.POINTER(_Fst), ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(Config), ctypes.c_int32]
        for n in ("k3o_lfd_sizes", "k3o_lfd_states", "k3o_lfd_arcs", "k3o_lfd_frame_stats", "k3o_lfd_free"):
            getattr(L, n).restype = None
        _lib = L
    return _lib

def decode(fst, loglikes, tid2pdf, cfg=None, mode=0):
    """fst: kaldi_amd.fst.Fst; loglikes [T x num_pdfs] float32 (acoustic scale already applied); returns
    (kaldi_amd.lattice.RawLattice BEFORE Connect, info dict)."""
    from kaldi_amd.lattice import RawLattice
    L = _load(); cfg = cfg or Config()
    ll = np.ascontiguousarray(loglikes, np.float32); t2p = np.ascontiguousarray(tid2pdf, np.int32)
    assert fst.ilabel.max() < t2p.size and t2p.max() < ll.shape[1]
    f = _Fst(fst.num_states, fst.start, fst.arc_offsets.ctypes.data, fst.ilabel.ctypes.data, fst.olabel.ctypes.data, fst.nextstate.ctypes.data,
             fst.weight.ctypes.data, fst.final.ctypes.data)
    h = ctypes.c_void_p(L.k3o_lfd_decode(ctypes.byref(f), ll.ctypes.data, ll.shape[0], ll.shape[1], t2p.ctypes.data, ctypes.byref(cfg), mode))
    try:
        sz = np.zeros(10, np.int64); L.k3o_lfd_sizes(h, sz.ctypes.data_as(ctypes.c_void_p))
        ns, na, nf = int(sz[0]), int(sz[1]), int(sz[2])
        fr, st = np.zeros(ns, np.int32), np.zeros(ns, np.int32); co, fc = np.zeros(ns, np.float32), np.zeros(ns, np.float32)
        L.k3o_lfd_states(h, fr.ctypes.data_as(ctypes.c_void_p), st.ctypes.data_as(ctypes.c_void_p), co.ctypes.data_as(ctypes.c_void_p), fc.ctypes.data_as(ctypes.c_void_p))
        a = [np.zeros(na, np.int32) for _ in range(4)] + [np.zeros(na, np.float32) for _ in range(2)]
        L.k3o_lfd_arcs(h, *[x.ctypes.data_as(ctypes.c_void_p) for x in a])
        nt = np.zeros(nf, np.int32); fs = [np.zeros(nf, np.float32) for _ in range(4)]
        L.k3o_lfd_frame_stats(h, nt.ctypes.data_as(ctypes.c_void_p), *[x.ctypes.data_as(ctypes.c_void_p) for x in fs])
    finally:
        L.k3o_lfd_free(h)
    lat = RawLattice(fr, st, fc, a[0], a[1], a[2], a[3], a[4], a[5], fst.start, st_cost=co)
    # order_sensitive_events: forward links (extra_links) / tokens (extra_toks) the literal algorithm created only because
    # next_cutoff was still loose when their arc was examined (tot >= the frame's final next_cutoff); 0 by construction in mode 1
    info = dict(order_sensitive_events=int(sz[3]) + int(sz[4]), replay_pops=int(sz[8]), replay_pushes=int(sz[9]), extra_links=int(sz[3]), extra_toks=int(sz[4]), best_ties=int(sz[5]), reached_final=bool(sz[6]), links_created=int(sz[7]),
                ntoks=nt, cur_cutoff=fs[0], adaptive_beam=fs[1], next_cutoff=fs[2], cost_offset=fs[3])
    return lat, info
