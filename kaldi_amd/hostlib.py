"""ctypes plumbing for include/k3host.h (kaldi_amd/lib/libk3host.so): the host tail of the path -- lattice determinization and
CompactLattice output.  No GPU involved; fails loudly when the library is missing (build: make -C kaldi_amd/host)."""
import ctypes, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("K3HOST_LIB", os.path.join(HERE, "lib", "libk3host.so"))

class K3HostError(RuntimeError): pass

class DetOpts(ctypes.Structure):
    """fst::DeterminizeLatticePhonePrunedOptions"""
    _fields_ = [("delta", ctypes.c_float), ("max_mem", ctypes.c_int32), ("phone_determinize", ctypes.c_int32), ("word_determinize", ctypes.c_int32), ("minimize", ctypes.c_int32)]

_lib = None
def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH): raise K3HostError(f"{LIB_PATH} not found: build it with `make -C kaldi_amd/host` (or __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        L.k3h_last_error.restype = ctypes.c_char_p
        L.k3h_transitions_num_ids.restype = ctypes.c_int32
        for n in ("k3h_transitions_free", "k3h_clat_free", "k3h_det_opts_default"): getattr(L, n).restype = None
        P = ctypes.c_void_p
        L.k3h_transitions_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(P)]
        L.k3h_transitions_num_ids.argtypes = [P]; L.k3h_transitions_free.argtypes = [P]; L.k3h_clat_free.argtypes = [P]
        lat = [ctypes.c_int32, ctypes.c_int32, P, ctypes.c_int64, P, P, P, P, P, P]
        L.k3h_determinize_lattice.argtypes = [P] + lat + [ctypes.c_double, ctypes.POINTER(DetOpts), ctypes.POINTER(P), ctypes.POINTER(ctypes.c_int32)]
        L.k3h_convert_lattice.argtypes = lat + [ctypes.POINTER(P)]
        L.k3h_postprocess_batch.argtypes = [P, ctypes.c_int32, P, P, ctypes.c_int32] + [P] * 9 + [ctypes.c_double, ctypes.POINTER(DetOpts), ctypes.c_int32, P, P, P, P]
        L.k3h_clat_sizes.argtypes = [P, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.k3h_clat_get.argtypes = [P, ctypes.POINTER(ctypes.c_int32)] + [P] * 11
        L.k3h_clat_scale_acoustic.argtypes = [P, ctypes.c_double]
        L.k3h_clat_write.argtypes = [P, ctypes.c_char_p, ctypes.c_char_p]
        L.k3h_lattice_table_to_ctm_model.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_float, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int64]; L.k3h_lattice_table_to_ctm_model.restype = ctypes.c_int64
        L.k3h_lattice_table_to_ctm.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_float, ctypes.c_char_p, ctypes.c_int64]; L.k3h_lattice_table_to_ctm.restype = ctypes.c_int64
        L.k3h_ivector_config_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(P)]; L.k3h_ivector_config_free.argtypes = [P]; L.k3h_ivector_config_free.restype = None
        L.k3h_ivector_config_get.argtypes = [P, P, P] + [ctypes.POINTER(P)] * 7
        _lib = L
    return _lib

def check(rc):
    if rc != 0: raise K3HostError(load().k3h_last_error().decode())


def last_error():
    return load().k3h_last_error().decode()
