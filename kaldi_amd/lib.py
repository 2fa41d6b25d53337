"""ctypes binding of libk3hip.so (include/k3hip.h).  Fails loudly when the library is missing."""
import ctypes, os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libk3hip.so")
if os.environ.get("K3HIP_LIB"):      # developer override (profiling builds of the same sources, tools/prof_*.py): never silent, and bench.py refuses to run with it
    import sys
    LIB_PATH = os.environ["K3HIP_LIB"]; print(f"kaldi_amd: DEVELOPER OVERRIDE K3HIP_LIB -> {LIB_PATH}", file=sys.stderr)

class K3Error(RuntimeError):
    """Raised for any non-zero k3_status (the C++ adapters raise KaldiFatalError instead)."""

class FeatOpts(ctypes.Structure):
    """k3_feat_opts (include/k3hip.h); defaults = FbankOptions (feat/feature-fbank.h:44-61)."""
    _fields_ = [("samp_freq", ctypes.c_float), ("frame_shift_ms", ctypes.c_float), ("frame_length_ms", ctypes.c_float),
                ("dither", ctypes.c_float), ("preemph_coeff", ctypes.c_float), ("blackman_coeff", ctypes.c_float),
                ("remove_dc_offset", ctypes.c_int32), ("round_to_power_of_two", ctypes.c_int32), ("snip_edges", ctypes.c_int32),
                ("window_type", ctypes.c_int32), ("num_bins", ctypes.c_int32),
                ("low_freq", ctypes.c_float), ("high_freq", ctypes.c_float), ("vtln_low", ctypes.c_float), ("vtln_high", ctypes.c_float),
                ("htk_mode", ctypes.c_int32), ("use_energy", ctypes.c_int32), ("energy_floor", ctypes.c_float),
                ("raw_energy", ctypes.c_int32), ("htk_compat", ctypes.c_int32), ("use_log_fbank", ctypes.c_int32), ("use_power", ctypes.c_int32),
                ("num_ceps", ctypes.c_int32), ("cepstral_lifter", ctypes.c_float), ("feature_type", ctypes.c_int32), ("vtln_warp", ctypes.c_float)]

class OnlineCmvnOpts(ctypes.Structure):      # k3_online_cmvn_opts
    _fields_ = [("cmn_window", ctypes.c_int32), ("speaker_frames", ctypes.c_int32), ("global_frames", ctypes.c_int32),
                ("normalize_mean", ctypes.c_int32), ("normalize_variance", ctypes.c_int32)]

class IvectorModel(ctypes.Structure):       # k3_ivector_model: host arrays
    _fields_ = [("feat_dim", ctypes.c_int32), ("lda_rows", ctypes.c_int32), ("lda_cols", ctypes.c_int32), ("num_gauss", ctypes.c_int32), ("ivector_dim", ctypes.c_int32),
                ("lda", ctypes.c_void_p), ("global_cmvn_stats", ctypes.c_void_p), ("gconsts", ctypes.c_void_p), ("means_invvars", ctypes.c_void_p), ("inv_vars", ctypes.c_void_p),
                ("M", ctypes.c_void_p), ("sigma_inv", ctypes.c_void_p), ("prior_offset", ctypes.c_double)]

class IvectorOpts(ctypes.Structure):        # k3_ivector_opts
    _fields_ = [("left_context", ctypes.c_int32), ("right_context", ctypes.c_int32), ("num_gselect", ctypes.c_int32), ("min_post", ctypes.c_float), ("posterior_scale", ctypes.c_float),
                ("max_count", ctypes.c_float), ("ivector_period", ctypes.c_int32), ("num_cg_iters", ctypes.c_int32), ("exact_solve", ctypes.c_int32),
                ("online_cmvn_iextractor", ctypes.c_int32), ("cmvn", OnlineCmvnOpts)]

class IvectorInfo(ctypes.Structure):        # k3_ivector_info
    _fields_ = [("feat_dim", ctypes.c_int32), ("lda_dim", ctypes.c_int32), ("num_gauss", ctypes.c_int32), ("ivector_dim", ctypes.c_int32), ("ivector_period", ctypes.c_int32)]

class NnetInfo(ctypes.Structure):
    """k3_nnet_info (include/k3hip.h)"""
    _fields_ = [("input_dim", ctypes.c_int32), ("output_dim", ctypes.c_int32), ("left_context", ctypes.c_int32), ("right_context", ctypes.c_int32),
                ("num_components", ctypes.c_int32), ("num_fused_nodes", ctypes.c_int32), ("has_priors", ctypes.c_int32), ("num_params", ctypes.c_int64),
                ("ivector_dim", ctypes.c_int32)]

class NnetStreamInfo(ctypes.Structure):
    """k3_nnet_stream_info (include/k3hip.h)"""
    _fields_ = [("num_channels", ctypes.c_int32), ("frames_per_chunk", ctypes.c_int32), ("subsampling", ctypes.c_int32), ("output_rows_per_pass", ctypes.c_int32),
                ("first_output_time", ctypes.c_int32), ("right_context", ctypes.c_int32), ("input_history", ctypes.c_int32), ("flops_per_pass", ctypes.c_double)]

class DecoderConfig(ctypes.Structure):
    """k3_decoder_config (include/k3hip.h); decoding fields = LatticeFasterDecoderConfig (decoder/lattice-faster-decoder.h:37-107)."""
    _fields_ = [("beam", ctypes.c_float), ("max_active", ctypes.c_int32), ("min_active", ctypes.c_int32), ("lattice_beam", ctypes.c_float),
                ("beam_delta", ctypes.c_float), ("frame_tokens_cap", ctypes.c_int32), ("frame_cands_cap", ctypes.c_int32),
                ("lane_tokens_cap", ctypes.c_int64), ("lane_links_cap", ctypes.c_int64), ("literal_order", ctypes.c_int32), ("hash_ratio", ctypes.c_float), ("fast_frame_tokens", ctypes.c_int32),
                ("spare_pool_bytes", ctypes.c_int64), ("resident_lanes", ctypes.c_int32), ("resident_exclusive", ctypes.c_int32)]

WINDOW_TYPES = {"hanning": 0, "sine": 1, "hamming": 2, "povey": 3, "rectangular": 4, "blackman": 5}

_lib = None

def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise K3Error(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950).  kaldi_amd has no CPU fallback.")
    # torch ships its own libamdhip64; load it FIRST so libk3hip.so binds to the same HIP runtime
    # (two runtimes in one process => "No HIP GPUs are available" in whichever initialises second).
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.k3_last_error.restype = ctypes.c_char_p
    L.k3_build_id.restype = ctypes.c_char_p
    L.k3_feat_plan_create.argtypes = [ctypes.POINTER(FeatOpts), ctypes.POINTER(vp)]
    L.k3_feat_plan_destroy.argtypes = [vp]; L.k3_feat_plan_destroy.restype = None
    L.k3_feat_dim.argtypes = [vp]; L.k3_feat_dim.restype = i32
    L.k3_feat_num_frames.argtypes = [vp, i64]; L.k3_feat_num_frames.restype = i32
    L.k3_feat_compute_batch.argtypes = [vp, vp, vp, vp, i32, i64, vp, i64, vp]; L.k3_feat_compute_batch_pcm16.argtypes = [vp, vp, vp, vp, i32, i64, vp, i64, vp]
    L.k3_resample_num_samples.argtypes = [i32, i32, i64]; L.k3_resample_num_samples.restype = i64; L.k3_resample_batch.argtypes = [i32, i32, vp, vp, i32, vp, vp, vp]
    L.k3_cmvn_offline_batch.argtypes = [vp, i64, i32, vp, i32, i32, vp, vp]
    L.k3_online_cmvn_opts_default.argtypes = [ctypes.POINTER(OnlineCmvnOpts)]; L.k3_online_cmvn_opts_default.restype = None
    L.k3_cmvn_online_batch.argtypes = [vp, i64, vp, i64, i32, vp, i32, ctypes.POINTER(OnlineCmvnOpts), vp, vp, vp, i32, vp]
    L.k3_ivector_opts_default.argtypes = [ctypes.POINTER(IvectorOpts)]; L.k3_ivector_opts_default.restype = None
    L.k3_ivector_create.argtypes = [ctypes.POINTER(IvectorModel), ctypes.POINTER(IvectorOpts), ctypes.POINTER(vp)]
    L.k3_ivector_destroy.argtypes = [vp]; L.k3_ivector_destroy.restype = None
    L.k3_ivector_get_info.argtypes = [vp, ctypes.POINTER(IvectorInfo)]
    L.k3_ivector_num_rows.argtypes = [vp, i32, vp, vp]; L.k3_ivector_num_rows.restype = i64
    L.k3_ivector_extract_batch.argtypes = [vp, vp, i64, vp, i32, vp, i64, vp]
    L.k3_ivector_stream_create.argtypes = [vp, ctypes.POINTER(vp)]; L.k3_ivector_stream_destroy.argtypes = [vp]; L.k3_ivector_stream_destroy.restype = None; L.k3_ivector_stream_reset.argtypes = [vp, vp]
    L.k3_ivector_stream_num_rows.argtypes = [vp]; L.k3_ivector_stream_num_rows.restype = i64; L.k3_ivector_stream_accept.argtypes = [vp, vp, i64, i32, i32, vp, i64, i32, ctypes.POINTER(i32), vp, vp]
    L.k3_ivector_stream_accept_batch.argtypes = [vp, i32, vp, i64, vp, vp, vp, i64, vp]
    L.k3_cmvn_online_batch_resume.argtypes = [vp, i64, vp, i64, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp]
    L.k3_ivector_extract_batch_adapt.argtypes = [vp, vp, i64, vp, i32, vp, i64, vp, vp, vp, vp]; L.k3_ivector_extract_batch_weighted.argtypes = [vp, vp, i64, vp, i32, vp, vp, i64, vp, vp, vp, vp]; L.k3_ivector_stats_size.argtypes = [vp]; L.k3_ivector_stats_size.restype = i64
    L.k3_nnet_load.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.k3_nnet_destroy.argtypes = [vp]; L.k3_nnet_destroy.restype = None
    L.k3_nnet_get_info.argtypes = [vp, ctypes.POINTER(NnetInfo)]
    L.k3_nnet_get_priors.argtypes = [vp, vp]
    L.k3_nnet_batch_create.argtypes = [vp, i32, vp, i32, vp, ctypes.c_float, ctypes.POINTER(vp)]
    L.k3_nnet_batch_destroy.argtypes = [vp]; L.k3_nnet_batch_destroy.restype = None
    L.k3_nnet_batch_output_rows.argtypes = [vp, vp]; L.k3_nnet_batch_output_rows.restype = i64
    L.k3_nnet_batch_flops.argtypes = [vp]; L.k3_nnet_batch_flops.restype = ctypes.c_double
    L.k3_nnet_forward.argtypes = [vp, vp, i64, vp, i64, vp]
    L.k3_nnet_batch_set_precision.argtypes = [vp, i32]
    L.k3_nnet_stream_create.argtypes = [vp, i32, i32, i32, vp, ctypes.c_float, ctypes.POINTER(vp)]
    L.k3_nnet_stream_destroy.argtypes = [vp]; L.k3_nnet_stream_destroy.restype = None
    L.k3_nnet_stream_get_info.argtypes = [vp, ctypes.POINTER(NnetStreamInfo)]
    L.k3_nnet_stream_reset.argtypes = [vp, vp, i32, vp, i64, vp]
    L.k3_nnet_stream_forward.argtypes = [vp, vp, i64, vp, vp, vp, i64, vp]
    L.k3_nnet_batch_create_ivector.argtypes = [vp, i32, vp, i32, vp, ctypes.c_float, i32, i32, vp, ctypes.POINTER(vp)]
    L.k3_nnet_batch_ivector_rows.argtypes = [vp]; L.k3_nnet_batch_ivector_rows.restype = i64
    L.k3_nnet_forward_ivector.argtypes = [vp, vp, i64, vp, i64, vp, i64, vp]
    L.k3_fst_create.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, ctypes.POINTER(vp)]
    L.k3_fst_create_empty.argtypes = [i32, i64, i32, ctypes.POINTER(vp)]
    L.k3_fst_destroy.argtypes = [vp]; L.k3_fst_destroy.restype = None
    L.k3_fst_num_arcs.argtypes = [vp]; L.k3_fst_num_arcs.restype = i64
    L.k3_fst_num_states.argtypes = [vp]; L.k3_fst_num_states.restype = i32
    L.k3_fst_start.argtypes = [vp]; L.k3_fst_start.restype = i32
    L.k3_fst_image.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    L.k3_comm_create.argtypes = [ctypes.c_char_p, i32, i32, i32, ctypes.POINTER(vp)]; L.k3_comm_destroy.argtypes = [vp]; L.k3_comm_destroy.restype = None
    L.k3_fst_bcast.argtypes = [ctypes.POINTER(vp), vp, i32, i32, vp]
    L.k3_decoder_config_default.argtypes = [ctypes.POINTER(DecoderConfig)]; L.k3_decoder_config_default.restype = None
    L.k3_decoder_create.argtypes = [vp, ctypes.POINTER(DecoderConfig), i32, i32, ctypes.POINTER(vp)]
    L.k3_decoder_destroy.argtypes = [vp]; L.k3_decoder_destroy.restype = None
    L.k3_decoder_decode_batch.argtypes = [vp, i32, vp, i64, vp, vp]
    L.k3_decoder_init_decoding.argtypes = [vp, i32, i32, vp]; L.k3_decoder_advance_decoding.argtypes = [vp, i32, vp, i64, vp, vp]
    L.k3_decoder_advance_decoding_lanes.argtypes = [vp, i32, vp, vp, i32, i64, vp]
    L.k3_decoder_advance_decoding_strided.argtypes = [vp, i32, vp, vp, i64, vp]
    L.k3_decoder_init_channels.argtypes = [vp, vp, i32, vp]; L.k3_decoder_finalize_channels.argtypes = [vp, vp, i32, vp]
    L.k3_decoder_finalize_decoding.argtypes = [vp, vp]; L.k3_decoder_num_frames_decoded.argtypes = [vp, i32]; L.k3_decoder_num_frames_decoded.restype = i32
    L.k3_decoder_lattice_info.argtypes = [vp, vp]; L.k3_decoder_order_sensitive_events.argtypes = [vp, vp]; L.k3_decoder_pool_growths.argtypes = [vp, vp]
    L.k3_decoder_get_raw_lattices.argtypes = [vp] + [vp] * 10
    L.k3_decoder_get_best_path.argtypes = [vp, vp, i32, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp]
    L.k3_fst_export_image.argtypes = [vp, vp]; L.k3_fst_import_image.argtypes = [vp, vp]
    L.k3_decoder_set_profiling.argtypes = [vp, i32]; L.k3_decoder_kernel_times.argtypes = [vp, vp]; L.k3_decoder_stream_wait_token_passing.argtypes = [vp, vp]
    L.k3_decoder_phase_cycles.argtypes = [vp, vp]
    L.k3_decoder_frame_stats.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    f32 = ctypes.c_float
    L.k3_mat_add_mat_mat.argtypes = [f32, vp, i64, i32, vp, i64, i32, f32, vp, i64, i32, i32, i32, vp]
    for n in ("set", "scale", "add", "apply_floor", "apply_ceiling"): getattr(L, "k3_mat_" + n).argtypes = [vp, i64, i32, i32, f32, vp]
    for n in ("copy_rows_from_vec", "mul_cols_vec", "mul_rows_vec"): getattr(L, "k3_mat_" + n).argtypes = [vp, i64, i32, i32, vp, vp]
    for n in ("add_vec_to_rows", "add_vec_to_cols"): getattr(L, "k3_mat_" + n).argtypes = [f32, vp, f32, vp, i64, i32, i32, vp]
    L.k3_mat_copy_from_mat.argtypes = [vp, i64, i32, i32, vp, i64, i32, vp]
    L.k3_mat_add_mat.argtypes = [f32, vp, i64, i32, vp, i64, i32, i32, vp]
    L.k3_mat_copy_rows.argtypes = [vp, i64, i32, i32, vp, i64, vp, vp]
    L.k3_mat_add_rows.argtypes = [f32, vp, i64, vp, vp, i64, i32, i32, vp]
    L.k3_mat_mul_elements.argtypes = [vp, i64, i32, i32, vp, i64, vp]; L.k3_mat_heaviside.argtypes = [vp, i64, i32, i32, vp, i64, vp]
    L.k3_mat_add_mat_diag_vec.argtypes = [f32, vp, i64, i32, vp, f32, vp, i64, i32, i32, vp]; L.k3_mat_add_row_ranges.argtypes = [vp, i64, i32, i32, vp, i64, i32, vp, vp]
    L.k3_mat_copy_lower_to_upper.argtypes = [vp, i64, i32, vp]; L.k3_mat_add_to_diag.argtypes = [vp, i64, i32, i32, f32, vp]; L.k3_mat_add_vec_vec.argtypes = [f32, vp, vp, vp, i64, i32, i32, vp]
    L.k3_mat_normalize_rows.argtypes = [i32, vp, i64, vp, i64, vp, i64, i32, i32, f32, i32, vp]; L.k3_mat_apply_map.argtypes = [i32, vp, i64, i32, i32, vp, i64, f32, i32, vp]
    L.k3_mat_diff_activation.argtypes = [i32, vp, i64, i32, i32, vp, i64, vp, i64, vp]; L.k3_mat_div_rows_vec.argtypes = [vp, i64, i32, i32, vp, vp]; L.k3_mat_copy_cols_from_vec.argtypes = [vp, i64, i32, i32, vp, vp]
    L.k3_mat_set_rand.argtypes = [i32, vp, i64, i32, i32, ctypes.c_uint64, ctypes.c_uint64, vp]
    L.k3_mat_mul_rows.argtypes = [vp, i64, i32, i32, vp, i64, vp, vp]
    L.k3_mat_elements3.argtypes = [i32, vp, i64, i32, i32, f32, vp, i64, vp, i64, vp, i64, f32, vp]
    L.k3_mat_copy_cols.argtypes = [i32, vp, i64, i32, i32, vp, i64, vp, vp]
    L.k3_mat_add_diag_vec_mat.argtypes = [f32, vp, vp, i64, i32, f32, vp, i64, i32, i32, vp]; L.k3_mat_div_elements.argtypes = [vp, i64, i32, i32, vp, i64, vp]; L.k3_mat_reduce_scalar.argtypes = [i32, vp, i64, vp, i64, i32, i32, vp, vp]
    L.k3_vec_col_reduce.argtypes = [i32, f32, vp, i64, vp, i64, i32, i32, f32, vp, vp]
    L.k3_chain_den_create.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]; L.k3_chain_den_destroy.argtypes = [vp]; L.k3_chain_den_destroy.restype = None
    L.k3_chain_den_num_states.argtypes = [vp]; L.k3_chain_den_initial_probs.argtypes = [vp, vp]
    L.k3_chain_den_forward_backward.argtypes = [vp, vp, i64, i32, i32, f32, f32, vp, i64, ctypes.POINTER(f32), ctypes.POINTER(i32), vp]
    L.k3_chain_supervision_create.argtypes = [i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]; L.k3_chain_supervision_create_e2e.argtypes = [i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]; L.k3_chain_supervision_destroy.argtypes = [vp]; L.k3_chain_supervision_destroy.restype = None
    L.k3_chain_numerator.argtypes = [vp, vp, i64, vp, i64, ctypes.POINTER(f32), vp]
    L.k3_chain_objf_and_deriv.argtypes = [vp, vp, ctypes.POINTER(ChainTrainingOpts), vp, i64, vp, i64, vp, i64, ctypes.POINTER(f32), ctypes.POINTER(f32), ctypes.POINTER(f32), vp]
    _lib = L
    return L

class ChainTrainingOpts(ctypes.Structure):
    """k3_chain_training_opts (include/k3hip.h) = chain::ChainTrainingOptions"""
    _fields_ = [("l2_regularize", ctypes.c_float), ("out_of_range_regularize", ctypes.c_float), ("leaky_hmm_coefficient", ctypes.c_float), ("apply_out_of_range_penalty", ctypes.c_int32)]

def source_digest():
    """The digest k3_build_id() carries, recomputed from the sources next to this file (kaldi_amd/csrc/Makefile: every *.hip / *.h of csrc in byte order of their
    names, then include/k3hip.h; first 16 hex digits of the SHA-256 of the concatenation)."""
    import glob, hashlib
    d = os.path.join(HERE, "csrc")
    files = sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")), key=lambda f: os.path.basename(f).encode())
    files.append(os.path.join(os.path.dirname(HERE), "include", "k3hip.h"))
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh: h.update(fh.read())
    return h.hexdigest()[:16]

def build_id():
    return load().k3_build_id().decode()

def check_provenance():
    """(build id of the mapped library, digest of the shipped sources); raises K3Error when the library was not built from these sources."""
    bid, dig = build_id(), source_digest()
    if not bid.endswith("+" + dig):
        raise K3Error(f"{LIB_PATH} was built from other sources: its build id is {bid}, the sources here digest to {dig} (run make -C kaldi_amd/csrc)")
    return bid, dig

def check(status):
    if status != 0:
        raise K3Error(f"k3 status {status}: {load().k3_last_error().decode()}")
