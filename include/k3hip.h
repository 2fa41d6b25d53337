/*
 * k3hip.h -- C ABI of libk3hip.so: the MI355X (gfx950) native kernels of the batched acoustic
 * pipeline (fbank/MFCC/CMVN -> nnet3 TDNN-F forward -> HCLG lattice decode).
 *
 * The reference (kaldi-asr/kaldi) has no C ABI for this path; its boundary is C++ classes
 * (SURVEY.md 8b).  Each entry point below names the reference interface it stands behind
 * (paths relative to the reference's src/).  All pointers named d_* are DEVICE pointers (HBM),
 * h_* are host pointers; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Every function returns 0 on success, a negative k3_status on error; k3_last_error() returns
 * the thread-local message (the C++ adapters turn it into KALDI_ERR / KaldiFatalError).
 * No torch types, no C++ types, no hidden host<->device copies inside *_compute / *_forward /
 * *_advance calls.
 */
#ifndef K3HIP_H_
#define K3HIP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { K3_OK = 0, K3_ERR_ARG = -1, K3_ERR_HIP = -2, K3_ERR_UNSUPPORTED = -3, K3_ERR_OVERFLOW = -4 } k3_status;
const char *k3_last_error(void);
int k3_version(void);
/* "k3hip-<abi revision>+<digest>": digest = first 16 hex digits of the SHA-256 over the sources the library was built from (kaldi_amd/csrc/ *.hip and *.h in byte order
 * of their names, then this header) -- a caller that has the sources can tell whether the shared object it mapped was built from them (kaldi_amd/lib.py source_digest()). */
const char *k3_build_id(void);

/* ---------------------------------------------------------------- features ------------------
 * Replaces: feat::OfflineFeatureTpl<FbankComputer|MfccComputer>::Compute (feat/feature-common-inl.h:59-83)
 * and the GPU reference CudaSpectralFeatures::ComputeFeatures (cudafeat/feature-spectral-cuda.cu:525-567),
 * OnlineCudaFeaturePipeline::ComputeFeatures (cudafeat/online-cuda-feature-pipeline.cc:66-87).
 * Field order/meaning = FrameExtractionOptions (feat/feature-window.h:35-67), MelBanksOptions
 * (feat/mel-computations.h:43-60), FbankOptions (feat/feature-fbank.h:44-61), MfccOptions
 * (feat/feature-mfcc.h:40-60).  dither must be 0 in parity runs. */
typedef struct k3_feat_opts {
  float samp_freq, frame_shift_ms, frame_length_ms, dither, preemph_coeff, blackman_coeff;
  int32_t remove_dc_offset, round_to_power_of_two, snip_edges;
  int32_t window_type; /* 0 hanning 1 sine 2 hamming 3 povey 4 rectangular 5 blackman */
  int32_t num_bins;
  float low_freq, high_freq, vtln_low, vtln_high;
  int32_t htk_mode;
  int32_t use_energy;
  float energy_floor;
  int32_t raw_energy, htk_compat, use_log_fbank, use_power;
  int32_t num_ceps;
  float cepstral_lifter;
  int32_t feature_type; /* 0 fbank, 1 mfcc */
  float vtln_warp;      /* the vtln_warp argument of Compute(); 1.0 = none */
} k3_feat_opts;

typedef struct k3_feat_plan k3_feat_plan;
/* Builds window / mel-bank / DCT / lifter / twiddle tables on the host (same formulas as
 * FeatureWindowFunction, MelBanks::MelBanks, ComputeDctMatrix, ComputeLifterCoeffs) and uploads them. */
int k3_feat_plan_create(const k3_feat_opts *opts, k3_feat_plan **plan);
void k3_feat_plan_destroy(k3_feat_plan *plan);
int32_t k3_feat_dim(const k3_feat_plan *plan);                       /* FbankComputer::Dim / MfccComputer::Dim */
int32_t k3_feat_num_frames(const k3_feat_plan *plan, int64_t nsamp); /* NumFrames(), feat/feature-window.cc:40-87, flush=true */
/* Batched whole-utterance extraction.  Utterance u occupies d_waves[d_wave_offsets[u] .. d_wave_offsets[u+1])
 * (float32 samples, as CuVector<BaseFloat> cu_wave in the reference) and produces rows
 * d_frame_offsets[u] .. d_frame_offsets[u+1] of d_feats (row-major, leading dimension ld floats).
 * total_frames == d_frame_offsets[U] is passed by value so no device->host read is needed. */
int k3_feat_compute_batch(k3_feat_plan *plan, const float *d_waves, const int64_t *d_wave_offsets,
                          const int64_t *d_frame_offsets, int32_t num_utts, int64_t total_frames,
                          float *d_feats, int64_t ld, void *stream);
/* The same with the samples as 16-bit PCM, the way WaveData reads them from a RIFF file (feat/wave-reader.cc:187-244) before it converts
 * them to float: the values are identical, the kernel reads 2 instead of 4 bytes per sample (480 B per frame) and the host
 * ships half the bytes over PCIe. */
int k3_feat_compute_batch_pcm16(k3_feat_plan *plan, const int16_t *d_waves, const int64_t *d_wave_offsets,
                                const int64_t *d_frame_offsets, int32_t num_utts, int64_t total_frames,
                                float *d_feats, int64_t ld, void *stream);
/* ResampleWaveform (feat/resample.cc:363-372: LinearResample with cutoff 0.99 * 0.5 * min(rate), six zero crossings, flushed): what OfflineFeatureTpl::ComputeFeatures
 * (feat/feature-common-inl.h:29-57) does to a waveform whose rate differs from --sample-frequency under --allow-downsample / --allow-upsample.  Utterance u's samples are
 * d_in[h_in_offsets[u] .. h_in_offsets[u+1]) and come out as d_out[h_out_offsets[u] ..), k3_resample_num_samples(rate_in, rate_out, n) of them.  Synchronous. */
int64_t k3_resample_num_samples(int32_t rate_in, int32_t rate_out, int64_t num_in);
int k3_resample_batch(int32_t rate_in, int32_t rate_out, const float *d_in, const int64_t *h_in_offsets, int32_t num_utts, float *d_out,
    const int64_t *h_out_offsets, void *stream);
/* Per-utterance CMVN in place: AccCmvnStats + ApplyCmvn (transform/cmvn.cc:30-115), what
 * `compute-cmvn-stats | apply-cmvn [--norm-vars]` do with one utterance per speaker.
 * fp64 accumulators like the reference.  d_stats (optional, may be NULL): [U x 2 x (dim+1)] doubles. */
int k3_cmvn_offline_batch(float *d_feats, int64_t ld, int32_t dim, const int64_t *d_frame_offsets,
                          int32_t num_utts, int32_t norm_vars, double *d_stats, void *stream);

/* Online (sliding-window) CMVN of whole utterances: OnlineCmvn::GetFrame for every frame (feat/online-feature.cc:361-468), what
 * online2bin/apply-cmvn-online.cc:92-129 and the GPU reference CudaOnlineCmvn::ComputeFeatures
 * (cudafeat/feature-online-cmvn-cuda.cu:174-215) do.  Options = OnlineCmvnOptions (feat/online-feature.h:201-240); fp64 window
 * statistics with the reference's add-new / subtract-oldest recursion.  d_global_stats [2 x (dim+1)] doubles (sums + count,
 * sums of squares) as read from a <global-cmvn-stats> file; d_speaker_stats (may be NULL) [U x 2 x (dim+1)]: per utterance, the
 * stats of the speaker's earlier utterances (OnlineCmvnState::speaker_cmvn_stats; count 0 = none).  skip_dims: host array
 * (--skip-dims).  d_out must not alias d_in.  Synchronises the stream (error flag); K3_ERR_ARG where the reference raises. */
typedef struct k3_online_cmvn_opts {
  int32_t cmn_window, speaker_frames, global_frames;   /* 600, 600, 200 */
  int32_t normalize_mean, normalize_variance;           /* 1, 0 */
} k3_online_cmvn_opts;
void k3_online_cmvn_opts_default(k3_online_cmvn_opts *opts);
int k3_cmvn_online_batch(const float *d_in, int64_t ld_in, float *d_out, int64_t ld_out, int32_t dim, const int64_t *d_frame_offsets,
                         int32_t num_utts, const k3_online_cmvn_opts *opts, const double *d_global_stats, const double *d_speaker_stats,
                         const int32_t *skip_dims, int32_t num_skip_dims, void *stream);
/* The same recursion continued over a stream (OnlineCmvn keeps its window statistics between GetFrame calls, feat/online-feature.cc:361-468): utterance u's rows
 * [0, t_begin[u]) are the history the window still reads (at least min(cmn_window, frames so far) of them) -- not written --, d_carry [U x dim x 3] float64 holds each
 * column's window (sum, sum of squares, count) after the last row of the previous call (read when t_begin[u] > 0) and receives it after the last row of this one: the
 * rows written are bit-identical to those of the whole utterance in one call. */
int k3_cmvn_online_batch_resume(const float *d_in, int64_t ld_in, float *d_out, int64_t ld_out, int32_t dim, const int64_t *d_frame_offsets,
                                int32_t num_utts, const k3_online_cmvn_opts *opts, const double *d_global_stats, const double *d_speaker_stats,
                                const int32_t *skip_dims, int32_t num_skip_dims, const int64_t *d_t_begin, double *d_carry, void *stream);

/* ---------------------------------------------------------------- online i-vectors -----------
 * Replaces, for whole utterances: OnlineIvectorFeature (online2/online-ivector-feature.h:278-420) as ivector-extract-online2 drives it
 * (online2bin/ivector-extract-online2.cc:120-175: every frame weighted 1, fresh adaptation state, use_most_recent_ivector=false) and the GPU
 * reference BatchedIvectorExtractorCuda::GetIvectors (cudafeat/feature-online-batched-ivector-cuda.h:30-61).  One i-vector per
 * ivector_period frames: row k of an utterance is the estimate from the statistics of frames 0..k*period (UpdateStatsUntilFrame :248-277),
 * with the prior offset subtracted from its first element like GetFrame (:327-355).  The arithmetic follows the CPU reference (fp64 statistics,
 * 15 warm-started conjugate-gradient iterations); exact_solve = 1 replaces the conjugate gradient by a Cholesky solve, the method of the GPU
 * reference (and LinearCgd's own fall-back).
 * The model is handed over as host arrays, the way the reference's readers hold them (libk3host's k3h_ivector_config_* reads the files of an
 * --ivector-extraction-config): lda [lda_rows x lda_cols] with lda_cols = feat_dim*(left+right+1) or one more (an offset column,
 * OnlineTransform); global_cmvn_stats [2 x (feat_dim+1)]; the diagonal UBM (DiagGmm gconsts / means_invvars / inv_vars, [G] / [G x D] / [G x D],
 * D = lda_rows); the extractor's M [G x D x R] and Sigma^-1 [G x D(D+1)/2] packed lower triangles (IvectorExtractor M_, Sigma_inv_), prior_offset. */
typedef struct k3_ivector k3_ivector;
typedef struct k3_ivector_model {
  int32_t feat_dim, lda_rows, lda_cols, num_gauss, ivector_dim;
  const float *lda;
  const double *global_cmvn_stats;
  const double *gconsts, *means_invvars, *inv_vars;
  const double *M, *sigma_inv;
  double prior_offset;
} k3_ivector_model;
typedef struct k3_ivector_opts {       /* OnlineIvectorExtractionConfig (online2/online-ivector-feature.h:60-150) + its splice / cmvn configs */
  int32_t left_context, right_context; /* --splice-config */
  int32_t num_gselect;                 /* 5 */
  float min_post, posterior_scale;     /* 0.025, 0.1 */
  float max_count;                     /* 0 = off */
  int32_t ivector_period;              /* 10 */
  int32_t num_cg_iters;                /* 15 */
  int32_t exact_solve;                 /* 0 */
  int32_t online_cmvn_iextractor;      /* 0: the statistics see the features without CMVN (--online-cmvn-iextractor) */
  k3_online_cmvn_opts cmvn;            /* --cmvn-config */
} k3_ivector_opts;
typedef struct k3_ivector_info { int32_t feat_dim, lda_dim, num_gauss, ivector_dim, ivector_period; } k3_ivector_info;
void k3_ivector_opts_default(k3_ivector_opts *opts);
int k3_ivector_create(const k3_ivector_model *model, const k3_ivector_opts *opts, k3_ivector **iv);
void k3_ivector_destroy(k3_ivector *iv);
int k3_ivector_get_info(const k3_ivector *iv, k3_ivector_info *info);
/* rows of the output: sum over utterances of ceil(frames / period); h_row_offsets (nullable) gets the U+1 row offsets */
int64_t k3_ivector_num_rows(const k3_ivector *iv, int32_t num_utts, const int64_t *h_frame_offsets, int64_t *h_row_offsets);
/* d_feats: the utterances' base features back to back (h_frame_offsets[u]..[u+1], host array of U+1, first 0), as k3_feat_compute_batch writes
 * them; d_ivectors [num_rows x ld_ivectors].  Asynchronous on `stream` after a short synchronisation for the offsets. */
int k3_ivector_extract_batch(k3_ivector *iv, const float *d_feats, int64_t ld_feats, const int64_t *h_frame_offsets, int32_t num_utts,
                             float *d_ivectors, int64_t ld_ivectors, void *stream);
/* The same with the speaker's adaptation state (OnlineIvectorExtractorAdaptationState, online-ivector-feature.h:185-230: what
 * ivector-extract-online2 carries from one utterance of a speaker to the next, :100-178): per utterance, the CMVN statistics of the speaker's earlier
 * utterances (d_cmvn_speaker_stats [U x 2 x (feat_dim+1)], count 0 = none; nullable) and the i-vector statistics so far (d_stats_in, nullable = fresh);
 * d_stats_out (nullable) receives the statistics after the utterance's last estimate (GetAdaptationState).  One record is k3_ivector_stats_size()
 * doubles: num_frames, the linear term [R], the quadratic term [R x R].  Utterances of one speaker go in consecutive calls. */
int64_t k3_ivector_stats_size(const k3_ivector *iv);
/* on: d_stats_out holds the statistics of EVERY frame of the utterance (the frames behind the last estimate included) -- what the reference's ivector-extract-online2 --repeat=true
 * carries to the speaker's next utterance (it asks for the last frame's i-vector: ivector-extract-online2.cc:121-127); off (default): the statistics at the last estimate */
void k3_ivector_set_accumulate_tail(k3_ivector *iv, int32_t on);
/* Streaming: one object per audio stream, fed with the feature frames as they come (OnlineIvectorFeature, online2/online-ivector-feature.h:233-330; per channel what
 * BatchedIvectorExtractorCuda keeps, cudafeat/feature-online-batched-ivector-cuda.h:30-61).  accept(): `num_frames` new feature rows (device), `finished` != 0 with the
 * stream's last frames (then the frames waiting for their splice context are processed with the last frame repeated).  Every estimate the frames ready allow is made
 * -- frames ready = all frames so far minus the splice's right context while the stream goes on; estimate k when k * ivector_period < frames ready -- and the rows are,
 * bit for bit, rows k of k3_ivector_extract_batch on the whole utterance; each frame is processed once (cost linear in the stream's length, memory bounded).
 * d_new_rows (nullable) [max_new_rows x ld_rows] receives the rows made by this call, *h_num_new_rows their number; d_latest (nullable) [ivector_dim] the most recent
 * estimate of the stream -- what the reference's online decodable hands the network (nnet3/decodable-online-looped.cc:182-197) --, zeros before the first. */
typedef struct k3_ivector_stream k3_ivector_stream;
int k3_ivector_stream_create(k3_ivector *iv, k3_ivector_stream **out);
void k3_ivector_stream_destroy(k3_ivector_stream *s);
int k3_ivector_stream_reset(k3_ivector_stream *s, void *stream);
int64_t k3_ivector_stream_num_rows(const k3_ivector_stream *s);
int k3_ivector_stream_accept(k3_ivector_stream *s, const float *d_feats, int64_t ld_feats, int32_t num_frames, int32_t finished, float *d_new_rows, int64_t ld_rows,
                             int32_t max_new_rows, int32_t *h_num_new_rows, float *d_latest, void *stream);
/* The same for the streams of one batch, one kernel launch per stage (what BatchedIvectorExtractorCuda::GetIvectors is per chunk, cudafeat/feature-online-batched-ivector-cuda.h:30-61):
 * stream i takes feature rows h_frame_offsets[i] .. h_frame_offsets[i + 1] of d_feats (offsets start at 0; a stream may get no rows),
     h_finished[i] != 0 ends it;
 d_latest (nullable)
 * [num_streams x ld_latest] receives every stream's most recent estimate.  Results per stream are those of k3_ivector_stream_accept.  All streams belong to one extractor, each is
 * listed once; one batched call at a time per extractor. */
int k3_ivector_stream_accept_batch(k3_ivector_stream **streams, int32_t num_streams, const float *d_feats, int64_t ld_feats, const int64_t *h_frame_offsets,
    const int32_t *h_finished,
                                   float *d_latest, int64_t ld_latest, void *stream);
int k3_ivector_extract_batch_adapt(k3_ivector *iv, const float *d_feats, int64_t ld_feats, const int64_t *h_frame_offsets, int32_t num_utts,
                                   float *d_ivectors, int64_t ld_ivectors, const double *d_cmvn_speaker_stats, const double *d_stats_in,
                                   double *d_stats_out, void *stream);
/* The same with per-frame weights on the statistics (silence weighting: OnlineIvectorFeature::UpdateFrameWeights with the whole utterance's weights known at the start, what
 * ivector-extract-online2 --frame-weights-rspecifier does,
     online2bin/ivector-extract-online2.cc:130-153): d_frame_weights [total frames] (NULL: every frame weighs 1) -- a frame of
 * weight 0 contributes nothing, otherwise its posteriors are pruned at GetMinPost(weight) = min(0.99, min_post / |weight|) and scaled by posterior_scale * weight
 * (online2/online-ivector-feature.cc:188-199, :226-236). */
int k3_ivector_extract_batch_weighted(k3_ivector *iv, const float *d_feats, int64_t ld_feats, const int64_t *h_frame_offsets, int32_t num_utts, const float *d_frame_weights,
                                      float *d_ivectors, int64_t ld_ivectors, const double *d_cmvn_speaker_stats, const double *d_stats_in, double *d_stats_out, void *stream);

/* ---------------------------------------------------------------- nnet3 forward -------------
 * Replaces, for "simple" feed-forward TDNN / TDNN-F models: nnet3::NnetComputer::Run over the compiled
 * program of DecodableNnetSimple (nnet3/nnet-am-decodable-simple.cc:93-276; nnet3/nnet-compute.cc:236-459)
 * and the GPU reference BatchedStaticNnet3::RunBatch (cudadecoder/batched-static-nnet3.cc:293-365).
 * Results equal nnet3-compute's: rows t = 0, s, 2s, .. (ceil(T/s) per utterance), edge frames replicated. */
typedef struct k3_nnet k3_nnet;
typedef struct k3_nnet_batch k3_nnet_batch;
typedef struct k3_nnet_info {
  int32_t input_dim, output_dim, left_context, right_context;
  int32_t num_components, num_fused_nodes, has_priors;
  int64_t num_params;
  int32_t ivector_dim;                 /* dim of input-node name=ivector, 0 = none */
} k3_nnet_info;
/* Nnet::Read (nnet3/nnet-nnet.cc:586-628) / AmNnetSimple::Read (nnet3/am-nnet-simple.cc:47-57): text or binary,
 * raw nnet or final.mdl (TransitionModel + AmNnetSimple).  BatchNorm/dropout are put in test mode
 * (SetBatchnormTestMode/SetDropoutTestMode, nnet3/nnet-utils.h:188,258) and layers are fused (cf. CollapseModel).
 * K3_ERR_UNSUPPORTED for component/descriptor types outside the TDNN/TDNN-F family. */
int k3_nnet_load(const char *model_path, k3_nnet **nnet);
void k3_nnet_destroy(k3_nnet *nnet);
int k3_nnet_get_info(const k3_nnet *nnet, k3_nnet_info *info);
int k3_nnet_get_priors(const k3_nnet *nnet, float *h_priors /* [output_dim] */);
/* Plans one ragged batch: h_num_frames[u] input frames per utterance, stored back to back (utterance u starts at
 * row sum_{v<u} h_num_frames[v] of the feature matrix -- the layout k3_feat_compute_batch writes).
 * h_log_priors (nullable) and acoustic_scale fold DecodableNnetSimple's "-log prior, * acwt" (:268-271) into the
 * last layer's epilogue.  Owns the activation workspace in HBM. */
int k3_nnet_batch_create(k3_nnet *nnet, int32_t num_utts, const int32_t *h_num_frames, int32_t frame_subsampling_factor,
                         const float *h_log_priors, float acoustic_scale, k3_nnet_batch **batch);
void k3_nnet_batch_destroy(k3_nnet_batch *batch);
/* total output rows; h_out_offsets (nullable) receives the U+1 row offsets of the utterances in d_out */
int64_t k3_nnet_batch_output_rows(const k3_nnet_batch *batch, int64_t *h_out_offsets);
double k3_nnet_batch_flops(const k3_nnet_batch *batch);   /* exact sum of 2*M*N*K over the launched GEMMs */
int k3_nnet_forward(k3_nnet_batch *batch, const float *d_feats, int64_t ld_feats, float *d_out, int64_t ld_out, void *stream);
/* EXPLORATORY (round 5), never the default and never behind bench.py's `value`: mode 1 runs this batch's affine products on the bf16 matrix core with both
 * operands split three ways (hi + mid + lo bf16 parts, exact; six of the nine cross products: as accurate as an fp32 GEMM, 0.375 of the FP32 matrix-core time) wherever a
 * node's time offsets are k-tile aligned; mode 0 = the FP32 matrix core with the reference's summation order (the parity path).  Outputs of mode 1 are NOT held to the 1e-4
 * fixture gates -- they round differently from the reference -- but to the float64 forward: no further from it than nnet3-compute is (bench.py split_bf16).
 * mode 2 = the same six products with the ACTIVATIONS' planes written by the producing kernel's epilogue (three bf16 planes next to the fp32 copy, 6 more bytes per element of HBM),
 * so that the consumer's loader only loads; bit-identical to mode 1's output. */
int k3_nnet_batch_set_precision(k3_nnet_batch *batch, int32_t mode);
/* Stateful streaming forward (round 5): what BatchedStaticNnet3::RunBatch (cudadecoder/batched-static-nnet3.cc:139-233) is for the online pipeline -- one network pass per
 * chunk of every active channel -- WITHOUT re-evaluating the chunk's left and right context: every node keeps, per channel, the last few rows it produced, a pass consumes
 * frames_per_chunk new input frames per channel and produces frames_per_chunk / subsampling output rows, every row of every node is computed once per stream.  Outputs are
 * bit-identical to k3_nnet_forward over the whole utterance (edge frames: a restarted channel's histories are its response to frame 0 replicated, the end of a stream
 * replicates its last frame; nnet-am-decodable-simple.cc:154-163).  K3_ERR_UNSUPPORTED for models it cannot run (i-vector input, row operations inside the network,
 * frames_per_chunk shorter than a node's history): callers fall back to planning chunk + context with k3_nnet_batch_create.
 * Channel c's output row k of a pass lies at d_out row k * num_channels + c and belongs to time first_output_time + P * frames_per_chunk + k * subsampling, P = the passes
 * channel c took part in since its last reset (rows at negative times, or beyond the stream's last frame, are to be ignored). */
typedef struct k3_nnet_stream k3_nnet_stream;
typedef struct k3_nnet_stream_info {
  int32_t num_channels, frames_per_chunk, subsampling, output_rows_per_pass, first_output_time, right_context, input_history;
  double flops_per_pass;
} k3_nnet_stream_info;
int k3_nnet_stream_create(k3_nnet *nnet, int32_t num_channels, int32_t frames_per_chunk, int32_t frame_subsampling_factor, const float *h_log_priors,
    float acoustic_scale, k3_nnet_stream **out);
void k3_nnet_stream_destroy(k3_nnet_stream *s);
int k3_nnet_stream_get_info(const k3_nnet_stream *s, k3_nnet_stream_info *info);
/* the listed channels start new streams; d_first_frames row i = the first feature frame of channel h_channels[i]'s stream */
int k3_nnet_stream_reset(k3_nnet_stream *s, const int32_t *h_channels, int32_t n, const float *d_first_frames, int64_t ld, void *stream);
/* one pass: channel c with h_row_count[c] >= 0 consumes rows h_row_start[c] .. + h_row_count[c] of d_new (frames_per_chunk of them; fewer or none only once its audio has
 * ended: the missing frames replicate the last one); h_row_count[c] < 0: the channel sits the pass out */
int k3_nnet_stream_forward(k3_nnet_stream *s, const float *d_new, int64_t ld_new, const int64_t *h_row_start, const int32_t *h_row_count, float *d_out,
    int64_t ld_out, void *stream);
/* Models with the recipes' i-vector input ("input-node name=ivector", tdnn1 fed by Append(.., ReplaceIndex(ivector, t, 0)); k3_nnet_info.ivector_dim > 0).
 * online_ivector_period > 0: nnet3-compute / nnet3-latgen-faster --online-ivectors=.. --online-ivector-period=P --frames-per-chunk=C: the network is
 * evaluated chunk by chunk (chunks of C frames rounded up to a multiple of the subsampling factor), chunk c with the row GetCurrentIvector picks for it
 * (frame in the middle of the chunk / P, the last row when the matrix is a little short; nnet-am-decodable-simple.cc:93-213), h_num_ivector_rows[u] rows
 * per utterance.  online_ivector_period = 0: --ivectors, one i-vector per utterance (row u), whole utterances at once.
 * d_ivectors: the utterances' rows back to back ([sum rows x ld_ivectors], k3_nnet_batch_ivector_rows in total) -- the layout k3_ivector_extract_batch writes. */
int k3_nnet_batch_create_ivector(k3_nnet *nnet, int32_t num_utts, const int32_t *h_num_frames, int32_t frame_subsampling_factor,
                                 const float *h_log_priors, float acoustic_scale, int32_t frames_per_chunk, int32_t online_ivector_period,
                                 const int32_t *h_num_ivector_rows, k3_nnet_batch **batch);
int64_t k3_nnet_batch_ivector_rows(const k3_nnet_batch *batch);
int k3_nnet_forward_ivector(k3_nnet_batch *batch, const float *d_feats, int64_t ld_feats, const float *d_ivectors, int64_t ld_ivectors,
                            float *d_out, int64_t ld_out, void *stream);

/* ---------------------------------------------------------------- decoding graph -------------
 * Replaces: cuda_decoder::CudaFst(const fst::StdFst &fst, const TransitionInformation *trans_model)
 * (cudadecoder/cuda-fst.h:62-149, cuda-fst.cc:38-197): the HCLG as a CSR resident in HBM, emitting arcs of a state
 * stored before its non-emitting (ilabel 0) arcs, transition-id -> pdf-id applied on the ilabels
 * (cuda-fst.cc:166-175).  Input is a generic host CSR in FST arc order (what an OpenFst reader or a graph
 * builder hands over): arcs of state s are [h_arc_offsets[s], h_arc_offsets[s+1]); h_final[s] is the final cost
 * (+inf = not final); h_tid2pdf[t] maps ilabel t in [1, num_tids) to a column of the log-likelihood matrix
 * (NULL = identity minus one is NOT assumed: pass the map).  The device copy is 16 B per arc + 8 B per state. */
typedef struct k3_fst k3_fst;
int k3_fst_create(int32_t num_states, int32_t start, const int32_t *h_arc_offsets, const int32_t *h_ilabel,
                  const int32_t *h_olabel, const float *h_weight, const int32_t *h_nextstate, const float *h_final,
                  const int32_t *h_tid2pdf, int32_t num_tids, k3_fst **fst);
void k3_fst_destroy(k3_fst *fst);
int64_t k3_fst_num_arcs(const k3_fst *fst);
int32_t k3_fst_num_states(const k3_fst *fst);
int32_t k3_fst_start(const k3_fst *fst);
/* Size in bytes and device address of the packed read-only graph image (one contiguous allocation), so that a
 * multi-GPU launcher can broadcast it once over RCCL (ncclBroadcast of `bytes` uint8 from the loading rank) and
 * attach it on the other ranks with k3_fst_attach() -- SURVEY 8e: "HCLG broadcast once over RCCL/xGMI". */
int k3_fst_image(const k3_fst *fst, void **d_image, int64_t *bytes);
int k3_fst_create_empty(int32_t num_states, int64_t num_arcs, int32_t start, k3_fst **fst); /* allocate an image of that shape (receiver side) */
int k3_fst_export_image(const k3_fst *fst, void *d_dst);   /* device-to-device copy of the image into a caller buffer (e.g. the collective's send buffer) */
int k3_fst_import_image(k3_fst *fst, const void *d_src);   /* the reverse, after the collective */

/* Multi-GPU: one process per GPU; the graph is read and converted on ONE rank and broadcast over RCCL (xGMI inside a node) into the
 * other ranks' HBM, once; nothing else is shared (utterances are independent, every rank writes its own lattices: lat.JOB of decode.sh:123).
 * k3_fst_bcast: on `root` *fst is the graph; on the other ranks *fst is NULL on entry and owns a graph with root's image on return.  `comm` is an
 * ncclComm_t -- the application's own, or k3_comm_create's: the 128-byte ncclUniqueId travels through a file every rank can see (rank 0 writes
 * id_file, the others wait up to timeout_seconds for it; k3_comm_rendezvous below).  RCCL is bound at run time (dlopen): single-GPU users never load it. */
int k3_comm_create(const char *id_file, int32_t rank, int32_t world_size, int32_t timeout_seconds, void **comm);
void k3_comm_destroy(void *comm);
/* the rendezvous' file protocol by itself (no RCCL): rank 0 publishes the 128 bytes at id_in, the others receive them in id_out; files of other runs are
 * refused (run identity: K3_COMM_NONCE / TORCHELASTIC_RUN_ID when set, otherwise nothing older than stale_seconds before the caller's start) */
int k3_comm_exchange_id(const char *id_file, int32_t rank, int32_t timeout_seconds, int32_t stale_seconds, const void *id_in, void *id_out);
/* what k3_comm_create does in front of ncclCommInitRank (which has no deadline of its own): the id exchange above plus an arrival handshake -- rank r announces itself in
 * <id_file>.arrived.<r>, rank 0 writes <id_file>.go once it has seen all world_size - 1 announcements of the id it published -- so that no rank enters the collective
 * unless every rank is there; a rank that never arrives makes every other rank return an error that names it within timeout_seconds instead of blocking for ever
 * (tests/test_parallel_cpu.py: fault injection with real processes, no RCCL needed) */
int k3_comm_rendezvous(const char *id_file, int32_t rank, int32_t world_size, int32_t timeout_seconds, int32_t stale_seconds, const void *id_in, void *id_out);
/* in-place sum over the ranks of `comm` (ncclAllReduce): the gradient exchange of data-parallel chain training; asynchronous on `stream` */
int k3_comm_allreduce_f32(void *comm, float *d_buf, int64_t count, void *stream);
int k3_fst_bcast(k3_fst **fst, void *comm /* ncclComm_t */, int32_t root, int32_t rank, void *stream);

/* ---------------------------------------------------------------- lattice decoder ------------
 * Replaces: LatticeFasterDecoder::Decode = InitDecoding + AdvanceDecoding + FinalizeDecoding + GetRawLattice
 * (decoder/lattice-faster-decoder.cc:63-197,588-649; the CPU parity oracle) behind the batched lane model of
 * cuda_decoder::CudaDecoder (cudadecoder/cuda-decoder.h:224-345: ctor(fst, config, nlanes, nchannels),
 * InitDecoding, AdvanceDecoding, GetRawLattice).  One workgroup per lane runs the whole frame recurrence.
 * Semantics = the reference CPU decoder's token passing with the beam applied against the FINAL per-frame
 * cutoff (order-independent; see DESIGN.md "decoder parity"), float32 costs formed in the reference's
 * evaluation order; lattice-beam pruning (PruneForwardLinks / PruneForwardLinksFinal / PruneTokensForFrame,
 * :308-507) runs on the GPU after the last frame.  Field names/defaults = LatticeFasterDecoderConfig
 * (decoder/lattice-faster-decoder.h:37-107) / CudaDecoderConfig (cuda-decoder.h:58-163). */
typedef struct k3_decoder_config {
  float beam;              /* 16.0 (recipes: 15.0) */
  int32_t max_active;      /* INT32_MAX  (CudaDecoderConfig: 10000) */
  int32_t min_active;      /* 200 */
  float lattice_beam;      /* 10.0 (recipes: 8.0) */
  float beam_delta;        /* 0.5 */
  /* capacities.  The per-FRAME ones are hard limits (exceeding one is K3_ERR_OVERFLOW for that utterance, never a silent beam change -- contrast
   * cuda-decoder.cc:944-976).  The per-LANE ones (tokens / links kept for every frame until FinalizeDecoding) are a RESERVATION, like the reference's
   * ntokens_pre_allocated (cuda-decoder.cc:232-238: reserve(), the per-channel vectors grow): a lane that outgrows them moves, inside the token-passing
   * kernel, to pools of at least twice the size carved off the decoder's spare arena (spare_pool_bytes below); K3_ERR_OVERFLOW only when that arena is exhausted. */
  int32_t frame_tokens_cap;   /* max tokens alive on one frame of one lane (hash table = 2x next pow2) */
  int32_t frame_cands_cap;    /* max emitting arcs that pass the pre-pass bound on one frame */
  int64_t lane_tokens_cap;    /* tokens of all frames of one lane: the reservation */
  int64_t lane_links_cap;     /* forward links of all frames of one lane: the reservation */
  /* literal_order = 1: reproduce the reference's SERIAL algorithm bit for bit -- next_cutoff tightened while the previous frame's tokens are
   * visited in HashList order (lattice-faster-decoder.cc:779-797, util/hash-list-inl.h:125-165), the first minimum-cost token of that list as
   * the best token, PruneForwardLinksFinal's in-place sweeps with its 1e-5 stop rule (:385-467).  The raw lattice then equals
   * LatticeFasterDecoder::GetRawLattice exactly (states, arcs, labels, cost bits); token passing costs about 2-3x the default mode, which
   * applies the frame's FINAL bound to every arc (order-independent, a sub-lattice with the same best path).  hash_ratio =
   * LatticeFasterDecoderConfig::hash_ratio (2.0): it decides the bucket count and with it the reference's visit order.
   * Needs frame_tokens_cap <= 65536 < frame_cands_cap. */
  int32_t literal_order;      /* 0; 1 = on; 2 = on, the closure's creation order by the one-wavefront replay (the fall-back path of 1, for A/B); 3 = 1 with the fall-back forced (tests); 4 = 1 with the hash-order passes of frames above 3072 tokens on their HBM fall-back form instead of the LDS-partition form (tests) */
  float hash_ratio;           /* 2.0 */
  /* literal_order = 1 only: frames of at most this many tokens (before and after) are processed entirely in LDS (k3_decoder_fast.h), the others on the
   * general path; same results either way.  -1 = the kernel's capacity (3072), 0 = general path only (A/B, tests), n = a smaller capacity (tests). */
  int32_t fast_frame_tokens;  /* -1 */
  /* HBM set aside for lanes that outgrow lane_tokens_cap / lane_links_cap (16 B per token + 20 B per link of the new pools; a lane's old pools are not reused):
   * -1 = a quarter of the lanes' reservation, at least 1 GiB and 14 x one lane's reservation (a lane growing 2x, 4x, 8x); 0 = none (the reservation is then a hard limit). */
  int64_t spare_pool_bytes;   /* -1 */
  /* literal_order only -- the token-passing launch's shape.  resident_lanes = 0: one workgroup per utterance, all in flight at once.  n > 0 (and fewer than the call's
   * utterances): n workgroups take the utterances from a work-queue, longest first (the reference reschedules lanes per chunk: cuda-online-pipeline-dynamic-batcher.cc;
   * batched-threaded-nnet3-cuda-online-pipeline.cc:316-413), so a batch of unequal lengths keeps every slot busy and a launch does not last as long as its longest lane
   * x the lanes per slot.  resident_exclusive = 1: a workgroup asks for more than half of a CU's LDS -- one lane per CU, the other half of the CU (registers, LDS, issue
   * slots) is left to whatever else is queued on the device, i.e. the next batch's features and network.  Same lattices in every shape. */
  int32_t resident_lanes;     /* 0 */
  int32_t resident_exclusive; /* 0 */
} k3_decoder_config;
void k3_decoder_config_default(k3_decoder_config *cfg);
typedef struct k3_decoder k3_decoder;
/* nlanes = max utterances decoded by one k3_decoder_decode_batch call; the fst must outlive the decoder
 * (cuda-decoder.h:220). num_pdfs = columns of the log-likelihood matrix. */
int k3_decoder_create(const k3_fst *fst, const k3_decoder_config *cfg, int32_t nlanes, int32_t num_pdfs, k3_decoder **dec);
void k3_decoder_destroy(k3_decoder *dec);
/* Decode num_utts utterances: utterance u owns rows h_row_offsets[u] .. h_row_offsets[u+1] of d_loglikes
 * (row-major, leading dimension ld; already scaled by the acoustic scale, as DecodableAmNnetSimple hands them
 * over).  Asynchronous on `stream` (the call first waits for what is already queued on `stream`, never for its own kernels): the caller
 * can queue the NEXT batch's features and network on a second stream right behind it -- their workgroups take the CUs this batch's lanes
 * free as they finish (bench.py, batched-wav-nnet3-cuda2).  Results are fetched with the calls below (which synchronise). */
int k3_decoder_decode_batch(k3_decoder *dec, int32_t num_utts, const float *d_loglikes, int64_t ld,
                            const int64_t *h_row_offsets, void *stream);
/* The same in the pieces of CudaDecoder's online interface (cuda-decoder.h:248-262, lanes == channels here): InitDecoding for num_utts
 * lanes, AdvanceDecoding chunk by chunk (lane u consumes rows h_row_offsets[u]..[u+1] of this call's d_loglikes as its next frames; an
 * empty range idles the lane), FinalizeDecoding = lattice-beam pruning with final-probs.  Chunked calls give results bit-identical to
 * k3_decoder_decode_batch.  max_total_frames bounds the frames a lane receives between init and finalize. */
int k3_decoder_init_decoding(k3_decoder *dec, int32_t num_utts, int32_t max_total_frames, void *stream);
int k3_decoder_advance_decoding(k3_decoder *dec, int32_t num_utts, const float *d_loglikes, int64_t ld, const int64_t *h_row_offsets, void *stream);
/* The reference's own form of the call, CudaDecoder::AdvanceDecoding(lanes_assignements) (cuda-decoder.h:262): each listed channel gets a DEVICE
 * pointer to the log-likelihoods of its next num_frames frames (rows ld floats apart), wherever they live; the other channels of the group idle. */
int k3_decoder_advance_decoding_lanes(k3_decoder *dec, int32_t num_channels, const int32_t *channels, const float *const *h_lane_frames, int32_t num_frames,
    int64_t ld, void *stream);
/* ... with a frame count per channel and the rows of a channel `ld` floats apart: log-likelihoods decoded where a producer left them, e.g. the time-major output of
 * k3_nnet_stream_forward (channel c's row k at k * num_channels + c: ld = num_channels * its row length).  h_lane_first[u] null or h_num_frames[u] = 0: channel u idles. */
int k3_decoder_advance_decoding_strided(k3_decoder *dec, int32_t num_utts, const float *const *h_lane_first, const int32_t *h_num_frames, int64_t ld, void *stream);
int k3_decoder_finalize_decoding(k3_decoder *dec, void *stream);
/* Channels with independent lifetimes inside one lane group (CudaDecoder::InitDecoding(channels) cuda-decoder.h:248 / the per-channel
 * end of an utterance in the online pipeline): k3_decoder_init_channels restarts the listed lanes (start token + eps closure at their
 * next k3_decoder_advance_decoding), k3_decoder_finalize_channels runs FinalizeDecoding on the listed lanes only; afterwards
 * k3_decoder_lattice_info / k3_decoder_get_raw_lattices return exactly those lanes, in the order given, while the other lanes of the
 * group keep their state and go on decoding.  A lane is a channel here: its tokens and links live in its own HBM pools. */
int k3_decoder_init_channels(k3_decoder *dec, const int32_t *channels, int32_t num_channels, void *stream);
int k3_decoder_finalize_channels(k3_decoder *dec, const int32_t *channels, int32_t num_channels, void *stream);
int32_t k3_decoder_num_frames_decoded(const k3_decoder *dec, int32_t utt);     /* NumFramesDecoded(channel) */
/* Per utterance: [0] lattice states, [1] lattice arcs, [2] status (0 ok, 1 no surviving tokens, <0 k3_status),
 * [3] reached_final (a final-state token was active on the last frame), [4] tokens created, [5] links created,
 * [6] max tokens on one frame, [7] emitting arcs traversed, [8] epsilon arcs traversed, [9] frames.
 * h_info: [num_utts x 10] int64. */
int k3_decoder_lattice_info(k3_decoder *dec, int64_t *h_info);
/* GetBestPath (cuda-decoder.h:306) and the traceback behind GetPartialHypothesis / EndpointDetected (:286-295, cuda-decoder.cc:1864-1960): the
 * one-best path of each listed lane from the tokens it holds NOW -- after k3_decoder_advance_decoding (a partial result: use_final_probs = 0)
 * or after finalisation.  Path u owns entries h_offsets[u] .. h_offsets[u+1] (n + 1 offsets) of the four arc arrays, first arc first; arc weight =
 * LatticeWeight(graph, acoustic) as in GetRawLattice.  h_final_cost[u] = the final cost that was added (0 when none); h_relative_cost[u] =
 * FinalRelativeCost() = min(cost + final) - min(cost) over the newest frame (+inf: no final state active), the quantity kaldi::EndpointDetected
 * takes; h_reached_final[u]; the three may be NULL.  cap_arcs = capacity of the arc arrays (K3_ERR_OVERFLOW when too small). */
int k3_decoder_get_best_path(k3_decoder *dec, const int32_t *channels, int32_t num_channels, int32_t use_final_probs, int64_t *h_offsets, int64_t cap_arcs,
                             int32_t *h_ilabel, int32_t *h_olabel, float *h_graph, float *h_ac, float *h_final_cost, float *h_relative_cost, int32_t *h_reached_final);
/* SURVEY 9.1 "order-sensitive events", per finalised utterance (same order as k3_decoder_lattice_info).  literal_order: forward links that exist
 * only because next_cutoff was still loose when their arc was examined (tot >= the frame's final next_cutoff); default mode: emitting arcs below
 * the pre-pass bound but not below the final bound (an upper bound on the arcs the serial and the two-pass rule can disagree on). */
int k3_decoder_order_sensitive_events(k3_decoder *dec, int64_t *h_events);
/* h_growths[u]: how often the lane of finalised utterance u has moved to bigger token / link pools since the decoder was created (0 while the reservation holds) */
int k3_decoder_pool_growths(k3_decoder *dec, int32_t *h_growths);
/* GetRawLattice for every utterance of the last batch, concatenated in utterance order (utterance u owns
 * states h_state_offsets[u]..[u+1] and arcs h_arc_offsets[u]..[u+1]; arc endpoints are indices local to the
 * utterance).  All output pointers are HOST buffers sized from k3_decoder_lattice_info.
 * arc weight = LatticeWeight(graph_cost, acoustic_cost - cost_offset) (lattice-faster-decoder.cc:174-181);
 * final_cost = +inf for non-final lattice states. */
int k3_decoder_get_raw_lattices(k3_decoder *dec, int32_t *h_st_frame, int32_t *h_st_state, float *h_st_cost,
                                float *h_st_final, int32_t *h_arc_src, int32_t *h_arc_dst, int32_t *h_arc_ilabel,
                                int32_t *h_arc_olabel, float *h_arc_graph, float *h_arc_ac);
/* HIP-event timing of the two decode kernels of the last batch on their launch stream: h_ms[0] = token passing
 * (k3_decode_forward_kernel), h_ms[1] = lattice-beam pruning (k3_decode_prune_kernel). */
int k3_decoder_set_profiling(k3_decoder *dec, int32_t on);
/* `stream` waits (on the device) for the decoder's latest token-passing launch: the last reader of the log-likelihoods handed to k3_decoder_decode_batch / k3_decoder_advance_decoding*.
 * What a pipelined caller puts in front of the kernels that refill that buffer, instead of waiting for the pruning / output kernels as well.  No-op before the first launch. */
int k3_decoder_stream_wait_token_passing(k3_decoder *dec, void *stream);
int k3_decoder_kernel_times(k3_decoder *dec, float *h_ms);
/* developer aid: per-phase shader-clock totals of the token-passing kernel (all zero unless the library was built with -DK3_DEC_PROF) */
int k3_decoder_phase_cycles(k3_decoder *dec, int64_t *h_cycles /* [16] */);
/* per-frame diagnostics of one utterance of the last batch (host arrays of length num_frames, any may be NULL):
 * tokens seen by GetCutoff, cur_cutoff, adaptive_beam, next_cutoff, cost_offset */
int k3_decoder_frame_stats(k3_decoder *dec, int32_t utt, int32_t *h_ntoks, float *h_cur_cutoff, float *h_adaptive_beam,
                           float *h_next_cutoff, float *h_cost_offset);

/* ---------------------------------------------------------------- CuMatrix operations --------
 * The CuMatrixBase<BaseFloat> methods the nnet3 forward pass of a TDNN / TDNN-F model executes through Kaldi's generic
 * NnetComputer, so that a Kaldi build can keep its graph compiler/executor and swap only the device kernels:
 * each entry point is the body of the CuMatrixBase method of the same name (cudamatrix/cu-matrix.h:79-791; kernels
 * cudamatrix/cu-kernels.cu).  Row-major float32, leading dimension = CuMatrixBase::Stride(), device pointers. */
/* CuMatrixBase::SoftMaxPerRow (op 0) / LogSoftMaxPerRow (1) (cudamatrix/cu-matrix.h:328,334): dst = f(a), row by row, in place allowed; DiffSoftmaxPerRow (2: a = value, b = diff) /
 * DiffLogSoftmaxPerRow (3: a = out_value, b = out_deriv) (:403,:411) */
int k3_mat_softmax_rows(int32_t op, float *d_dst, int64_t ldd, const float *d_a, int64_t lda, const float *d_b, int64_t ldb, int32_t rows, int32_t cols, void *stream);
/* cu::NormalizePerRow (op 0) / cu::DiffNormalizePerRow (op 1) (cudamatrix/cu-math.h:272-300, cu-math.cc:280-409; NormalizeComponent): op 0: dst [rows x cols (+1 with add_log_stddev)] from
 * in [rows x cols]; op 1: dst = in_deriv, ADDED to (kBackpropAdds) unless it aliases out_deriv (the in-place backprop), out_deriv [rows x cols (+1)] */
int k3_mat_normalize_rows(int32_t op, float *d_dst, int64_t ldd, const float *d_in, int64_t ldi, const float *d_out_deriv, int64_t ldo, int32_t rows,
    int32_t cols, float target_rms, int32_t add_log_stddev, void *stream);
/* CuMatrixBase::Sigmoid (op 0) / Tanh (1) / Log (2) / Pow (3: power a) / PowAbs (4: power a, flag = include_sign) / Max (5) (cudamatrix/cu-matrix.h:288-307,:386,:501): dst = f(src), in place allowed */
int k3_mat_apply_map(int32_t op, float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, float a, int32_t flag, void *stream);
/* CuMatrixBase::DiffSigmoid (op 0) / DiffTanh (1) (cudamatrix/cu-matrix.h:390-396): dst = diff .* value .* (1 - value) | diff .* (1 - value^2) */
int k3_mat_diff_activation(int32_t op, float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_value, int64_t ldv, const float *d_diff, int64_t ldf, void *stream);
int k3_mat_mul_rows(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, const int32_t *d_indexes, void *stream);   /* MulRows (cudamatrix/cu-matrix.h:148): row r *= src row indexes[r]; -1 = unchanged (dropout masks per sequence) */
/* CuMatrixBase::SetMatMatDivMat (op 0: dst = A .* (B ./ C3), = A where C3 is 0: DropoutComponent::Backprop) / AddMatMatElements (op 1: dst = beta dst + alpha A .* B) (cudamatrix/cu-matrix.h:580,:608) */
int k3_mat_elements3(int32_t op, float *d_C, int64_t ldc, int32_t rows, int32_t cols, float alpha, const float *d_A, int64_t lda, const float *d_B,
    int64_t ldb, const float *d_C3, int64_t ldc3, float beta, void *stream);
int k3_mat_div_rows_vec(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_div, void *stream);            /* DivRowsVec: row r divided by div[r] */
int k3_mat_copy_cols_from_vec(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_col, void *stream);      /* CopyColsFromVec with a vector of dimension rows: every column = v */
int k3_mat_copy_cols(int32_t add, float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, const int32_t *d_indexes, void *stream);   /* CopyCols (add 0) / AddCols (1): dst(r, c) (+)= src(r, indexes[c]), -1 = zero / skip */
/* CuRand<BaseFloat>::RandUniform (kind 0: [0, 1)) / RandGaussian (kind 1) (cudamatrix/cu-rand.h:50-56): Philox-4x32-10 keyed by `seed`, element i of the logical rows x cols matrix from
 * counter offset + i / 4 -- reproducible for (seed, offset) independent of stride and launch shape;
 a fill consumes ceil(rows * cols / 4) counters.  (The reference's device stream is
 * cuRAND's and its CPU stream is rand(): neither is reproduced; parity for this entry point is distributional.) */
int k3_mat_set_rand(int32_t kind, float *d_C, int64_t ldc, int32_t rows, int32_t cols, uint64_t seed, uint64_t offset, void *stream);
int64_t k3_mat_gemm_flops(int32_t reset);      /* 2 M N K summed over this process's k3_mat_add_mat_mat calls (reset != 0: read and clear) -- the flop count of a training iteration for its roofline */
int k3_mat_add_mat_mat(float alpha, const float *d_A, int64_t lda, int32_t trans_a, const float *d_B, int64_t ldb, int32_t trans_b, float beta,
                       float *d_C, int64_t ldc, int32_t M, int32_t N, int32_t K, void *stream);      /* AddMatMat: C = alpha op(A) op(B) + beta C, FP32 MFMA */
int k3_mat_set(float *d_C, int64_t ldc, int32_t rows, int32_t cols, float value, void *stream);                 /* Set / SetZero */
int k3_mat_scale(float *d_C, int64_t ldc, int32_t rows, int32_t cols, float value, void *stream);               /* Scale */
int k3_mat_add(float *d_C, int64_t ldc, int32_t rows, int32_t cols, float value, void *stream);                 /* Add */
int k3_mat_apply_floor(float *d_C, int64_t ldc, int32_t rows, int32_t cols, float floor_val, void *stream);     /* ApplyFloor (ReLU = floor 0) */
int k3_mat_apply_ceiling(float *d_C, int64_t ldc, int32_t rows, int32_t cols, float ceiling_val, void *stream); /* ApplyCeiling */
int k3_mat_copy_rows_from_vec(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_v, void *stream);    /* CopyRowsFromVec (bias broadcast) */
int k3_mat_mul_cols_vec(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_scale, void *stream);      /* MulColsVec (BatchNorm scale) */
int k3_mat_mul_rows_vec(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_scale, void *stream);      /* MulRowsVec */
int k3_mat_add_vec_to_rows(float alpha, const float *d_row, float beta, float *d_C, int64_t ldc, int32_t rows, int32_t cols, void *stream);  /* AddVecToRows (BatchNorm offset) */
int k3_mat_add_vec_to_cols(float alpha, const float *d_col, float beta, float *d_C, int64_t ldc, int32_t rows, int32_t cols, void *stream);  /* AddVecToCols */
int k3_mat_copy_from_mat(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, int32_t trans, void *stream);  /* CopyFromMat */
int k3_mat_add_mat(float alpha, const float *d_A, int64_t lda, int32_t trans_a, float *d_C, int64_t ldc, int32_t rows, int32_t cols, void *stream); /* AddMat */
int k3_mat_copy_rows(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, const int32_t *d_indexes, void *stream); /* CopyRows, index -1 = zero row */
int k3_mat_add_rows(float alpha, const float *d_src, int64_t lds, const int32_t *d_indexes, float *d_C, int64_t ldc, int32_t rows, int32_t cols, void *stream); /* AddRows, index -1 = skip */

int k3_mat_mul_elements(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_A, int64_t lda, void *stream);   /* MulElements */
int k3_mat_heaviside(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, void *stream);   /* Heaviside(src): 1 where src > 0, else 0 (ReLU backprop) */
int k3_mat_add_mat_diag_vec(float alpha, const float *d_M, int64_t ldm, int32_t trans_m, const float *d_v, float beta, float *d_C, int64_t ldc, int32_t rows, int32_t cols, void *stream); /* AddMatDiagVec: C = beta C + alpha M diag(v) */
int k3_mat_add_row_ranges(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, int32_t src_rows, const int32_t *d_ranges /* [rows][2] */, void *stream); /* AddRowRanges: C[r] += sum of src rows [first, second) */
int k3_mat_copy_lower_to_upper(float *d_C, int64_t ldc, int32_t n, void *stream);                                 /* CopyLowerToUpper (after SymAddMat2) */
int k3_mat_add_to_diag(float *d_C, int64_t ldc, int32_t rows, int32_t cols, float value, void *stream);            /* AddToDiag */
int k3_mat_add_vec_vec(float alpha, const float *d_x, const float *d_y, float *d_C, int64_t ldc, int32_t rows, int32_t cols, void *stream);   /* AddVecVec: C += alpha x y^T */
int k3_mat_add_diag_vec_mat(float alpha, const float *d_v, const float *d_M, int64_t ldm, int32_t trans_m, float beta, float *d_C, int64_t ldc, int32_t rows, int32_t cols, void *stream); /* AddDiagVecMat: C = beta C + alpha diag(v) M */
int k3_mat_div_elements(float *d_C, int64_t ldc, int32_t rows, int32_t cols, const float *d_A, int64_t lda, void *stream);                    /* DivElements */
/* Scalar reductions, result to the host (synchronises the stream): op 0 TraceMatMat(A, B, kTrans) = VecVec = sum A(i,j) B(i,j); 1 TraceMatMat(A, B, kNoTrans) = sum A(i,j) B(j,i)
 * (B is cols x rows); 2 Trace(A) (square); 3 Sum; 4 Max; 5 Min.  Partial sums in double, folded in a fixed order (cudamatrix/cu-matrix.h:80-95,658-671, cu-vector.h:36-40). */
int k3_mat_reduce_scalar(int32_t op, const float *d_A, int64_t lda, const float *d_B, int64_t ldb, int32_t rows, int32_t cols, double *h_result, void *stream);
/* Reductions into a vector (CuVectorBase): op 0 AddRowSumMat: v[c] = beta v[c] + alpha sum_r M(r, c); 1 AddDiagMat2(M, kTrans): sum_r M(r, c)^2; 2 AddDiagMatMat(M, kTrans, N, kNoTrans):
 * sum_r M(r, c) N(r, c); 3 AddDiagMat2(M, kNoTrans): v[r] = .. sum_c M(r, c)^2; 4 AddColSumMat: v[r] = .. sum_c M(r, c).  Sums in double. */
int k3_vec_col_reduce(int32_t op, float alpha, const float *d_M, int64_t ldm, const float *d_N, int64_t ldn, int32_t rows, int32_t cols, float beta, float *d_v, void *stream);

/* CuVectorBase: a vector is a [1 x dim] matrix for Set / Add / Scale / ApplyFloor / AddVec (k3_mat_add_vec_to_rows) / MulElements
 * (k3_mat_mul_cols_vec); the three operations below have no matrix counterpart (cudamatrix/cu-vector.h:79-103,147-160). */
int k3_vec_convert(const void *d_src, int32_t src_is_f64, void *d_dst, int32_t dst_is_f64, int32_t n, void *stream);   /* CopyFromVec(const CuVectorBase<OtherReal>&) */
int k3_vec_pow(const float *d_src, float *d_dst, int32_t n, float power, void *stream);                                /* Pow / ApplyPow */
int k3_vec_add_vec_vec(float alpha, const float *d_a, const float *d_b, float beta, float *d_v, int32_t n, void *stream);  /* AddVecVec: v = alpha a .* b + beta v */
int k3_vec_unary(int32_t op, float *d_v, int32_t n, void *stream);   /* op 0 ApplyLog, 1 ApplyExp, 2 InvertElements */
/* CuVectorBase<double> (a model's accumulated statistics): op 0 Scale(alpha); 1 Pow: v = a ^ alpha; 2 AddVec: v = alpha a + beta v; 3 AddVecVec: v = alpha a .* b + beta v */
int k3_vec_f64(int32_t op, double alpha, const double *d_a, const double *d_b, double beta, double *d_v, int32_t n, void *stream);

/* ---- chain training, first slice: the LF-MMI denominator.  Stands behind chain::DenominatorGraph's constructor
 * (chain/chain-den-graph.cc:29-143) and chain::DenominatorComputation::Forward / Backward (chain/chain-denominator.h:203-318,
 * chain-denominator.cc:106-440; GPU reference chain/chain-kernels.cu:108-296).
 * k3_chain_den_create: the denominator FST as host CSR arrays -- arcs of state s are arc_offsets[s] .. arc_offsets[s+1], ilabel = pdf-id + 1
 * (what CreateDenominatorFst writes), weight = -log transition probability, final_cost[s] = -log final probability (+inf: not final; only
 * the initial probabilities' normalisation uses it).  Transitions by source and by destination and the initial probabilities are built as
 * the reference's constructor builds them.
 * k3_chain_den_forward_backward: d_nnet_output is the network's output for num_sequences sequences of frames_per_sequence frames, row
 * t * num_sequences + s = frame t of sequence s (the layout of a chain minibatch), ld floats apart.  *h_objf = the total log-probability of the
 * minibatch (Forward()'s return value).  d_nnet_output_deriv (may be NULL: forward only) += deriv_weight * posterior of each pdf on each
 * frame (Backward()'s contract; chain training passes -supervision.weight); *h_ok (may be NULL) = Backward()'s return value: 0 when the
 * alpha-beta check of chain-denominator.cc:404-440 fails and the minibatch should be abandoned.  Synchronises the stream.
 * One launch for the whole minibatch, one workgroup per sequence; K3_ERR_UNSUPPORTED when 16 states + 12 pdfs bytes exceed 150 KB of LDS. */
typedef struct k3_chain_den k3_chain_den;
int k3_chain_den_create(int32_t num_states, int32_t start, int32_t num_pdfs, const int64_t *arc_offsets, const int32_t *ilabel, const int32_t *nextstate,
                        const float *weight, const float *final_cost, k3_chain_den **den);
void k3_chain_den_destroy(k3_chain_den *den);
int k3_chain_den_num_states(const k3_chain_den *den);
int k3_chain_den_initial_probs(const k3_chain_den *den, float *h_probs /* [num_states] */);   /* DenominatorGraph::InitialProbs() */
int k3_chain_den_forward_backward(k3_chain_den *den, const float *d_nnet_output, int64_t ld, int32_t num_sequences, int32_t frames_per_sequence,
                                  float leaky_hmm_coefficient, float deriv_weight, float *d_nnet_output_deriv, int64_t ld_deriv, float *h_objf, int32_t *h_ok, void *stream);

/* The numerator and the objective (chain::NumeratorComputation chain/chain-numerator.{h,cc}; chain::ComputeChainObjfAndDeriv chain/chain-training.cc:242-337,
 * the branch for ordinary -- not end-to-end -- supervisions).
 * k3_chain_supervision_create: the supervision FSTs of the minibatch's num_sequences sequences, UNMERGED (what the examples hold before
 * MergeSupervision concatenates them; sequence n's states are state_offsets[n] .. state_offsets[n+1] of one CSR numbering, arcs of global state s
 * arc_offsets[s] .. arc_offsets[s+1], nextstate local to the sequence, ilabel = pdf-id + 1).  Each must have the properties the reference asserts
 * (chain-supervision.cc:663-700): start state 0, epsilon-free, states sorted by path length, every path frames_per_sequence arcs long.  weight =
 * Supervision::weight.  The reference walks the merged FST serially on the CPU; here every sequence is walked by its own wavefront.
 * k3_chain_numerator = NumeratorComputation::Forward (+ Backward when d_nnet_output_deriv is not NULL: += weight * occupation probabilities).
 * k3_chain_objf_and_deriv = ComputeChainObjfAndDeriv: derivative zeroed, denominator (deriv -= weight * den posteriors), out-of-range penalty
 * (the reference applies it on a coin flip, RandInt(0, 1): here when opts->apply_out_of_range_penalty), numerator (into d_xent_output_deriv when
 * given, then added), objf = numerator - denominator log-probs (weighted), weight = supervision weight * sequences * frames, the "-10 per frame"
 * fall-back with zeroed derivatives when the objective is not finite or the denominator's check fails, l2 term.  Pointers to the two derivative
 * matrices may be NULL (objective only).  Synchronises the stream. */
typedef struct k3_chain_training_opts {      /* chain::ChainTrainingOptions (chain/chain-training.h:45-104) */
  float l2_regularize;               /* 0.0 */
  float out_of_range_regularize;     /* 0.01 */
  float leaky_hmm_coefficient;       /* 1.0e-05 */
  int32_t apply_out_of_range_penalty;
} k3_chain_training_opts;
typedef struct k3_chain_supervision k3_chain_supervision;
int k3_chain_supervision_create(int32_t num_sequences, int32_t frames_per_sequence, int32_t label_dim, float weight, const int32_t *state_offsets, const int64_t *arc_offsets,
                                const int32_t *ilabel, const int32_t *nextstate, const float *arc_weight, const float *final_cost, k3_chain_supervision **sup);
/* End-to-end (flat-start) supervisions: chain::Supervision::e2e_fsts, one FST per sequence that may have self-loops and several final states (epsilon-free, ilabel = pdf-id + 1,
 * start state 0;
 same array layout as above).  k3_chain_numerator / k3_chain_objf_and_deriv then run chain::GenericNumeratorComputation (chain/chain-generic-numerator.cc:30-463) and
 * the end-to-end branch of ComputeChainObjfAndDeriv (chain-training.cc:86-215) -- including the reference's quirk that the numerator log-probability enters the objective without the
 * supervision weight (:270) while its derivative carries it. */
int k3_chain_supervision_create_e2e(int32_t num_sequences, int32_t frames_per_sequence, int32_t label_dim, float weight, const int32_t *state_offsets, const int64_t *arc_offsets,
                                    const int32_t *ilabel, const int32_t *nextstate, const float *arc_weight, const float *final_cost, k3_chain_supervision **sup);
void k3_chain_supervision_destroy(k3_chain_supervision *sup);
int k3_chain_numerator(k3_chain_supervision *sup, const float *d_nnet_output, int64_t ld, float *d_nnet_output_deriv, int64_t ld_deriv, float *h_logprob_weighted, void *stream);
int k3_chain_objf_and_deriv(k3_chain_den *den, k3_chain_supervision *sup, const k3_chain_training_opts *opts, const float *d_nnet_output, int64_t ld,
                            float *d_nnet_output_deriv, int64_t ld_deriv, float *d_xent_output_deriv, int64_t ld_xent, float *h_objf, float *h_l2_term,
                                float *h_weight, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* K3HIP_H_ */
