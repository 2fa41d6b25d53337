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
/* Per-utterance CMVN in place: AccCmvnStats + ApplyCmvn (transform/cmvn.cc:30-115), what
 * `compute-cmvn-stats | apply-cmvn [--norm-vars]` do with one utterance per speaker.
 * fp64 accumulators like the reference.  d_stats (optional, may be NULL): [U x 2 x (dim+1)] doubles. */
int k3_cmvn_offline_batch(float *d_feats, int64_t ld, int32_t dim, const int64_t *d_frame_offsets,
                          int32_t num_utts, int32_t norm_vars, double *d_stats, void *stream);

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

#ifdef __cplusplus
}
#endif
#endif /* K3HIP_H_ */
