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

#ifdef __cplusplus
}
#endif
#endif /* K3HIP_H_ */
