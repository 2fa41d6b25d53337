/* k3_cuda_features.h -- kaldi::CudaSpectralFeatures with the reference's signatures (cudafeat/feature-spectral-cuda.h:34-107) over the C ABI of k3hip.h.
 *
 * For a Kaldi build that keeps its own CuMatrix / CuVector types (here: the reference's declarations over kaldi_amd/adapter/cu-k3.cc, whose storage lives in
 * HBM): the constructor takes the reference's CudaSpectralFeatureOptions (made from MfccOptions or FbankOptions exactly as there), ComputeFeatures takes the
 * waveform as a CuVectorBase<BaseFloat> on the device and fills a CuMatrix<BaseFloat>.  One fused kernel (k3_feat_compute_batch) replaces the reference's
 * ExtractWindows / ProcessWindows / cuFFT / mel / DCT sequence; results are held to the reference's CPU compute-fbank-feats / compute-mfcc-feats at 1e-4
 * (tests/test_adapter_gpu.py).  Header-only; needs the reference's feat/feature-mfcc.h, feat/feature-fbank.h and cudamatrix headers on the include path. */
#ifndef K3_CUDA_FEATURES_H_
#define K3_CUDA_FEATURES_H_
#include <map>
#include <hip/hip_runtime_api.h>
#include "cudamatrix/cu-matrix.h"
#include "cudamatrix/cu-vector.h"
#include "feat/feature-fbank.h"
#include "feat/feature-mfcc.h"
#include "k3hip.h"

namespace kaldi {
enum SpectralFeatureType { MFCC, FBANK };
struct CudaSpectralFeatureOptions {      /* cudafeat/feature-spectral-cuda.h:36-68 */
  MfccOptions mfcc_opts; bool use_log_fbank, use_power, use_dct; SpectralFeatureType feature_type;
  CudaSpectralFeatureOptions(MfccOptions opts_in) : mfcc_opts(opts_in), use_log_fbank(true), use_power(true), use_dct(true), feature_type(MFCC) {}
  CudaSpectralFeatureOptions(FbankOptions opts) {
    mfcc_opts.frame_opts = opts.frame_opts; mfcc_opts.mel_opts = opts.mel_opts; mfcc_opts.use_energy = opts.use_energy; mfcc_opts.energy_floor = opts.energy_floor;
    mfcc_opts.raw_energy = opts.raw_energy; mfcc_opts.htk_compat = opts.htk_compat; mfcc_opts.cepstral_lifter = 0.0f;
    use_log_fbank = opts.use_log_fbank; use_power = opts.use_power; use_dct = false; feature_type = FBANK;
  }
  CudaSpectralFeatureOptions() : use_log_fbank(true), use_power(true), use_dct(true), feature_type(MFCC) {}
};

class CudaSpectralFeatures {
 public:
  CudaSpectralFeatureOptions cumfcc_opts_;
  explicit CudaSpectralFeatures(const CudaSpectralFeatureOptions &opts) : cumfcc_opts_(opts) { Plan(1.0f); }
  ~CudaSpectralFeatures() { for (auto &p : plans_) k3_feat_plan_destroy(p.second); if (d_off_) (void)hipFree(d_off_); }
  CudaSpectralFeatures(const CudaSpectralFeatures &) = delete; CudaSpectralFeatures &operator=(const CudaSpectralFeatures &) = delete;
  int32 Dim() { return k3_feat_dim(Plan(1.0f)); }
  const FrameExtractionOptions &GetFrameOptions() const { return cumfcc_opts_.mfcc_opts.frame_opts; }
  /* cudafeat/feature-spectral-cuda.cu:525-567 */
  void ComputeFeatures(const CuVectorBase<BaseFloat> &cu_wave, BaseFloat sample_freq, BaseFloat vtln_warp, CuMatrix<BaseFloat> *cu_features) {
    if (sample_freq != cumfcc_opts_.mfcc_opts.frame_opts.samp_freq) KALDI_ERR << "Waveform and config sample Frequency mismatch: " << sample_freq << " .vs " <<
        cumfcc_opts_.mfcc_opts.frame_opts.samp_freq;
    k3_feat_plan *plan = Plan(vtln_warp);
    const int64_t nsamp = cu_wave.Dim(), nframes = k3_feat_num_frames(plan, nsamp);
    cu_features->Resize((int32)nframes, k3_feat_dim(plan), kUndefined);
    if (nframes == 0) return;
    if (!d_off_ && hipMalloc((void **)&d_off_, 4 * sizeof(int64_t)) != hipSuccess) KALDI_ERR << "hipMalloc failed";
    const int64_t h[4] = {0, nsamp, 0, nframes};
    if (hipMemcpy(d_off_, h, sizeof h, hipMemcpyHostToDevice) != hipSuccess) KALDI_ERR << "hipMemcpy failed";
    if (k3_feat_compute_batch(plan, cu_wave.Data(), d_off_, d_off_ + 2, 1, nframes, cu_features->Data(), cu_features->Stride(), NULL) != K3_OK) KALDI_ERR <<
        "k3_feat_compute_batch: " << k3_last_error();
    if (hipDeviceSynchronize() != hipSuccess) KALDI_ERR << "feature kernel failed";
  }
 private:
  k3_feat_plan *Plan(float vtln_warp) {      /* window / mel-bank / DCT tables depend on the warp factor: one plan per factor seen */
    auto it = plans_.find(vtln_warp); if (it != plans_.end()) return it->second;
    const MfccOptions &m = cumfcc_opts_.mfcc_opts; const FrameExtractionOptions &f = m.frame_opts; const MelBanksOptions &b = m.mel_opts;
    k3_feat_opts o; memset(&o, 0, sizeof o);
    o.samp_freq = f.samp_freq;
    o.frame_shift_ms = f.frame_shift_ms;
    o.frame_length_ms = f.frame_length_ms;
    o.dither = f.dither;
    o.preemph_coeff = f.preemph_coeff;
    o.blackman_coeff = f.blackman_coeff;
    o.remove_dc_offset = f.remove_dc_offset; o.round_to_power_of_two = f.round_to_power_of_two; o.snip_edges = f.snip_edges;
    const char *names[] = {"hanning", "sine", "hamming", "povey", "rectangular", "blackman"}; o.window_type = -1;
    for (int i = 0; i < 6; i++) if (f.window_type == names[i]) o.window_type = i;
    if (o.window_type < 0) KALDI_ERR << "Invalid window type " << f.window_type;
    o.num_bins = b.num_bins; o.low_freq = b.low_freq; o.high_freq = b.high_freq; o.vtln_low = b.vtln_low; o.vtln_high = b.vtln_high; o.htk_mode = b.htk_mode;
    o.use_energy = m.use_energy;
    o.energy_floor = m.energy_floor;
    o.raw_energy = m.raw_energy;
    o.htk_compat = m.htk_compat;
    o.use_log_fbank = cumfcc_opts_.use_log_fbank;
    o.use_power = cumfcc_opts_.use_power;
    o.num_ceps = m.num_ceps; o.cepstral_lifter = m.cepstral_lifter; o.feature_type = cumfcc_opts_.feature_type == MFCC ? 1 : 0; o.vtln_warp = vtln_warp;
    k3_feat_plan *p = NULL;
    if (k3_feat_plan_create(&o, &p) != K3_OK) KALDI_ERR << "k3_feat_plan_create: " << k3_last_error();
    plans_[vtln_warp] = p; return p;
  }
  std::map<float, k3_feat_plan *> plans_; int64_t *d_off_ = NULL;
};
}  /* namespace kaldi */
#endif
