// option registration shared by the CLI programs: the flag names of FrameExtractionOptions (feat/feature-window.h:69-101),
// MelBanksOptions (feat/mel-computations.h:60-80), FbankOptions (feat/feature-fbank.h:62-79), MfccOptions (feat/feature-mfcc.h:62-79)
#pragma once
#include "k3_host.h"
#include "../../include/k3hip.h"
#include <hip/hip_runtime.h>
namespace k3host {
// OfflineFeatureTpl::ComputeFeatures' rate check (feat/feature-common-inl.h:29-57): true = the wave is (now) at `want` Hz; false = mismatch without the flag that allows it (the
// reference raises an error there, which the feature programs turn into a warning and a skipped file).  Resampling runs on the GPU (k3_resample_batch = ResampleWaveform).
inline bool MatchSampleRate(Wave *w, float want, bool allow_downsample, bool allow_upsample) {
  if (w->samp_freq == want) return true;
  if ((want < w->samp_freq && !allow_downsample) || (want > w->samp_freq && !allow_upsample)) return false;
  const int32_t ri = (int32_t)w->samp_freq, ro = (int32_t)want; const int64_t n_in = (int64_t)w->samples.size(), n_out = k3_resample_num_samples(ri, ro, n_in);
  if (n_out < 0) return false;
  std::vector<float> out((size_t)n_out);
  if (n_out > 0 && n_in > 0) {
    float *d_in = nullptr, *d_out = nullptr; const int64_t io[2] = {0, n_in}, oo[2] = {0, n_out};
    if (hipMalloc((void **)&d_in, (size_t)n_in * 4) != hipSuccess || hipMalloc((void **)&d_out, (size_t)n_out * 4) != hipSuccess ||
        hipMemcpy(d_in, w->samples.data(), (size_t)n_in * 4, hipMemcpyHostToDevice) != hipSuccess)
      K3H_ERR << "HIP error while resampling";
    if (k3_resample_batch(ri, ro, d_in, io, 1, d_out, oo, nullptr) != 0) K3H_ERR << k3_last_error();
    if (hipMemcpy(out.data(), d_out, (size_t)n_out * 4, hipMemcpyDeviceToHost) != hipSuccess) K3H_ERR << "HIP error while resampling";
    (void)hipFree(d_in); (void)hipFree(d_out);
  }
  w->samples.swap(out); w->samp_freq = want;
  return true;
}
struct FeatOptions {
  k3_feat_opts o; std::string window_type = "povey";
  bool remove_dc = true, round_pow2 = true, snip_edges = true, use_energy = false, raw_energy = true, htk_compat = false, use_log_fbank = true, use_power = true;
  bool allow_downsample = false, allow_upsample = false, debug_mel = false;
  explicit FeatOptions(bool mfcc) {
    memset(&o, 0, sizeof o);
    o.samp_freq = 16000; o.frame_shift_ms = 10; o.frame_length_ms = 25; o.dither = 1.0f; o.preemph_coeff = 0.97f; o.blackman_coeff = 0.42f;
    o.num_bins = 23; o.low_freq = 20; o.high_freq = 0; o.vtln_low = 100; o.vtln_high = -500; o.energy_floor = 0.0f; o.num_ceps = 13; o.cepstral_lifter = 22.0f;
    o.feature_type = mfcc ? 1 : 0; o.vtln_warp = 1.0f; use_energy = mfcc;
  }
  void Register(ParseOptions *po) {
    po->Register("sample-frequency", &o.samp_freq, "Waveform data sample frequency (must match the waveform file, if specified there)");
    po->Register("frame-length", &o.frame_length_ms, "Frame length in milliseconds"); po->Register("frame-shift", &o.frame_shift_ms, "Frame shift in milliseconds");
    po->Register("preemphasis-coefficient", &o.preemph_coeff, "Coefficient for use in signal preemphasis");
    po->Register("remove-dc-offset", &remove_dc, "Subtract mean from waveform on each frame");
    po->Register("dither", &o.dither, "Dithering constant (0.0 means no dither)");
    po->Register("window-type", &window_type, "Type of window (\"hamming\"|\"hanning\"|\"povey\"|\"rectangular\"|\"sine\"|\"blackman\")");
    po->Register("blackman-coeff", &o.blackman_coeff, "Constant coefficient for generalized Blackman window.");
    po->Register("round-to-power-of-two", &round_pow2, "If true, round window size to power of two by zero-padding input to FFT.");
    po->Register("snip-edges", &snip_edges, "If true, end effects will be handled by outputting only frames that completely fit in the file");
    po->Register("allow-downsample", &allow_downsample,
        "If true, allow the input waveform to have a higher frequency than the specified --sample-frequency (and we'll downsample).");
    po->Register("allow-upsample", &allow_upsample,
        "If true, allow the input waveform to have a lower frequency than the specified --sample-frequency (and we'll upsample).");
    po->Register("num-mel-bins", &o.num_bins, "Number of triangular mel-frequency bins"); po->Register("low-freq", &o.low_freq, "Low cutoff frequency for mel bins");
    po->Register("high-freq", &o.high_freq, "High cutoff frequency for mel bins (if <= 0, offset from Nyquist)");
    po->Register("vtln-low", &o.vtln_low, "Low inflection point in piecewise linear VTLN warping function");
    po->Register("vtln-high", &o.vtln_high, "High inflection point in piecewise linear VTLN warping function (if negative, offset from high-mel-freq");
    po->Register("debug-mel", &debug_mel, "(accepted, ignored)");
    po->Register("use-energy", &use_energy, "Add an extra dimension with energy / use energy (not C0)");
    po->Register("energy-floor", &o.energy_floor, "Floor on energy (absolute, not relative)");
    po->Register("raw-energy", &raw_energy, "If true, compute energy before preemphasis and windowing");
    po->Register("htk-compat", &htk_compat, "If true, put energy/C0 last and (mfcc) use a factor of sqrt(2) on C0");
    po->Register("use-log-fbank", &use_log_fbank, "If true, produce log-filterbank, else produce linear.");
    po->Register("use-power", &use_power, "If true, use power, else use magnitude.");
    po->Register("num-ceps", &o.num_ceps, "Number of cepstra in MFCC computation (including C0)");
    po->Register("cepstral-lifter", &o.cepstral_lifter, "Constant that controls scaling of MFCCs");
    po->Register("vtln-warp", &o.vtln_warp, "Vtln warp factor (only applicable if vtln-map not specified)");
  }
  const k3_feat_opts &Finish() {
    static const char *names[] = {"hanning", "sine", "hamming", "povey", "rectangular", "blackman"};
    o.window_type = -1; for (int i = 0; i < 6; i++) if (window_type == names[i]) o.window_type = i;
    if (o.window_type < 0) K3H_ERR << "Invalid window type " << window_type;
    o.remove_dc_offset = remove_dc; o.round_to_power_of_two = round_pow2; o.snip_edges = snip_edges; o.use_energy = use_energy; o.raw_energy = raw_energy;
    o.htk_compat = htk_compat; o.use_log_fbank = use_log_fbank; o.use_power = use_power;
    return o;
  }
};
}  // namespace k3host
