// option registration shared by the CLI programs: the flag names of FrameExtractionOptions (feat/feature-window.h:69-101),
// MelBanksOptions (feat/mel-computations.h:60-80), FbankOptions (feat/feature-fbank.h:62-79), MfccOptions (feat/feature-mfcc.h:62-79)
#pragma once
#include "k3_host.h"
#include "../../include/k3hip.h"
namespace k3host {
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
    po->Register("preemphasis-coefficient", &o.preemph_coeff, "Coefficient for use in signal preemphasis"); po->Register("remove-dc-offset", &remove_dc, "Subtract mean from waveform on each frame");
    po->Register("dither", &o.dither, "Dithering constant (0.0 means no dither)"); po->Register("window-type", &window_type, "Type of window (\"hamming\"|\"hanning\"|\"povey\"|\"rectangular\"|\"sine\"|\"blackman\")");
    po->Register("blackman-coeff", &o.blackman_coeff, "Constant coefficient for generalized Blackman window."); po->Register("round-to-power-of-two", &round_pow2, "If true, round window size to power of two by zero-padding input to FFT.");
    po->Register("snip-edges", &snip_edges, "If true, end effects will be handled by outputting only frames that completely fit in the file");
    po->Register("allow-downsample", &allow_downsample, "(accepted, resampling is not implemented)"); po->Register("allow-upsample", &allow_upsample, "(accepted, resampling is not implemented)");
    po->Register("num-mel-bins", &o.num_bins, "Number of triangular mel-frequency bins"); po->Register("low-freq", &o.low_freq, "Low cutoff frequency for mel bins");
    po->Register("high-freq", &o.high_freq, "High cutoff frequency for mel bins (if <= 0, offset from Nyquist)"); po->Register("vtln-low", &o.vtln_low, "Low inflection point in piecewise linear VTLN warping function");
    po->Register("vtln-high", &o.vtln_high, "High inflection point in piecewise linear VTLN warping function (if negative, offset from high-mel-freq"); po->Register("debug-mel", &debug_mel, "(accepted, ignored)");
    po->Register("use-energy", &use_energy, "Add an extra dimension with energy / use energy (not C0)"); po->Register("energy-floor", &o.energy_floor, "Floor on energy (absolute, not relative)");
    po->Register("raw-energy", &raw_energy, "If true, compute energy before preemphasis and windowing"); po->Register("htk-compat", &htk_compat, "If true, put energy/C0 last and (mfcc) use a factor of sqrt(2) on C0");
    po->Register("use-log-fbank", &use_log_fbank, "If true, produce log-filterbank, else produce linear."); po->Register("use-power", &use_power, "If true, use power, else use magnitude.");
    po->Register("num-ceps", &o.num_ceps, "Number of cepstra in MFCC computation (including C0)"); po->Register("cepstral-lifter", &o.cepstral_lifter, "Constant that controls scaling of MFCCs");
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
