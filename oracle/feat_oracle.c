/*
 * oracle/feat_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float32 arithmetic like the reference's BaseFloat) of the Kaldi
 * fbank / MFCC / CMVN feature path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's shared object; kaldi_amd/ never does.
 *
 * Pinned against (tests/test_oracle_feat.py):
 *   - the reference's own HTK golden vectors src/feat/test_data/test.wav.fbank_htk.{1..4} and
 *     test.wav.fea_htk.{1..6} with the tolerances of feat/feature-fbank-test.cc / feature-mfcc-test.cc
 *   - outputs of the reference binaries compute-fbank-feats / compute-mfcc-feats / apply-cmvn built
 *     from /root/reference by oracle/build_ref.sh (fixtures in tests/golden/, generator committed).
 *
 * Each function cites the reference file:line it restates (paths relative to /root/reference/src).
 * The one deliberate deviation: the complex FFT of length N/2 is an iterative radix-2 Cooley-Tukey
 * in float32 instead of the reference's split-radix butterflies (matrix/srfft.cc:211-352); both are
 * exact DFTs up to float32 round-off (measured max |delta log-mel| vs the reference binary ~2e-6).
 * The real-FFT unpacking step and the packed output layout follow srfft.cc:356-432 literally.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifndef M_PI
#define M_PI 3.1415926535897932384626433832795
#endif
#define M_2PI 6.283185307179586476925286766559005

typedef struct {
  /* FrameExtractionOptions, feat/feature-window.h:35-67 */
  float samp_freq, frame_shift_ms, frame_length_ms, dither, preemph_coeff, blackman_coeff;
  int32_t remove_dc_offset, round_to_power_of_two, snip_edges;
  int32_t window_type; /* 0 hanning 1 sine 2 hamming 3 povey 4 rectangular 5 blackman */
  /* MelBanksOptions, feat/mel-computations.h:43-60 */
  int32_t num_bins;
  float low_freq, high_freq, vtln_low, vtln_high;
  int32_t htk_mode;
  /* FbankOptions feat/feature-fbank.h:44-61 / MfccOptions feat/feature-mfcc.h:40-60 */
  int32_t use_energy;
  float energy_floor;
  int32_t raw_energy, htk_compat, use_log_fbank, use_power;
  int32_t num_ceps;
  float cepstral_lifter;
  int32_t feature_type; /* 0 fbank, 1 mfcc */
  float vtln_warp;      /* argument of Compute(wave, vtln_warp, ...); 1.0 = none */
} k3o_feat_opts;

/* feat/feature-window.h:106-115 */
static int32_t window_shift(const k3o_feat_opts *o) { return (int32_t)(o->samp_freq * 0.001 * o->frame_shift_ms); }
static int32_t window_size(const k3o_feat_opts *o) { return (int32_t)(o->samp_freq * 0.001 * o->frame_length_ms); }
static int32_t padded_window_size(const k3o_feat_opts *o) {
  int32_t w = window_size(o);
  if (!o->round_to_power_of_two) return w;
  int32_t n = 1; while (n < w) n <<= 1; return n;
}

/* feat/feature-window.cc:28-38 */
static int64_t first_sample_of_frame(int32_t frame, const k3o_feat_opts *o) {
  int64_t shift = window_shift(o);
  if (o->snip_edges) return frame * shift;
  int64_t mid = shift * frame + shift / 2;
  return mid - window_size(o) / 2;
}

/* feat/feature-window.cc:40-87 (flush = true, the offline case) */
int32_t k3o_num_frames(int64_t num_samples, const k3o_feat_opts *o) {
  int64_t shift = window_shift(o), len = window_size(o);
  if (o->snip_edges) {
    if (num_samples < len) return 0;
    return (int32_t)(1 + ((num_samples - len) / shift));
  }
  return (int32_t)((num_samples + (shift / 2)) / shift);
}

int32_t k3o_feat_dim(const k3o_feat_opts *o) {
  if (o->feature_type == 1) return o->num_ceps;                 /* feature-mfcc.h Dim() */
  return o->num_bins + (o->use_energy ? 1 : 0);                  /* feature-fbank.h Dim() */
}

/* feat/feature-window.cc:109-135 */
static void make_window(const k3o_feat_opts *o, float *w) {
  int32_t L = window_size(o);
  double a = M_2PI / (L - 1);
  for (int32_t i = 0; i < L; i++) {
    double x = (double)i;
    switch (o->window_type) {
      case 0: w[i] = (float)(0.5 - 0.5 * cos(a * x)); break;
      case 1: w[i] = (float)(sin(0.5 * a * x)); break;
      case 2: w[i] = (float)(0.54 - 0.46 * cos(a * x)); break;
      case 3: w[i] = (float)(pow(0.5 - 0.5 * cos(a * x), 0.85)); break;
      case 4: w[i] = 1.0f; break;
      default: w[i] = (float)(o->blackman_coeff - 0.5 * cos(a * x) + (0.5 - o->blackman_coeff) * cos(2 * a * x));
    }
  }
}

/* feat/mel-computations.h:81-87 */
static float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

static float inv_mel_scale(float m) { return 700.0f * (expf(m / 1127.0f) - 1.0f); }

/* feat/mel-computations.cc:150-210 VtlnWarpFreq, :212-222 VtlnWarpMelFreq */
static float vtln_warp_freq(float vlow, float vhigh, float low, float high, float warp, float freq) {
  if (freq < low || freq > high) return freq;
  float one = 1.0f;
  float l = vlow * (one > warp ? one : warp), h = vhigh * (one < warp ? one : warp);
  float scale = 1.0f / warp, Fl = scale * l, Fh = scale * h;
  float scale_left = (Fl - low) / (l - low), scale_right = (high - Fh) / (high - h);
  if (freq < l) return low + scale_left * (freq - low);
  else if (freq < h) return scale * freq;
  else return high + scale_right * (freq - high);
}
static float vtln_warp_mel(float vlow, float vhigh, float low, float high, float warp, float mel) {
  return mel_scale(vtln_warp_freq(vlow, vhigh, low, high, warp, inv_mel_scale(mel)));
}

typedef struct { int32_t offset, len; float *w; } melbin;

/* feat/mel-computations.cc:33-142 */
static melbin *make_mel_banks(const k3o_feat_opts *o) {
  int32_t nb = o->num_bins, npad = padded_window_size(o), nfft = npad / 2;
  float nyq = 0.5f * o->samp_freq, low = o->low_freq;
  float high = (o->high_freq > 0.0f) ? o->high_freq : nyq + o->high_freq;
  float bin_w = o->samp_freq / npad;
  float mlow = mel_scale(low), mhigh = mel_scale(high);
  float delta = (mhigh - mlow) / (nb + 1);
  float vlow = o->vtln_low, vhigh = o->vtln_high; if (vhigh < 0.0f) vhigh += nyq;
  melbin *b = (melbin *)calloc(nb, sizeof(melbin));
  float *tmp = (float *)malloc(sizeof(float) * nfft);
  for (int32_t bin = 0; bin < nb; bin++) {
    float left = mlow + bin * delta, center = mlow + (bin + 1) * delta, right = mlow + (bin + 2) * delta;
    if (o->vtln_warp != 1.0f) {
      left = vtln_warp_mel(vlow, vhigh, low, high, o->vtln_warp, left);
      center = vtln_warp_mel(vlow, vhigh, low, high, o->vtln_warp, center);
      right = vtln_warp_mel(vlow, vhigh, low, high, o->vtln_warp, right);
    }
    int32_t first = -1, last = -1;
    memset(tmp, 0, sizeof(float) * nfft);
    for (int32_t i = 0; i < nfft; i++) {
      float freq = bin_w * i, mel = mel_scale(freq);
      if (mel > left && mel < right) {
        float wt = (mel <= center) ? (mel - left) / (center - left) : (right - mel) / (right - center);
        tmp[i] = wt;
        if (first == -1) first = i;
        last = i;
      }
    }
    b[bin].offset = first; b[bin].len = last + 1 - first;
    b[bin].w = (float *)malloc(sizeof(float) * b[bin].len);
    memcpy(b[bin].w, tmp + first, sizeof(float) * b[bin].len);
    if (o->htk_mode && bin == 0 && mlow != 0.0f) b[bin].w[0] = 0.0f;
  }
  free(tmp);
  return b;
}

/* The data path (FFT, ProcessWindow, power spectrum, mel, log, DCT) lives in feat_oracle_path.inc and is compiled twice: float32 (the restatement proper) and
 * float64 over the same float32 tables (k3o_compute_features_f64path: the exact value of the reference's formulas, the yardstick of the truth-distance gates). */
#define REAL float
#define FN(x) x
#define RCOS cosf
#define RSIN sinf
#define RLOG logf
#define RPOW powf
#include "feat_oracle_path.inc"
#undef REAL
#undef FN
#undef RCOS
#undef RSIN
#undef RLOG
#undef RPOW
#define REAL double
#define FN(x) x##_f64path
#define RCOS cos
#define RSIN sin
#define RLOG log
#define RPOW pow
#include "feat_oracle_path.inc"
#undef REAL
#undef FN
#undef RCOS
#undef RSIN
#undef RLOG
#undef RPOW

/* transform/cmvn.cc:30-62 AccCmvnStats (double accumulators) then :64-115 ApplyCmvn, per utterance
 * (what `compute-cmvn-stats | apply-cmvn` do for a one-utterance speaker).  In place. */
int32_t k3o_cmvn_offline(float *feats, int32_t T, int32_t dim, int32_t norm_vars) {
  if (T < 1) return -1;
  double *mean = (double *)calloc(dim, sizeof(double)), *var = (double *)calloc(dim, sizeof(double));
  for (int32_t t = 0; t < T; t++) for (int32_t d = 0; d < dim; d++) { float x = feats[(int64_t)t*dim+d]; mean[d] += x * 1.0f; var[d] += x * x * 1.0f; }
  double count = T;
  for (int32_t d = 0; d < dim; d++) {
    if (!norm_vars) { /* Vector<float>::AddVec(float alpha, Vector<double>) kaldi-vector.cc:1044-1052 */
      float alpha = (float)(-1.0 / count); float off = (float)(0.0f + alpha * mean[d]);
      for (int32_t t = 0; t < T; t++) feats[(int64_t)t*dim+d] += off; }
    else { double m = mean[d] / count, v = var[d] / count - m * m; if (v < 1.0e-20) v = 1.0e-20; double sc = 1.0 / sqrt(v); float fo = (float)(-(m * sc)), fs = (float)sc;
      for (int32_t t = 0; t < T; t++) { float x = feats[(int64_t)t*dim+d]; x *= fs; x += fo; feats[(int64_t)t*dim+d] = x; } }
  }
  free(mean); free(var);
  return 0;
}

/* feat/online-feature.cc OnlineCmvn::GetFrame (:441-468) for every frame of one utterance (what online2bin/apply-cmvn-online.cc
 * does per utterance): sliding-window stats in double built by the add-new / subtract-oldest recursion of ComputeStatsForFrame
 * (:361-393; the caching at :272-338 only copies stats), SmoothOnlineCmvnStats (:397-439) with the speaker stats (may be NULL)
 * and the global stats, FakeStatsForSomeDims (transform/cmvn.cc:168-179) for skip_dims, then ApplyCmvn (transform/cmvn.cc:64-115)
 * on the single row.  Stats layout [2 x (dim+1)] doubles: row 0 = sums and count, row 1 = sums of squares.  Returns 0, or -1 on the
 * conditions the reference raises errors for (no global stats, count < 1). */
int32_t k3o_cmvn_online(const float *in, float *out, int32_t T, int32_t dim, int32_t cmn_window, int32_t speaker_frames, int32_t global_frames,
                        int32_t norm_means, int32_t norm_vars, const double *global_stats, const double *speaker_stats,
                        const int32_t *skip_dims, int32_t n_skip) {
  if (!global_stats || dim < 1 || speaker_frames > cmn_window || global_frames > speaker_frames) return -1;
  if (norm_vars && !norm_means) return -1;
  const int32_t C = dim + 1;
  double *st = (double *)calloc(2 * C, sizeof(double)), *sm = (double *)calloc(2 * C, sizeof(double));
  int32_t rc = 0;
  for (int32_t t = 0; t < T && rc == 0; t++) {
    for (int32_t d = 0; d < dim; d++) { double x = in[(int64_t)t * dim + d]; st[d] += 1.0 * x; if (norm_vars) st[C + d] += 1.0 * x * x; }
    st[dim] += 1.0;
    if (t - cmn_window >= 0) {
      const float *old = in + (int64_t)(t - cmn_window) * dim;
      for (int32_t d = 0; d < dim; d++) { double x = old[d]; st[d] += -1.0 * x; if (norm_vars) st[C + d] += -1.0 * x * x; }
      st[dim] -= 1.0;
    }
    for (int32_t i = 0; i < 2 * C; i++) sm[i] = st[i];
    const int32_t rows = norm_vars ? 2 : 1;          /* the one-row shortcut of SmoothOnlineCmvnStats when variance is not needed */
    double cur = sm[dim];
    if (cur < cmn_window) {
      if (speaker_stats) {
        double from_spk = cmn_window - cur, spk_count = speaker_stats[dim];
        if (from_spk > speaker_frames) from_spk = speaker_frames;
        if (from_spk > spk_count) from_spk = spk_count;
        if (from_spk > 0.0) { double a = from_spk / spk_count; for (int32_t r = 0; r < rows; r++) for (int32_t i = 0; i < C; i++) sm[r * C + i] += a * speaker_stats[r * C + i]; }
        cur = sm[dim];
      }
      if (cur < cmn_window) {
        double from_glob = cmn_window - cur, glob_count = global_stats[dim];
        if (!(glob_count > 0.0)) { rc = -1; break; }
        if (from_glob > global_frames) from_glob = global_frames;
        if (from_glob > 0.0) { double a = from_glob / glob_count; for (int32_t r = 0; r < rows; r++) for (int32_t i = 0; i < C; i++) sm[r * C + i] += a * global_stats[r * C + i]; }
      }
    }
    for (int32_t k = 0; k < n_skip; k++) { sm[skip_dims[k]] = 0.0; sm[C + skip_dims[k]] = sm[dim]; }
    const float *x = in + (int64_t)t * dim; float *y = out + (int64_t)t * dim;
    if (!norm_means) { for (int32_t d = 0; d < dim; d++) y[d] = x[d]; continue; }
    const double count = sm[dim];
    if (count < 1.0) { rc = -1; break; }
    for (int32_t d = 0; d < dim; d++) {
      if (!norm_vars) { float alpha = (float)(-1.0 / count); float off = (float)(0.0f + alpha * sm[d]); y[d] = x[d] + 1.0f * off; }
      else {
        double m = sm[d] / count, v = sm[C + d] / count - m * m; if (v < 1.0e-20) v = 1.0e-20;
        double sc = 1.0 / sqrt(v); float fo = (float)(-(m * sc)), fs = (float)sc;
        float z = x[d]; z *= fs; z += 1.0f * fo; y[d] = z;
      }
    }
  }
  free(st); free(sm);
  return rc;
}
