// k3_feat.hip -- batched fbank / MFCC / CMVN for gfx950 (MI355X), hand-written HIP.
//
// One fused kernel per batch: window extraction (copy/reflect, zero-pad) -> DC removal ->
// raw log-energy -> pre-emphasis -> window multiply -> 512-point real FFT -> power spectrum ->
// mel filterbank -> floor/log [-> DCT -> lifter] -> store.  Nothing but the waveform is read from
// HBM and nothing but the feature rows is written: algorithmic traffic = 160 new samples * 4 B in
// + dim * 4 B out per frame (HBM-bound stage, intensity ~20-30 flop/B).
//
// CDNA4 mapping (wave64): a wavefront processes 4 frames at once, 16 lanes per frame.  The 512-point
// real FFT is a 256-point complex FFT of z[n] = x[2n] + i x[2n+1] (the packing the reference's
// SplitRadixRealFft uses, matrix/srfft.cc:356-432) factored 16 x 16: each lane does a radix-16 FFT in
// registers, the 16x16 transpose between the two passes goes through a padded (stride 17) LDS tile
// (conflict-free ds_write_b64 / ds_read_b64), then the real-FFT unpacking + |.|^2 produce the 257
// power bins in LDS and the sparse mel dot-products run 16 bins at a time.  All tables (window, twiddles,
// mel weights, DCT) are staged in LDS once per workgroup, which then loops over 64 frames.
//
// Reference semantics restated (paths relative to the reference's src/):
//   ExtractWindow feat/feature-window.cc:166-224, ProcessWindow :137-160, Preemphasize :101-107,
//   FeatureWindowFunction :109-135, ComputePowerSpectrum feat/feature-functions.cc:30-52,
//   MelBanks feat/mel-computations.cc:33-142,226-251, FbankComputer::Compute feat/feature-fbank.cc:72-123,
//   MfccComputer::Compute feat/feature-mfcc.cc:28-80, ComputeDctMatrix matrix/matrix-functions.cc:592-608,
//   ComputeLifterCoeffs feat/mel-computations.cc:253-259, AccCmvnStats/ApplyCmvn transform/cmvn.cc:30-115.
// The FFT butterfly order differs from split-radix (round-off level only; parity bound 1e-4 on log-mel).
#include "k3_common.h"
#include <mutex>
#include <cmath>
#include <cfloat>
#include <vector>
#include <cstring>

namespace {

// The kernel is instantiated for padded window sizes N = 32 R, R = 8 / 16 / 32 (256, 512, 1024 samples: 8 kHz, 16 kHz and 32..44.1 kHz speech at
// 25 ms): the complex FFT of length NC = 16 R is a Cooley-Tukey R x 16 (a lane holds R points, 16 lanes hold a frame).
//
// Arithmetic (round 4): the data path from the samples to the logarithm is FLOAT64; the tables (window, mel weights, DCT, lifter, pre-emphasis
// coefficient) are the reference's own float32 values, made on the host the way its code makes them.  Why: on the benchmark audio the reference's
// float32 binary is up to 1.07e-4 away from the exact value of its own formulas (low mel bins: pre-emphasis leaves ~1e-3 of the mean power there and
// every float32 butterfly adds noise relative to the frame's rms), and so was the float32 version of this kernel, with independent errors: the two
// differed by up to 1.35e-4.  In float64 this kernel sits ~1e-6 (the output's own float32 rounding) from the exact value, so |kernel - reference| is
// the reference's own rounding error and nothing else (tests/test_feat_gpu.py, bench.py e2e_parity: truth64 gates).  The stage is ~8 GFLOP per
// 512 x 10 s batch: float64 costs nothing measurable against the 2.6 TFLOP network behind it.
constexpr int kFramesPerIter = 16;    // 4 waves x 4 frames
constexpr int kThreads = 256;
constexpr int kTpad = 17;             // transpose tile row stride (doubles)
// R = 16: 2176 B per frame: transpose tile (real parts, then imaginary parts) / P
__host__ __device__ constexpr int frame_buf_bytes(int R) {
  return R * kTpad * 8;
}
// twiddles NC (double2), twiddles N (N/4 + 1, double2), window (float)
__host__ __device__ constexpr int lds_fixed_bytes(int R) {
  return 16 * R * 16 + ((8 * R + 1) * 16) + 32 * R * 4;
}

struct FeatParams {
  int win_len, win_shift, snip_edges, remove_dc, use_energy, raw_energy, htk_compat, use_log, use_power,
      htk_mode, feature_type, num_bins, num_ceps, dim, has_energy_floor, has_lifter, total_w;
  float preemph, log_energy_floor;
  float dither; unsigned dither_seed;   // Dither (feat/feature-window.cc:90-98): x[i] += RandGauss() * dither, independently per frame
  const float *window;      // [win_len]
  const double2 *tw256;     // [NC] exp(-2 pi i m / NC)
  const double2 *tw512;     // [N/4 + 1] exp(-2 pi i k / N), k = 0..N/4
  const int *bin_meta;      // [3 * num_bins]: first fft bin, length, offset into bin_w
  const float *bin_w;       // [total_w]
  const float *dct;         // [num_ceps x num_bins] (mfcc)
  const float *lifter;      // [num_ceps] (mfcc)
};

typedef double2 cd;
__device__ __forceinline__ cd mk(double x, double y) { cd r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ cd cadd(cd a, cd b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cd csub(cd a, cd b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cd cmul(cd a, cd b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// forward 4-point DFT in place: (a,b,c,d) -> (X0,X1,X2,X3)
__device__ __forceinline__ void fft4(cd &a, cd &b, cd &c, cd &d) {
  cd s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = csub(b, d);
  a = cadd(s0, s2);
  c = csub(s0, s2);
  b = mk(s1.x + s3.y, s1.y - s3.x);   // s1 - i s3
  d = mk(s1.x - s3.y, s1.y + s3.x);   // s1 + i s3
}

// forward 16-point DFT in registers; on return X[k] sits at v[4*(k&3) + (k>>2)]
__device__ __forceinline__ void fft16(cd (&v)[16]) {
  constexpr double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178, r2 = 0.70710678118654752440;
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) fft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
  // v[4*k1 + n2] *= exp(-2 pi i n2 k1 / 16)
  v[5] = cmul(v[5], mk(c1, -s1));      // m = 1
  v[6] = cmul(v[6], mk(r2, -r2));      // m = 2
  v[7] = cmul(v[7], mk(s1, -c1));      // m = 3
  v[9] = cmul(v[9], mk(r2, -r2));      // m = 2
  v[10] = mk(v[10].y, -v[10].x);       // m = 4: * (-i)
  v[11] = cmul(v[11], mk(-r2, -r2));   // m = 6
  v[13] = cmul(v[13], mk(s1, -c1));    // m = 3
  v[14] = cmul(v[14], mk(-r2, -r2));   // m = 6
  v[15] = cmul(v[15], mk(-c1, s1));    // m = 9
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) fft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
__host__ __device__ constexpr int fft16_slot(int k) { return 4 * (k & 3) + (k >> 2); }

// forward R-point DFT in registers, R = 8 / 16 / 32; X[k] sits at v[FftR<R>::slot(k)]
template <int R> struct FftR;
template <> struct FftR<16> {
  static __device__ __forceinline__ void run(cd (&v)[16]) { fft16(v); }
  static __host__ __device__ constexpr int slot(int k) { return fft16_slot(k); }
};
template <> struct FftR<8> {      // two 4-point DFTs (even / odd samples) + one radix-2 stage
  static __device__ __forceinline__ void run(cd (&v)[8]) {
    constexpr double r2 = 0.70710678118654752440;
    fft4(v[0], v[2], v[4], v[6]); fft4(v[1], v[3], v[5], v[7]);
    const cd o1 = cmul(v[3], mk(r2, -r2)), o2 = mk(v[5].y, -v[5].x), o3 = cmul(v[7], mk(-r2, -r2)), o0 = v[1];
    const cd e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0); v[1] = cadd(e1, o1); v[5] = csub(e1, o1); v[2] = cadd(e2, o2); v[6] = csub(e2, o2); v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
  }
  static __host__ __device__ constexpr int slot(int k) { return k; }
};
template <> struct FftR<32> {     // two 16-point DFTs (even / odd samples) + one radix-2 stage
  static __device__ __forceinline__ void run(cd (&v)[32]) {
    constexpr double c[16] = {1.0, 0.98078528040323044913, 0.92387953251128673848, 0.83146961230254523708, 0.70710678118654752440, 0.55557023301960222474, 0.38268343236508978178,
                              0.19509032201612826785, 0.0, -0.19509032201612826785, -0.38268343236508978178, -0.55557023301960222474, -0.70710678118654752440,
                                  -0.83146961230254523708,
                              -0.92387953251128673848, -0.98078528040323044913};
    constexpr double sn[16] = {0.0, 0.19509032201612826785, 0.38268343236508978178, 0.55557023301960222474, 0.70710678118654752440, 0.83146961230254523708, 0.92387953251128673848,
                               0.98078528040323044913, 1.0, 0.98078528040323044913, 0.92387953251128673848, 0.83146961230254523708, 0.70710678118654752440, 0.55557023301960222474,
                               0.38268343236508978178, 0.19509032201612826785};
    cd e[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    fft16(e); fft16(o);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const cd ek = e[fft16_slot(k)], ok = k == 0 ? o[fft16_slot(0)] : cmul(o[fft16_slot(k)], mk(c[k], -sn[k]));
      v[k] = cadd(ek, ok); v[k + 16] = csub(ek, ok);
    }
  }
  static __host__ __device__ constexpr int slot(int k) { return k; }
};

// A frame belongs to the 16 lanes of one DPP row: cross-lane traffic inside a frame is DPP row rotations (VALU, no LDS crossbar trip).
// row_ror:n -- lane l of a row receives lane (l - n) mod 16.
template <int N> __device__ __forceinline__ double row_ror(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// every lane of the row gets the row's sum (the four partial orders differ by lane: callers use lane 0's, or exact sums)
__device__ __forceinline__ double group_sum16(double x) {
  x += row_ror<8>(x); x += row_ror<4>(x); x += row_ror<2>(x); x += row_ror<1>(x);
  return x;
}
// The 4 frames of a wavefront own their LDS buffers: the phases of a frame are ordered by the wave's own program order (LDS instructions of a
// wave execute in order); only the compiler has to be told not to move LDS accesses across the phase boundary.  No workgroup barrier in the frame loop.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// TS = float (CuVector<BaseFloat> waves, the reference's interface) or int16_t (PCM16 as it sits in a wav file: 2 of the 480 B / frame of SURVEY 8d)
template <typename TS, int R>
__global__ __launch_bounds__(kThreads) void k3_feat_kernel(FeatParams p, const TS *__restrict__ waves,
                                                          const int64_t *__restrict__ wave_off,
                                                          const int64_t *__restrict__ frame_off, int num_utts,
                                                          int64_t total_frames, float *__restrict__ feats, int64_t ld,
                                                          int frames_per_block) {
  constexpr int kNc = 16 * R, kNfft = 32 * R, kFrameBufBytes = frame_buf_bytes(R), kTwN = (8 * R + 1) * 16, kFixed = lds_fixed_bytes(R);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS carve (all offsets multiples of 16)
  cd *s_tw256 = reinterpret_cast<cd *>(smem);                         // NC * 16 B (R = 16: 4096)
  cd *s_tw512 = reinterpret_cast<cd *>(smem + kNc * 16);              // (N/4 + 1) * 16 (2064)
  float *s_window = reinterpret_cast<float *>(smem + kNc * 16 + kTwN); // N * 4 B (2048)
  int *s_meta = reinterpret_cast<int *>(smem + kFixed);                // 3*num_bins ints, padded to 16
  const int meta_bytes = ((3 * p.num_bins * 4 + 15) / 16) * 16;
  float *s_binw = reinterpret_cast<float *>(smem + kFixed + meta_bytes);
  const int binw_bytes = ((p.total_w * 4 + 15) / 16) * 16;
  char *s_frames = smem + kFixed + meta_bytes + binw_bytes;           // 16 frame buffers

  const int tid = threadIdx.x;
  for (int i = tid; i < kNc; i += kThreads) s_tw256[i] = p.tw256[i];
  for (int i = tid; i < kNc / 2 + 1; i += kThreads) s_tw512[i] = p.tw512[i];
  for (int i = tid; i < kNfft; i += kThreads) s_window[i] = (i < p.win_len) ? p.window[i] : 0.0f;
  for (int i = tid; i < 3 * p.num_bins; i += kThreads) s_meta[i] = p.bin_meta[i];
  for (int i = tid; i < p.total_w; i += kThreads) s_binw[i] = p.bin_w[i];
  __syncthreads();      // the only workgroup barrier: the tables

  const int lane = tid & 63, wave = tid >> 6, grp = lane >> 4, l = lane & 15;
  const int fslot = wave * 4 + grp;
  double *T = reinterpret_cast<double *>(s_frames + fslot * kFrameBufBytes);      // R x 17 transpose tile / X / P
  const int L = p.win_len;
  const double eps = (double)FLT_EPSILON;

  const int64_t block_first = (int64_t)blockIdx.x * frames_per_block;
  for (int it = 0; it < frames_per_block; it += kFramesPerIter) {
    const int64_t g = block_first + it + fslot;
    const bool valid = g < total_frames;
    double x0[R], x1[R];
    int64_t n = 0, start = 0;
    const TS *base = waves;
    if (valid) {
      // utterance lookup: largest u with frame_off[u] <= g
      int lo = 0, hi = num_utts;   // invariant frame_off[lo] <= g < frame_off[hi]
      while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (frame_off[mid] <= g) lo = mid; else hi = mid; }
      const int64_t f = g - frame_off[lo];
      const int64_t w0 = wave_off[lo];
      n = wave_off[lo + 1] - w0;
      base = waves + w0;
      start = p.snip_edges ? f * p.win_shift : f * p.win_shift + p.win_shift / 2 - L / 2;   // FirstSampleOfFrame
    }
    const bool interior = valid && start >= 0 && start + L <= n;
#pragma unroll
    for (int j = 0; j < R; j++) {
      const int s = 2 * (l + 16 * j);
      TS a = 0, b = 0;
      if (interior) {
        if (s < L) a = base[start + s];
        if (s + 1 < L) b = base[start + s + 1];
      } else if (valid) {
        if (s < L) { int64_t si = start + s; while (si < 0 || si >= n) si = (si < 0) ? -si - 1 : 2 * n - 1 - si; a = base[si]; }
        if (s + 1 < L) { int64_t si = start + s + 1; while (si < 0 || si >= n) si = (si < 0) ? -si - 1 : 2 * n - 1 - si; b = base[si]; }
      }
      x0[j] = (double)a; x1[j] = (double)b;
    }
    // ---- ProcessWindow ----
    if (p.dither != 0.0f && valid) {     // counter-based generator (frame, sample pair, seed) -> Box-Muller pair; not the reference's rand() stream
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int s = 2 * (l + 16 * j);
        unsigned long long z = ((unsigned long long)g * 1024ull + (unsigned)(s >> 1)) * 0x9E3779B97F4A7C15ull + ((unsigned long long)p.dither_seed << 32 | 0x7F4A7C15u);
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;          // splitmix64 finaliser
        const float u1 = ((unsigned)(z >> 40) + 1.0f) * (1.0f / 16777217.0f), u2 = (unsigned)((z >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * logf(u1)), th = 6.283185307179586f * u2;
        if (s < L) x0[j] += (double)(p.dither * r * cosf(th));
        if (s + 1 < L) x1[j] += (double)(p.dither * r * sinf(th));
      }
    }
    if (p.remove_dc) {
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < R; j++) sum += x0[j] + x1[j];
      sum = group_sum16(sum);
      const double m = -sum / (double)L;
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int s = 2 * (l + 16 * j);
        if (s < L) x0[j] += m;
        if (s + 1 < L) x1[j] += m;
      }
    }
    double log_energy = 0.0;
    if (p.use_energy && p.raw_energy) {
      double e = 0.0;
#pragma unroll
      for (int j = 0; j < R; j++) e += x0[j] * x0[j] + x1[j] * x1[j];
      e = group_sum16(e);
      log_energy = log(fmax(e, eps));
    }
    cd v[R];
    {
      const double c = (double)p.preemph;      // the reference's float32 coefficient
      double prev_wrap = 0.0;     // lane 15's x1[j-1], as rotated into lane 0
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int s = 2 * (l + 16 * j);
        const double rot = row_ror<1>(x1[j]);       // lane l: x1[j] of lane l - 1; lane 0: lane 15's
        const double prev0 = (l > 0) ? rot : ((j > 0) ? prev_wrap : x0[0]);      // sample s - 1 (Preemphasize: the first sample is its own predecessor)
        prev_wrap = rot;
        double y0 = x0[j], y1 = x1[j];
        if (c != 0.0) {
          if (s < L) y0 = x0[j] - c * prev0;
          if (s + 1 < L) y1 = x1[j] - c * x0[j];
        }
        const float2 w = *reinterpret_cast<const float2 *>(&s_window[s]);
        v[j] = mk(y0 * (double)w.x, y1 * (double)w.y);
      }
    }
    if (p.use_energy && !p.raw_energy) {
      double e = 0.0;
#pragma unroll
      for (int j = 0; j < R; j++) e += v[j].x * v[j].x + v[j].y * v[j].y;
      e = group_sum16(e);
      log_energy = log(fmax(e, eps));
    }
    // ---- NC-point complex FFT, NC = R x 16.  Pass 1: a lane's R points (n = l + 16 j) -> Y[l][k2]; twiddle W_NC^(l k2); transpose;
    //      pass 2: for every k2 a 16-point DFT over l -> X[k2 + R k1].  The tile holds doubles: real parts go through it first, then the imaginary parts ----
    FftR<R>::run(v);
#pragma unroll
    for (int k2 = 1; k2 < R; k2++) v[FftR<R>::slot(k2)] = cmul(v[FftR<R>::slot(k2)], s_tw256[(l * k2) & (kNc - 1)]);
    constexpr int kPer = (R + 15) / 16;          // k2 values a lane transforms in pass 2 (R = 8: lanes 8..15 idle)
    cd u[kPer][16];
#pragma unroll
    for (int k2 = 0; k2 < R; k2++) T[k2 * kTpad + l] = v[FftR<R>::slot(k2)].x;
    wave_sync();
#pragma unroll
    for (int q = 0; q < kPer; q++) { const int k2 = l + 16 * q; if (k2 < R) {
#pragma unroll
      for (int i = 0; i < 16; i++) u[q][i].x = T[k2 * kTpad + i]; } }
    wave_sync();
#pragma unroll
    for (int k2 = 0; k2 < R; k2++) T[k2 * kTpad + l] = v[FftR<R>::slot(k2)].y;
    wave_sync();
#pragma unroll
    for (int q = 0; q < kPer; q++) { const int k2 = l + 16 * q; if (k2 < R) {
#pragma unroll
      for (int i = 0; i < 16; i++) u[q][i].y = T[k2 * kTpad + i];
      fft16(u[q]); } }   // now X[k2 + R*k1] = u[q][slot(k1)]
    wave_sync();
    // ---- X in natural order through the tile (real parts, then imaginary parts); real-FFT unpacking (srfft.cc:372-405) + power spectrum ----
    cd Bk[R / 2], Bm[R / 2], B128;
#pragma unroll
    for (int q = 0; q < kPer; q++) { const int k2 = l + 16 * q; if (k2 < R) {
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) T[k2 + R * k1] = u[q][fft16_slot(k1)].x; } }
    wave_sync();
#pragma unroll
    for (int i = 0; i < R / 2; i++) { const int k = l + 16 * i; Bk[i].x = T[k]; Bm[i].x = T[(kNc - k) & (kNc - 1)]; }
    B128.x = T[kNc / 2];
    wave_sync();
#pragma unroll
    for (int q = 0; q < kPer; q++) { const int k2 = l + 16 * q; if (k2 < R) {
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) T[k2 + R * k1] = u[q][fft16_slot(k1)].y; } }
    wave_sync();
#pragma unroll
    for (int i = 0; i < R / 2; i++) { const int k = l + 16 * i; Bk[i].y = T[k]; Bm[i].y = T[(kNc - k) & (kNc - 1)]; }
    B128.y = T[kNc / 2];
    wave_sync();
    double *P = T;
#pragma unroll
    for (int i = 0; i < R / 2; i++) {
      const int k = l + 16 * i;
      if (k == 0) {
        const double a0 = Bk[0].x + Bk[0].y, an = Bk[0].x - Bk[0].y;
        double p0 = a0 * a0, pn = an * an;
        if (!p.use_power) { p0 = sqrt(p0); pn = sqrt(pn); }
        P[0] = p0; P[kNc] = pn;
      } else {
        const cd w = s_tw512[k];
        const double Cr = 0.5 * (Bk[i].x + Bm[i].x), Ci = 0.5 * (Bk[i].y - Bm[i].y);
        const double Dr = 0.5 * (Bk[i].y + Bm[i].y), Di = -0.5 * (Bk[i].x - Bm[i].x);
        const double Ar = Cr + (Dr * w.x - Di * w.y), Ai = Ci + (Dr * w.y + Di * w.x);
        const double Er = Cr + (Dr * (-w.x) + Di * w.y), Ei = -Ci + (Dr * w.y + Di * w.x);
        double pk = Ar * Ar + Ai * Ai, pm = Er * Er + Ei * Ei;
        if (!p.use_power) { pk = sqrt(pk); pm = sqrt(pm); }
        P[k] = pk; P[kNc - k] = pm;
      }
    }
    if (l == 0) {  // k = N/4 (kdash == k)
      const cd w = s_tw512[kNc / 2];
      const double Ar = B128.x + B128.y * w.x, Ai = B128.y * w.y;
      double pk = Ar * Ar + Ai * Ai;
      if (!p.use_power) pk = sqrt(pk);
      P[kNc / 2] = pk;
    }
    wave_sync();
    // ---- mel filterbank: MelBanks::Compute ----
    float *row = feats + g * ld;
    const int mel_off = (p.feature_type == 0 && p.use_energy && !p.htk_compat) ? 1 : 0;
    double logmel[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int b = l + 16 * r;
      double e = 0.0;
      if (r * 16 < p.num_bins && b < p.num_bins) {
        const int first = s_meta[3 * b], len = s_meta[3 * b + 1], wo = s_meta[3 * b + 2];
        for (int i = 0; i < len; i++) e += (double)s_binw[wo + i] * P[first + i];
        if (p.htk_mode && e < 1.0) e = 1.0;
        if (p.feature_type == 1 || p.use_log) e = log(fmax(e, eps));
      }
      logmel[r] = e;
    }
    if (p.feature_type == 0) {
      if (valid) {
#pragma unroll
        for (int r = 0; r < 8; r++) { const int b = l + 16 * r; if (b < p.num_bins) row[mel_off + b] = (float)logmel[r]; }
        if (p.use_energy && l == 0) {
          float e = (float)log_energy;
          if (p.has_energy_floor && e < p.log_energy_floor) e = p.log_energy_floor;
          row[p.htk_compat ? p.num_bins : 0] = e;
        }
      }
    } else {
      // DCT (MfccComputer::Compute feature-mfcc.cc:61-80): a lane holds the log-mel values of its bins; one row sum per cepstrum
      for (int c = 0; c < p.num_ceps; c++) {
        const float *drow = p.dct + c * p.num_bins;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 8; r++) { const int b = l + 16 * r; if (b < p.num_bins) acc += (double)drow[b] * logmel[r]; }
        acc = group_sum16(acc);
        if (valid && l == (c & 15)) {
          if (p.has_lifter) acc *= (double)p.lifter[c];
          float out = (float)acc;
          if (c == 0 && p.use_energy) {
            out = (float)log_energy;
            if (p.has_energy_floor && out < p.log_energy_floor) out = p.log_energy_floor;
          }
          int oc = c;
          if (p.htk_compat) {
            if (c == 0) { oc = p.num_ceps - 1; if (!p.use_energy) out = (float)(acc * 1.41421356237309504880); }
            else oc = c - 1;
          }
          row[oc] = out;
        }
      }
    }
    wave_sync();   // P is overwritten by the next iteration's transpose tile
  }
}

// ------------------------------------------------------------------ CMVN ----------------------
// The reference rounds every product before it is added (it is compiled without FMA).  __fmul_rn / __dmul_rn are plain
// operators in this HIP and get contracted with a following add; an empty asm makes the rounded product opaque to the optimiser.
__device__ __forceinline__ float rounded(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double rounded(double x) { asm volatile("" : "+v"(x)); return x; }
// One workgroup per (utterance, 8-column group): fp64 sums of x and x^2 over time (AccCmvnStats),
// then x*scale+offset in place (ApplyCmvn).  Rows of an utterance are contiguous, so a wave reads
// full rows (coalesced) and each lane owns column (lane % dim_tile).
__global__ __launch_bounds__(256) void k3_cmvn_kernel(float *__restrict__ feats, int64_t ld, int dim,
                                                      const int64_t *__restrict__ frame_off, int norm_vars,
                                                      double *__restrict__ stats) {
#pragma clang fp contract(off)      // the reference rounds the product and the sum separately (MulColsVec, then AddVecToRows)
  __shared__ double s_sum[256], s_sq[256];
  __shared__ float s_scale[64], s_offset[64];
  const int u = blockIdx.x;
  const int col0 = blockIdx.y * 64;
  const int64_t r0 = frame_off[u], r1 = frame_off[u + 1];
  const int64_t T = r1 - r0;
  const int c = threadIdx.x & 63, rlane = threadIdx.x >> 6;   // 4 row-lanes x 64 columns
  const int col = col0 + c;
  double sum = 0.0, sq = 0.0;
  if (col < dim)
    for (int64_t t = r0 + rlane; t < r1; t += 4) {
      const float x = feats[t * ld + col];
      sum += (double)x; sq += (double)(x * x);   // reference: float product, double accumulate (cmvn.cc:46-47)
    }
  s_sum[threadIdx.x] = sum; s_sq[threadIdx.x] = sq;
  __syncthreads();
  if (rlane == 0 && col < dim) {
    const double m = s_sum[c] + s_sum[64 + c] + s_sum[128 + c] + s_sum[192 + c];
    const double v = s_sq[c] + s_sq[64 + c] + s_sq[128 + c] + s_sq[192 + c];
    const double count = (double)T;
    if (stats) {
      double *st = stats + (int64_t)u * 2 * (dim + 1);
      st[col] = m; st[dim + 1 + col] = v;
      if (col == 0) { st[dim] = count; st[2 * dim + 1] = 0.0; }
    }
    if (!norm_vars) {
      const float alpha = (float)(-1.0 / count);            // Vector<float>::AddVec(float, Vector<double>)
      s_scale[c] = 1.0f; s_offset[c] = (float)(0.0f + alpha * m);
    } else {
      const double mean = m / count;
      double var = v / count - mean * mean;
      if (var < 1.0e-20) var = 1.0e-20;
      const double scale = 1.0 / sqrt(var);
      s_scale[c] = (float)scale; s_offset[c] = (float)(-(mean * scale));
    }
  }
  __syncthreads();
  if (col < dim && T > 0) {
    const float sc = s_scale[c], of = s_offset[c];
    for (int64_t t = r0 + rlane; t < r1; t += 4) {
      float x = feats[t * ld + col];
      if (norm_vars) x = rounded(x * sc);        // MulColsVec then AddVecToRows: two roundings
      feats[t * ld + col] = x + of;
    }
  }
}


// ------------------------------------------------------------------ online CMVN ---------------
// OnlineCmvn::GetFrame for every frame (feat/online-feature.cc:361-468): one lane per (utterance, column) walks the frames in
// order, carrying the window sums in fp64 exactly as ComputeStatsForFrame does (add the new frame, then subtract the one that
// leaves the window), smooths them with the speaker / global stats (SmoothOnlineCmvnStats) and applies ApplyCmvn to the row.
// The recursion is sequential in t by definition (its rounding depends on the order); rows are read coalesced across the
// columns of a wave, 8 frames ahead.  No fp contraction anywhere: the reference is compiled without FMA.
struct OnlineCmvnParams {
  const float *in; float *out; long long ld_in, ld_out; int dim; const long long *frame_off;
  int cmn_window, speaker_frames, global_frames, norm_means, norm_vars;
  const double *global_stats, *speaker_stats;      // [2 x (dim+1)], [U x 2 x (dim+1)] or null
  unsigned long long skip[4];                      // bit d set: FakeStatsForSomeDims for column d
  int *err;                                        // set to 1 where the reference raises (count < 1, global count <= 0)
  // (streaming, many streams in one launch) per utterance: its own buffers, length, first new row and carry; the fields above that they replace are unused
  const k3::CmvnSeg *segs;
  // resume (streaming): rows [0, t_begin[u]) of utterance u are history -- read for the window, not written; carry [U][dim][3] = the
  const long long *t_begin;
  double *carry;
                                                   // window's (sum, sum of squares, count) after the last row, read when t_begin[u] > 0 and written back.  Both null: whole utterances
};

__global__ __launch_bounds__(64) void k3_cmvn_online_kernel(OnlineCmvnParams p) {
#pragma clang fp contract(off)      // __fmul_rn & co are plain operators in this HIP: keep the compiler from fusing them into fma
  const int u = blockIdx.x, d = blockIdx.y * 64 + threadIdx.x;
  if (d >= p.dim) return;
  const long long r0 = p.segs ? 0 : p.frame_off[u], T = p.segs ? p.segs[u].rows : p.frame_off[u + 1] - r0;
  const int C = p.dim + 1, W = p.cmn_window;
  const float *x = p.segs ? p.segs[u].in + d : p.in + r0 * p.ld_in + d; float *y = p.segs ? p.segs[u].out + d : p.out + r0 * p.ld_out + d;
  const double g_m = p.global_stats[d], g_v = p.global_stats[C + d], g_n = p.global_stats[p.dim];
  const double *sp = p.speaker_stats ? p.speaker_stats + (long long)u * 2 * C : nullptr;
  const double s_m = sp ? sp[d] : 0.0, s_v = sp ? sp[C + d] : 0.0, s_n = sp ? sp[p.dim] : 0.0;
  const bool skip = d < 256 && ((p.skip[d >> 6] >> (d & 63)) & 1ull);
  double sum = 0.0, sq = 0.0, n = 0.0;
  const long long tb = p.segs ? p.segs[u].t_begin : p.t_begin ? p.t_begin[u] : 0;
  double *cy = p.segs ? p.segs[u].carry + (long long)d * 3 : p.carry ? p.carry + ((long long)u * p.dim + d) * 3 : nullptr;
  if (cy && tb > 0) { sum = cy[0]; sq = cy[1]; n = cy[2]; }
  constexpr int kAhead = 8;
  for (long long t0 = tb; t0 < T; t0 += kAhead) {
    float xn[kAhead], xo[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; k++) {
      const long long t = t0 + k;
      xn[k] = t < T ? x[t * p.ld_in] : 0.0f;
      xo[k] = (t < T && t - W >= 0) ? x[(t - W) * p.ld_in] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < kAhead; k++) {
      const long long t = t0 + k;
      if (t >= T) break;
      const double xd = (double)xn[k];
      sum = __dadd_rn(sum, xd); if (p.norm_vars) sq = __dadd_rn(sq, rounded(xd * xd)); n = __dadd_rn(n, 1.0);
      if (t - W >= 0) { const double od = (double)xo[k]; sum = __dadd_rn(sum, -od); if (p.norm_vars) sq = __dadd_rn(sq, -rounded(od * od)); n = __dadd_rn(n, -1.0); }
      double m = sum, v = sq, cnt = n;
      if (cnt < (double)W) {
        if (sp) {
          double from_spk = (double)W - cnt;
          if (from_spk > (double)p.speaker_frames) from_spk = (double)p.speaker_frames;
          if (from_spk > s_n) from_spk = s_n;
          if (from_spk > 0.0) {
            const double a = __ddiv_rn(from_spk, s_n);
            m = __dadd_rn(m, rounded(a * s_m));
            v = __dadd_rn(v, rounded(a * s_v));
            cnt = __dadd_rn(cnt, rounded(a * s_n));
          }
        }
        if (cnt < (double)W) {
          double from_glob = (double)W - cnt;
          if (!(g_n > 0.0)) { *p.err = 1; return; }
          if (from_glob > (double)p.global_frames) from_glob = (double)p.global_frames;
          if (from_glob > 0.0) {
            const double a = __ddiv_rn(from_glob, g_n);
            m = __dadd_rn(m, rounded(a * g_m));
            v = __dadd_rn(v, rounded(a * g_v));
            cnt = __dadd_rn(cnt, rounded(a * g_n));
          }
        }
      }
      if (skip) { m = 0.0; v = cnt; }
      float z = xn[k];
      if (p.norm_means) {
        if (cnt < 1.0) { *p.err = 1; return; }
        if (!p.norm_vars) {
          const float alpha = (float)__ddiv_rn(-1.0, cnt);
          const float off = (float)__dadd_rn(0.0, rounded((double)alpha * m));
          z = __fadd_rn(z, off);
        } else {
          const double mean = __ddiv_rn(m, cnt);
          double var = __dadd_rn(__ddiv_rn(v, cnt), -rounded(mean * mean));
          if (var < 1.0e-20) var = 1.0e-20;
          const double scale = __ddiv_rn(1.0, __dsqrt_rn(var));
          z = rounded(z * (float)scale); z = __fadd_rn(z, (float)(-rounded(mean * scale)));
        }
      }
      y[t * p.ld_out] = z;
    }
  }
  if (cy) { cy[0] = sum; cy[1] = sq; cy[2] = n; }
}

}  // namespace

// ------------------------------------------------------------------ host side ------------------
struct k3_feat_plan {
  k3_feat_opts opts;
  FeatParams prm;
  int dim, win_len, win_shift, padded;
  size_t lds_bytes;
  void *d_blob;
};

static int32_t window_shift(const k3_feat_opts &o) { return (int32_t)(o.samp_freq * 0.001 * o.frame_shift_ms); }
static int32_t window_size(const k3_feat_opts &o) { return (int32_t)(o.samp_freq * 0.001 * o.frame_length_ms); }
static float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }
static float inv_mel_scale(float m) { return 700.0f * (expf(m / 1127.0f) - 1.0f); }
static float vtln_warp_freq(float vlo, float vhi, float lo, float hi, float warp, float freq) {
  if (freq < lo || freq > hi) return freq;
  const float l = vlo * std::max(1.0f, warp), h = vhi * std::min(1.0f, warp), scale = 1.0f / warp;
  const float Fl = scale * l, Fh = scale * h;
  const float sl = (Fl - lo) / (l - lo), sr = (hi - Fh) / (hi - h);
  if (freq < l) return lo + sl * (freq - lo);
  if (freq < h) return scale * freq;
  return hi + sr * (freq - hi);
}

extern "C" int k3_feat_plan_create(const k3_feat_opts *opts, k3_feat_plan **out) {
  K3_REQUIRE(opts && out, "k3_feat_plan_create: null argument");
  const k3_feat_opts &o = *opts;
  const int L = window_size(o), shift = window_shift(o);
  int padded = L;
  if (o.round_to_power_of_two) { padded = 1; while (padded < L) padded <<= 1; }
  K3_REQUIRE(L > 1 && shift > 0, "k3_feat_plan_create: bad frame length/shift");
  if (padded != 256 && padded != 512 && padded != 1024) {
    k3::set_error("k3_feat_plan_create: padded window size %d unsupported (this build handles 256, 512 and 1024, i.e. "
                  "frame lengths of 129..1024 samples with --round-to-power-of-two=true)", padded);
    return K3_ERR_UNSUPPORTED;
  }
  const int R = padded / 32, NC = padded / 2;
  // dither != 0 is supported with a counter-based generator (statistically like the reference's RandGauss, not the same stream:
  // parity runs use --dither=0)
  K3_REQUIRE(o.num_bins >= 3 && o.num_bins <= 128, "k3_feat_plan_create: need 3 <= num_bins <= 128");
  K3_REQUIRE(o.preemph_coeff >= 0.0f && o.preemph_coeff <= 1.0f, "k3_feat_plan_create: preemph_coeff out of [0,1]");
  if (o.feature_type == 1) K3_REQUIRE(o.num_ceps >= 1 && o.num_ceps <= o.num_bins, "num-ceps cannot be larger than num-mel-bins");

  // window: FeatureWindowFunction, feature-window.cc:109-135
  std::vector<float> window(L);
  const double a = 6.283185307179586476925286766559005 / (L - 1);
  for (int i = 0; i < L; i++) {
    const double x = i;
    switch (o.window_type) {
      case 0: window[i] = 0.5 - 0.5 * cos(a * x); break;
      case 1: window[i] = sin(0.5 * a * x); break;
      case 2: window[i] = 0.54 - 0.46 * cos(a * x); break;
      case 3: window[i] = pow(0.5 - 0.5 * cos(a * x), 0.85); break;
      case 4: window[i] = 1.0; break;
      case 5: window[i] = o.blackman_coeff - 0.5 * cos(a * x) + (0.5 - o.blackman_coeff) * cos(2 * a * x); break;
      default: return k3::fail(K3_ERR_ARG, "Invalid window type", __FILE__, __LINE__);
    }
  }
  // mel banks: MelBanks::MelBanks, mel-computations.cc:33-142
  const int nb = o.num_bins, nfft_bins = padded / 2;
  const float nyq = 0.5f * o.samp_freq, low = o.low_freq;
  const float high = (o.high_freq > 0.0f) ? o.high_freq : nyq + o.high_freq;
  if (low < 0.0f || low >= nyq || high <= 0.0f || high > nyq || high <= low) {
    k3::set_error("Bad values in options: low-freq %g and high-freq %g vs. nyquist %g", low, high, nyq);
    return K3_ERR_ARG;
  }
  const float bin_width = o.samp_freq / padded;
  const float mlow = mel_scale(low), mhigh = mel_scale(high), delta = (mhigh - mlow) / (nb + 1);
  float vlo = o.vtln_low, vhi = o.vtln_high;
  if (vhi < 0.0f) vhi += nyq;
  if (o.vtln_warp != 1.0f && (vlo < 0.0f || vlo <= low || vlo >= high || vhi <= 0.0f || vhi >= high || vhi <= vlo)) {
    k3::set_error("Bad values in options: vtln-low %g and vtln-high %g, versus low-freq %g and high-freq %g", vlo, vhi, low, high);
    return K3_ERR_ARG;
  }
  std::vector<int> meta(3 * nb);
  std::vector<float> binw;
  for (int bin = 0; bin < nb; bin++) {
    float left = mlow + bin * delta, center = mlow + (bin + 1) * delta, right = mlow + (bin + 2) * delta;
    if (o.vtln_warp != 1.0f) {
      left = mel_scale(vtln_warp_freq(vlo, vhi, low, high, o.vtln_warp, inv_mel_scale(left)));
      center = mel_scale(vtln_warp_freq(vlo, vhi, low, high, o.vtln_warp, inv_mel_scale(center)));
      right = mel_scale(vtln_warp_freq(vlo, vhi, low, high, o.vtln_warp, inv_mel_scale(right)));
    }
    int first = -1, last = -1;
    std::vector<float> w(nfft_bins, 0.0f);
    for (int i = 0; i < nfft_bins; i++) {
      const float mel = mel_scale(bin_width * i);
      if (mel > left && mel < right) {
        w[i] = (mel <= center) ? (mel - left) / (center - left) : (right - mel) / (right - center);
        if (first == -1) first = i;
        last = i;
      }
    }
    if (first == -1) return k3::fail(K3_ERR_ARG, "You may have set --num-mel-bins too large.", __FILE__, __LINE__);
    meta[3 * bin] = first; meta[3 * bin + 1] = last + 1 - first; meta[3 * bin + 2] = (int)binw.size();
    for (int i = first; i <= last; i++) binw.push_back(w[i]);
    if (o.htk_mode && bin == 0 && mlow != 0.0f) binw[meta[2]] = 0.0f;
  }
  // twiddles: exact to float64 (the data path is float64; the reference builds exp(-2 pi i k / N) by a float32 recurrence, srfft.cc:370-376, which is part of ITS rounding error)
  std::vector<double> tw256(2 * (size_t)NC), tw512(2 * ((size_t)NC / 2 + 1));
  for (int m = 0; m < NC; m++) { const double ang = -6.283185307179586476925286766559005 * m / (double)NC; tw256[2 * m] = cos(ang); tw256[2 * m + 1] = sin(ang); }
  for (int k = 0; k <= NC / 2; k++) { const double ang = -6.283185307179586476925286766559005 * k / (double)padded; tw512[2 * k] = cos(ang); tw512[2 * k + 1] = sin(ang); }
  // MFCC tables
  std::vector<float> dct, lifter;
  if (o.feature_type == 1) {
    dct.resize((size_t)o.num_ceps * nb);
    const float n0 = std::sqrt(1.0f / (float)nb), n1 = std::sqrt(2.0f / (float)nb);
    for (int j = 0; j < nb; j++) dct[j] = n0;
    for (int k = 1; k < o.num_ceps; k++)
      for (int n = 0; n < nb; n++) dct[(size_t)k * nb + n] = n1 * std::cos((double)M_PI / nb * (n + 0.5) * k);
    lifter.resize(o.num_ceps);
    for (int i = 0; i < o.num_ceps; i++) lifter[i] = 1.0 + 0.5 * o.cepstral_lifter * sin(M_PI * i / o.cepstral_lifter);
  }
  // one device blob
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  size_t off_win = 0, off_tw256 = al(off_win + window.size() * 4), off_tw512 = al(off_tw256 + tw256.size() * 8),
         off_meta = al(off_tw512 + tw512.size() * 8), off_binw = al(off_meta + meta.size() * 4),
         off_dct = al(off_binw + binw.size() * 4), off_lift = al(off_dct + dct.size() * 4), total = al(off_lift + lifter.size() * 4 + 4);
  std::vector<char> host(total, 0);
  memcpy(&host[off_win], window.data(), window.size() * 4);
  memcpy(&host[off_tw256], tw256.data(), tw256.size() * 8);
  memcpy(&host[off_tw512], tw512.data(), tw512.size() * 8);
  memcpy(&host[off_meta], meta.data(), meta.size() * 4);
  memcpy(&host[off_binw], binw.data(), binw.size() * 4);
  if (!dct.empty()) { memcpy(&host[off_dct], dct.data(), dct.size() * 4); memcpy(&host[off_lift], lifter.data(), lifter.size() * 4); }
  void *blob = nullptr;
  K3_HIP_CHECK(hipMalloc(&blob, total));
  K3_HIP_CHECK(hipMemcpy(blob, host.data(), total, hipMemcpyHostToDevice));

  k3_feat_plan *pl = new k3_feat_plan();
  pl->opts = o; pl->win_len = L; pl->win_shift = shift; pl->padded = padded; pl->d_blob = blob;
  pl->dim = (o.feature_type == 1) ? o.num_ceps : nb + (o.use_energy ? 1 : 0);
  FeatParams &p = pl->prm;
  p.win_len = L; p.win_shift = shift; p.snip_edges = o.snip_edges; p.remove_dc = o.remove_dc_offset;
  p.dither = o.dither; p.dither_seed = 0x1234567u;
  p.use_energy = o.use_energy; p.raw_energy = o.raw_energy; p.htk_compat = o.htk_compat; p.use_log = o.use_log_fbank;
  p.use_power = (o.feature_type == 1) ? 1 : o.use_power; p.htk_mode = o.htk_mode; p.feature_type = o.feature_type;
  p.num_bins = nb; p.num_ceps = o.num_ceps; p.dim = pl->dim;
  p.has_energy_floor = o.energy_floor > 0.0f; p.log_energy_floor = p.has_energy_floor ? logf(o.energy_floor) : 0.0f;
  p.has_lifter = (o.feature_type == 1 && o.cepstral_lifter != 0.0f);
  p.total_w = (int)binw.size(); p.preemph = o.preemph_coeff;
  char *b = (char *)blob;
  p.window = (const float *)(b + off_win); p.tw256 = (const double2 *)(b + off_tw256); p.tw512 = (const double2 *)(b + off_tw512);
  p.bin_meta = (const int *)(b + off_meta); p.bin_w = (const float *)(b + off_binw);
  p.dct = (const float *)(b + off_dct); p.lifter = (const float *)(b + off_lift);
  const size_t meta_bytes = ((3 * nb * 4 + 15) / 16) * 16, binw_bytes = ((binw.size() * 4 + 15) / 16) * 16;
  pl->lds_bytes = lds_fixed_bytes(R) + meta_bytes + binw_bytes + (size_t)kFramesPerIter * frame_buf_bytes(R);
  *out = pl;
  return K3_OK;
}

extern "C" void k3_feat_plan_destroy(k3_feat_plan *plan) {
  if (!plan) return;
  if (plan->d_blob) (void)hipFree(plan->d_blob);
  delete plan;
}

extern "C" int32_t k3_feat_dim(const k3_feat_plan *plan) { return plan ? plan->dim : -1; }

extern "C" int32_t k3_feat_num_frames(const k3_feat_plan *plan, int64_t nsamp) {
  if (!plan) return -1;
  const int64_t shift = plan->win_shift, len = plan->win_len;
  if (plan->opts.snip_edges) return nsamp < len ? 0 : (int32_t)(1 + (nsamp - len) / shift);
  return (int32_t)((nsamp + shift / 2) / shift);
}

template <typename TS>
static int feat_launch(k3_feat_plan *plan, const TS *d_waves, const int64_t *d_wave_offsets, const int64_t *d_frame_offsets, int32_t num_utts, int64_t total_frames,
                       float *d_feats, int64_t ld, void *stream) {
  K3_REQUIRE(plan && d_waves && d_wave_offsets && d_frame_offsets && d_feats, "k3_feat_compute_batch: null argument");
  K3_REQUIRE(num_utts >= 0 && total_frames >= 0 && ld >= plan->dim, "k3_feat_compute_batch: bad sizes");
  if (total_frames == 0 || num_utts == 0) return K3_OK;
  const int frames_per_block = 64;
  const int64_t blocks = (total_frames + frames_per_block - 1) / frames_per_block;
  K3_REQUIRE(blocks < (1ll << 31), "k3_feat_compute_batch: too many frames for one launch");
  static std::once_flag once; int rc = K3_OK;
  std::call_once(once, [&]() { rc = [&]() -> int {
    K3_HIP_CHECK(hipFuncSetAttribute((const void *)k3_feat_kernel<TS, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    K3_HIP_CHECK(hipFuncSetAttribute((const void *)k3_feat_kernel<TS, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    K3_HIP_CHECK(hipFuncSetAttribute((const void *)k3_feat_kernel<TS, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); return K3_OK; }(); });
  if (rc) return rc;
  const dim3 grid((unsigned)blocks), block(kThreads); hipStream_t st = (hipStream_t)stream;
  switch (plan->padded) {      // one instantiation per padded window size
    case 256: hipLaunchKernelGGL((k3_feat_kernel<TS, 8>), grid, block, plan->lds_bytes, st, plan->prm, d_waves, d_wave_offsets, d_frame_offsets, (int)num_utts,
        total_frames, d_feats, ld, frames_per_block);
    break;
    case 512: hipLaunchKernelGGL((k3_feat_kernel<TS, 16>), grid, block, plan->lds_bytes, st, plan->prm, d_waves, d_wave_offsets, d_frame_offsets,
        (int)num_utts, total_frames, d_feats, ld, frames_per_block);
    break;
    default: hipLaunchKernelGGL((k3_feat_kernel<TS, 32>), grid, block, plan->lds_bytes, st, plan->prm, d_waves, d_wave_offsets, d_frame_offsets, (int)num_utts,
        total_frames, d_feats, ld, frames_per_block);
    break;
  }
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
extern "C" int k3_feat_compute_batch(k3_feat_plan *plan, const float *d_waves, const int64_t *d_wave_offsets, const int64_t *d_frame_offsets, int32_t num_utts,
    int64_t total_frames,
                                     float *d_feats, int64_t ld, void *stream) {
  return feat_launch(plan, d_waves, d_wave_offsets, d_frame_offsets, num_utts, total_frames, d_feats, ld, stream);
}
extern "C" int k3_feat_compute_batch_pcm16(k3_feat_plan *plan, const int16_t *d_waves, const int64_t *d_wave_offsets, const int64_t *d_frame_offsets,
    int32_t num_utts, int64_t total_frames,
                                           float *d_feats, int64_t ld, void *stream) {
  return feat_launch(plan, d_waves, d_wave_offsets, d_frame_offsets, num_utts, total_frames, d_feats, ld, stream);
}

extern "C" int k3_cmvn_offline_batch(float *d_feats, int64_t ld, int32_t dim, const int64_t *d_frame_offsets,
                                     int32_t num_utts, int32_t norm_vars, double *d_stats, void *stream) {
  K3_REQUIRE(d_feats && d_frame_offsets && dim > 0 && ld >= dim && num_utts >= 0, "k3_cmvn_offline_batch: bad argument");
  if (num_utts == 0) return K3_OK;
  hipLaunchKernelGGL(k3_cmvn_kernel, dim3((unsigned)num_utts, (unsigned)((dim + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                     d_feats, ld, (int)dim, d_frame_offsets, (int)norm_vars, d_stats);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}

extern "C" int k3_cmvn_online_batch(const float *d_in, int64_t ld_in, float *d_out, int64_t ld_out, int32_t dim, const int64_t *d_frame_offsets,
                                    int32_t num_utts, const k3_online_cmvn_opts *opts, const double *d_global_stats, const double *d_speaker_stats,
                                    const int32_t *skip_dims, int32_t num_skip_dims, void *stream) {
  return k3_cmvn_online_batch_resume(d_in, ld_in, d_out, ld_out, dim, d_frame_offsets, num_utts, opts, d_global_stats, d_speaker_stats, skip_dims,
      num_skip_dims, nullptr, nullptr, stream);
}
extern "C" int k3_cmvn_online_batch_resume(const float *d_in, int64_t ld_in, float *d_out, int64_t ld_out, int32_t dim, const int64_t *d_frame_offsets,
                                           int32_t num_utts, const k3_online_cmvn_opts *opts, const double *d_global_stats, const double *d_speaker_stats,
                                           const int32_t *skip_dims, int32_t num_skip_dims, const int64_t *d_t_begin, double *d_carry, void *stream) {
  K3_REQUIRE((d_t_begin == nullptr) == (d_carry == nullptr), "k3_cmvn_online_batch_resume: t_begin and carry go together");
  K3_REQUIRE(d_in && d_out && d_in != d_out && d_frame_offsets && opts && d_global_stats && dim > 0 && ld_in >= dim && ld_out >= dim && num_utts >= 0,
             "k3_cmvn_online_batch: bad argument (in-place operation is not possible: the window needs the raw frames)");
  // OnlineCmvnOptions::Check (feat/online-feature.h:226-229) and the assertion in OnlineCmvn::GetFrame (:465)
  K3_REQUIRE(opts->speaker_frames <= opts->cmn_window && opts->global_frames <= opts->speaker_frames && opts->cmn_window > 0 && opts->global_frames >= 0,
             "k3_cmvn_online_batch: need global_frames <= speaker_frames <= cmn_window");
  K3_REQUIRE(opts->normalize_mean || !opts->normalize_variance, "k3_cmvn_online_batch: cannot normalize the variance but not the mean");
  K3_REQUIRE(num_skip_dims == 0 || skip_dims, "k3_cmvn_online_batch: skip_dims missing");
  if (num_utts == 0) return K3_OK;
  OnlineCmvnParams p{};
  p.in = d_in; p.out = d_out; p.ld_in = ld_in; p.ld_out = ld_out; p.dim = dim; p.frame_off = (const long long *)d_frame_offsets;
  p.cmn_window = opts->cmn_window; p.speaker_frames = opts->speaker_frames; p.global_frames = opts->global_frames;
  p.norm_means = opts->normalize_mean; p.norm_vars = opts->normalize_variance;
  p.global_stats = d_global_stats; p.speaker_stats = d_speaker_stats;
  for (int32_t i = 0; i < num_skip_dims; i++) {
    K3_REQUIRE(skip_dims[i] >= 0 && skip_dims[i] < dim && skip_dims[i] < 256, "k3_cmvn_online_batch: skip dimension out of range (0 <= d < min(dim, 256))");
    p.skip[skip_dims[i] >> 6] |= 1ull << (skip_dims[i] & 63);
  }
  static int *d_err = nullptr;
  if (!d_err) { K3_HIP_CHECK(hipMalloc(&d_err, sizeof(int))); K3_HIP_CHECK(hipMemset(d_err, 0, sizeof(int))); }
  p.err = d_err; p.t_begin = (const long long *)d_t_begin; p.carry = d_carry;
  hipLaunchKernelGGL(k3_cmvn_online_kernel, dim3((unsigned)num_utts, (unsigned)((dim + 63) / 64)), dim3(64), 0, (hipStream_t)stream, p);
  K3_HIP_CHECK(hipGetLastError());
  int h_err = 0;
  K3_HIP_CHECK(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  K3_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  if (h_err) {
    K3_HIP_CHECK(hipMemset(d_err, 0, sizeof(int)));
    K3_REQUIRE(false, "k3_cmvn_online_batch: insufficient stats (count < 1 or empty global stats), the reference raises an error here");
  }
  return K3_OK;
}

int k3::cmvn_online_resume_segs_async(const k3::CmvnSeg *d_segs, int num_segs, int dim, const void *opts_, const double *d_global_stats, void *stream) {
  const k3_online_cmvn_opts *opts = static_cast<const k3_online_cmvn_opts *>(opts_);
  K3_REQUIRE(d_segs && num_segs > 0 && opts && d_global_stats && dim > 0, "cmvn_online_resume_segs_async: bad argument");
  OnlineCmvnParams p{};
  p.segs = d_segs; p.ld_in = dim; p.ld_out = dim; p.dim = dim;
  p.cmn_window = opts->cmn_window;
  p.speaker_frames = opts->speaker_frames;
  p.global_frames = opts->global_frames;
  p.norm_means = opts->normalize_mean;
  p.norm_vars = opts->normalize_variance;
  p.global_stats = d_global_stats;
  static int *d_err = nullptr;      // (never read here: see the header)
  if (!d_err) { K3_HIP_CHECK(hipMalloc(&d_err, sizeof(int))); K3_HIP_CHECK(hipMemset(d_err, 0, sizeof(int))); }
  p.err = d_err;
  hipLaunchKernelGGL(k3_cmvn_online_kernel, dim3((unsigned)num_segs, (unsigned)((dim + 63) / 64)), dim3(64), 0, (hipStream_t)stream, p);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
int k3::cmvn_online_resume_async(const float *d_in, long long ld_in, float *d_out, long long ld_out, int dim, const long long *d_frame_offsets, int num_utts, const void *opts_,
                                 const double *d_global_stats, const long long *d_t_begin, double *d_carry, void *stream) {
  const k3_online_cmvn_opts *opts = static_cast<const k3_online_cmvn_opts *>(opts_);
  K3_REQUIRE(d_in && d_out && d_in != d_out && d_frame_offsets && opts && d_global_stats && d_t_begin && d_carry && dim > 0 && num_utts > 0,
      "cmvn_online_resume_async: bad argument");
  OnlineCmvnParams p{};
  p.in = d_in; p.out = d_out; p.ld_in = ld_in; p.ld_out = ld_out; p.dim = dim; p.frame_off = d_frame_offsets;
  p.cmn_window = opts->cmn_window;
  p.speaker_frames = opts->speaker_frames;
  p.global_frames = opts->global_frames;
  p.norm_means = opts->normalize_mean;
  p.norm_vars = opts->normalize_variance;
  p.global_stats = d_global_stats; p.speaker_stats = nullptr; p.t_begin = d_t_begin; p.carry = d_carry;
  static int *d_err = nullptr;      // (never read here: see the header)
  if (!d_err) { K3_HIP_CHECK(hipMalloc(&d_err, sizeof(int))); K3_HIP_CHECK(hipMemset(d_err, 0, sizeof(int))); }
  p.err = d_err;
  hipLaunchKernelGGL(k3_cmvn_online_kernel, dim3((unsigned)num_utts, (unsigned)((dim + 63) / 64)), dim3(64), 0, (hipStream_t)stream, p);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}

extern "C" void k3_online_cmvn_opts_default(k3_online_cmvn_opts *o) {
  if (!o) return;
  o->cmn_window = 600; o->speaker_frames = 600; o->global_frames = 200; o->normalize_mean = 1; o->normalize_variance = 0;
}

// ---- ResampleWaveform (feat/resample.cc:363-372) = LinearResample (:33-230) with cutoff 0.99 * 0.5 * min(rates), six zero crossings, flush = true: what
// OfflineFeatureTpl::ComputeFeatures does when a file's rate differs from --sample-frequency (feat/feature-common-inl.h:29-57, --allow-downsample / --allow-upsample).
// The windowed-sinc weights of the out_rate / gcd phases are made on the host with the reference's float / double mix (FilterFunc takes and returns float, evaluates in double);
// one thread per output sample forms its dot product in tap order.
namespace {
struct ResamplePlan { int in_unit, out_unit, max_taps; std::vector<int> first, ntaps; std::vector<float> w; };
int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }
ResamplePlan make_resample_plan(int rate_in, int rate_out) {
  ResamplePlan r; const float min_freq = (float)std::min(rate_in, rate_out), cutoff = (float)(0.99 * 0.5 * min_freq); const int num_zeros = 6;
  const int base = gcd_i(rate_in, rate_out); r.in_unit = rate_in / base; r.out_unit = rate_out / base;
  const double window_width = num_zeros / (2.0 * cutoff);
  auto filter_func = [&](float t) -> float {
    float window, filter;
    if (std::fabs(t) < num_zeros / (2.0 * cutoff)) window = (float)(0.5 * (1 + std::cos(6.283185307179586476925286766559005 * cutoff / num_zeros * t))); else window = 0.0f;
    if (t != 0) filter = (float)(std::sin(6.283185307179586476925286766559005 * cutoff * t) / (M_PI * t)); else filter = 2 * cutoff;
    return filter * window;
  };
  r.first.resize(r.out_unit); r.ntaps.resize(r.out_unit); r.max_taps = 0; std::vector<std::vector<float>> ws(r.out_unit);
  for (int i = 0; i < r.out_unit; i++) {
    const double output_t = i / (double)rate_out, min_t = output_t - window_width, max_t = output_t + window_width;
    const int lo = (int)std::ceil(min_t * rate_in), hi = (int)std::floor(max_t * rate_in), n = hi - lo + 1;
    r.first[i] = lo; r.ntaps[i] = n; r.max_taps = std::max(r.max_taps, n); ws[i].resize(n);
    for (int j = 0; j < n; j++) { const double input_t = (lo + j) / (double)rate_in, delta_t = input_t - output_t; ws[i][j] = filter_func((float)delta_t) / rate_in; }
  }
  r.w.assign((size_t)r.out_unit * r.max_taps, 0.0f);
  for (int i = 0; i < r.out_unit; i++) std::copy(ws[i].begin(), ws[i].end(), r.w.begin() + (size_t)i * r.max_taps);
  return r;
}
long long resample_num_out(int rate_in, int rate_out, long long n_in) {      // GetNumOutputSamples(n, flush = true), :61-103
  const long long tick = (long long)rate_in / gcd_i(rate_in, rate_out) * rate_out, per_in = tick / rate_in, per_out = tick / rate_out, interval = n_in * per_in;
  if (interval <= 0) return 0;
  long long last = interval / per_out; if (last * per_out == interval) last--;
  return last + 1;
}
__global__ __launch_bounds__(256) void k3_resample_kernel(const float *in, const long long *in_off, float *out, const long long *out_off, int num_utts,
    int in_unit, int out_unit, int max_taps,
                                                          const int *first, const int *ntaps, const float *w) {
  const int u = blockIdx.y; const long long i0 = in_off[u], n_in = in_off[u + 1] - i0, o0 = out_off[u], n_out = out_off[u + 1] - o0;
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < n_out; s += (long long)gridDim.x * 256) {
    const long long unit = s / out_unit; const int ph = (int)(s - unit * out_unit);
    const long long f = first[ph] + unit * in_unit; const float *wp = w + (long long)ph * max_taps; const int n = ntaps[ph];
    float acc = 0.0f;
    // (samples before the start / beyond the end do not exist: flush = true, no remainder)
    for (int t = 0; t < n; t++) {
      const long long idx = f + t;
      if (idx >= 0 && idx < n_in) acc += wp[t] * in[i0 + idx];
    }
    out[o0 + s] = acc;
  }
}
}  // namespace

extern "C" int64_t k3_resample_num_samples(int32_t rate_in, int32_t rate_out, int64_t num_in) {
  return (rate_in > 0 && rate_out > 0 && num_in >= 0) ? resample_num_out(rate_in, rate_out, num_in) : -1;
}
extern "C" int k3_resample_batch(int32_t rate_in, int32_t rate_out, const float *d_in, const int64_t *h_in_offsets, int32_t num_utts, float *d_out,
    const int64_t *h_out_offsets, void *stream) {
  K3_REQUIRE(rate_in > 0 && rate_out > 0 && rate_in != rate_out && d_in && d_out && h_in_offsets && h_out_offsets && num_utts >= 0, "k3_resample_batch: bad argument");
  if (num_utts == 0) return K3_OK;
  long long max_out = 0;
  for (int u = 0; u < num_utts; u++) {
    K3_REQUIRE(h_out_offsets[u + 1] - h_out_offsets[u] == resample_num_out(rate_in, rate_out, h_in_offsets[u + 1] - h_in_offsets[u]),
        "k3_resample_batch: output offsets do not match k3_resample_num_samples");
    max_out = std::max<long long>(max_out, h_out_offsets[u + 1] - h_out_offsets[u]);
  }
  const ResamplePlan pl = make_resample_plan(rate_in, rate_out);
  // one allocation for the tables and the offsets; the call is synchronous (a rate mismatch is the rare path of a feature program)
  const size_t nb_w = pl.w.size() * 4, nb_i = (size_t)pl.out_unit * 4, nb_o = (size_t)(num_utts + 1) * 8; char *d = nullptr;
  K3_HIP_CHECK(hipMalloc((void **)&d, nb_w + 2 * nb_i + 2 * nb_o + 64));
  float *d_w = (float *)d;
  int *d_first = (int *)(d + nb_w), *d_nt = (int *)(d + nb_w + nb_i);
  long long *d_io = (long long *)(d + ((nb_w + 2 * nb_i + 15) & ~(size_t)15)), *d_oo = d_io + num_utts + 1;
  hipStream_t st = (hipStream_t)stream; int rc = K3_OK;
  if (hipMemcpyAsync(d_w, pl.w.data(), nb_w, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(d_first, pl.first.data(), nb_i, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_nt, pl.ntaps.data(), nb_i, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(d_io, h_in_offsets, nb_o, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_oo, h_out_offsets, nb_o, hipMemcpyHostToDevice, st) != hipSuccess) rc = K3_ERR_HIP;
  if (rc == K3_OK && max_out > 0) {
    hipLaunchKernelGGL(k3_resample_kernel, dim3((unsigned)std::min<long long>(1024, (max_out + 255) / 256), (unsigned)num_utts), dim3(256), 0, st, d_in, d_io,
        d_out, d_oo, num_utts, pl.in_unit, pl.out_unit, pl.max_taps, d_first, d_nt, d_w);
    if (hipGetLastError() != hipSuccess) rc = K3_ERR_HIP;
  }
  if (hipStreamSynchronize(st) != hipSuccess) rc = K3_ERR_HIP;
  (void)hipFree(d);
  K3_REQUIRE(rc == K3_OK, "k3_resample_batch: HIP error");
  return K3_OK;
}
