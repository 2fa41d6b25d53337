// k3_ivector.hip -- online i-vector extraction of whole utterances on gfx950.  Paths relative to the reference's src/.
//
// What is computed, per utterance, is what OnlineIvectorFeature computes when ivector-extract-online2 drives it with every frame weighted 1 and a
// fresh adaptation state (online2/online-ivector-feature.cc): an i-vector after every frame t with t % ivector_period == 0, from statistics of
// frames 0..t, solved by a warm-started conjugate-gradient descent.  The GPU reference BatchedIvectorExtractorCuda::GetIvectors
// (cudafeat/feature-online-batched-ivector-cuda.h:30-61) has the same stages (splice, LDA, posteriors, statistics, solution) in float and with a
// direct solve; opts.exact_solve = 1 gives that solution method, the default follows the CPU reference because that is what the golden vectors of
// tests/golden/ivector pin.
//
//   stage                        kernel                          reference
//   online CMVN (posterior side) k3_cmvn_online_batch            OnlineCmvn, online-ivector-feature.cc:144-163
//   splice(+-c) + LDA            ivec_splice_lda_kernel          OnlineSpliceFrames + OnlineTransform, :144-163 / :239-243 (no CMVN on the statistics side)
//   UBM log-likes + pruning      ivec_posterior_kernel           DiagGmm::LogLikelihoods gmm/diag-gmm.cc:557-586; VectorToPosteriorEntry hmm/posterior.cc:440-510
//   derived model terms          ivec_sigma_inv_m / ivec_u       IvectorExtractor::ComputeDerivedVars ivector/ivector-extractor.cc:208-218 (once, at create)
//   statistics + solution        ivec_estimate_kernel            OnlineIvectorEstimationStats::AccStats :611-670, GetIvector :732-756, LinearCgd matrix/optimization.cc:453-560
//
// Layout in HBM: model terms in fp64 (U_g [G][R][R] symmetric, Sigma_g^-1 M_g [G][D][R]) like the reference's Matrix<double>; the UBM in fp32,
// stored [D][G] so that a wave reading one feature dimension of 64 Gaussians reads 256 contiguous bytes.  One workgroup per utterance walks the
// periods in order (the statistics are cumulative and the solver is warm-started, so periods of one utterance are serial; utterances are not).
// The quadratic term lives in LDS when it fits (R <= 128: R*R*8 B <= 128 KB of the CU's 160 KB), else in a per-utterance global scratch.
#include "k3_common.h"
#include <cmath>
#include <cstdlib>
#include <memory>
#include <vector>

namespace {
constexpr int kBlock = 256;
constexpr int kWave = 64;

struct DevBuf {
  void *p = nullptr; size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return K3_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    K3_HIP_CHECK(hipMalloc(&p, bytes)); cap = bytes; return K3_OK;
  }
  ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

struct k3_ivector {
  k3_ivector_opts o;
  int32_t F = 0, D = 0, G = 0, R = 0, splice = 0, has_offset = 0;
  double prior_offset = 0;
  float *lda = nullptr;              // [D][lda_cols]
  double *global_stats = nullptr;    // [2][F+1]
  float *gconsts = nullptr, *miv_t = nullptr, *iv_t = nullptr;   // [G], [D][G], [D][G]
  double *U = nullptr, *SM = nullptr;                            // [G][R][R], [G][D][R]
  bool quad_in_lds = true;
  DevBuf frame_off, cmvn, xpost, xstats, post_g, post_w, post_n, quad, state; int acc_tail = 0;
  ~k3_ivector() { for (void *p : {(void *)lda, (void *)global_stats, (void *)gconsts, (void *)miv_t, (void *)iv_t, (void *)U, (void *)SM}) if (p) (void)hipFree(p); }
};

// ---------------------------------------------------------------------------------------------------------------- derived model terms
// SM[g][d][r] = sum_e SigmaInv_g[d][e] M_g[e][r]   (SigmaInv packed lower triangle, row-major: (i, j<=i) at i(i+1)/2 + j)
__global__ void ivec_sigma_inv_m(const double *__restrict__ M, const double *__restrict__ Sp, double *__restrict__ SM, int G, int D, int R) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (long)G * D * R) return;
  const int r = (int)(i % R), d = (int)((i / R) % D), g = (int)(i / ((long)R * D));
  const double *S = Sp + (size_t)g * (D * (D + 1) / 2), *Mg = M + (size_t)g * D * R; double a = 0;
  for (int e = 0; e < D; e++) { const double s = e <= d ? S[d * (d + 1) / 2 + e] : S[e * (e + 1) / 2 + d]; a += s * Mg[(size_t)e * R + r]; }
  SM[i] = a;
}
// U[g][r][s] = sum_d M_g[d][r] SM[g][d][s], computed for s <= r and mirrored (the reference keeps U_g as a packed symmetric row)
__global__ void ivec_u(const double *__restrict__ M, const double *__restrict__ SM, double *__restrict__ U, int G, int D, int R) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (long)G * R * R) return;
  const int s = (int)(i % R), r = (int)((i / R) % R), g = (int)(i / ((long)R * R));
  if (s > r) return;
  const double *Mg = M + (size_t)g * D * R, *SMg = SM + (size_t)g * D * R; double a = 0;
  for (int d = 0; d < D; d++) a += Mg[(size_t)d * R + r] * SMg[(size_t)d * R + s];
  U[((size_t)g * R + r) * R + s] = a; U[((size_t)g * R + s) * R + r] = a;
}

// One stream of a batched streaming call (k3_ivector_stream_accept_batch): where each stage of this call reads and writes for that stream.  A device array of these is what the
// batched kernels index by stream instead of the packed utterance layout of the whole-utterance path.
struct BatchSeg {
  const float *raw_old, *cm_old; float *raw, *cm;                        // the stream's raw / normalised frame buffers before and after this call
  // raw_old[raw_from .. + keep) -> raw[0 ..), the call's feature rows feat_off .. + n_new -> raw[keep ..); cm_old[cm_from .. + cm_keep) -> cm[0 ..)
  long long raw_from, keep, n_new, feat_off, cm_from, cm_keep;
  long long nP, out_off, row0_cm, row0_raw, cm_rows, raw_rows;           // posterior stage: nP frames, rows out_off .. of the packed work arrays; first frame's row in cm / raw
  float *px; int32_t *pg; float *pw; int32_t *pn;                        // where the new posterior-stage rows go (behind the rows still waiting for their period)
  const float *e_px; const int32_t *e_pg; const float *e_pw; const int32_t *e_pn;      // the waiting rows from stream frame t_base on
  double *est, *x_io, *chol, *quad; long long t_base, t_limit; int k_begin, nk; float *rows, *latest;
  // after the estimates: rows shift_from .. + left of the waiting arrays move to the other buffers
  float *px_alt;
  int32_t *pg_alt;
  float *pw_alt;
  int32_t *pn_alt;
  long long shift_from, left;
};
// last stream whose out_off <= row (streams without rows share their successor's offset)
__device__ __forceinline__ int seg_of_row(const BatchSeg *segs, int n, long long row) {
  int lo = 0, hi = n; while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (segs[m].out_off <= row) lo = m; else hi = m; } return lo;
}

// ---------------------------------------------------------------------------------------------------------------- splice + LDA
// out[t][d] = offset[d] + sum_{o=-lc..rc} sum_f lda[d][(o+lc) F + f] in[clamp(t+o)][f]; one thread per (frame, output dim); rows of one utterance only
__global__ void ivec_splice_lda_kernel(const float *__restrict__ in, int64_t ld_in, const int64_t *__restrict__ frame_off, int num_utts, const float *__restrict__ lda,
                                       int lda_cols, int has_offset, int F, int D, int lc, int rc, float *__restrict__ out, int64_t total_frames, int64_t row0) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= total_frames * D) return;      // rows row0 .. row0 + total_frames - 1 of `in`, written to out rows 0 ..
  const int64_t row = row0 + i / D; const int d = (int)(i % D);
  int lo = 0, hi = num_utts;                                  // utterance of this row: last u with frame_off[u] <= row
  while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (frame_off[m] <= row) lo = m; else hi = m; }
  const int64_t b = frame_off[lo], e = frame_off[lo + 1];
  const float *w = lda + (size_t)d * lda_cols; float a = 0.f;
  for (int o = -lc; o <= rc; o++) {
    int64_t t = row + o; t = t < b ? b : (t >= e ? e - 1 : t);
    const float *x = in + t * ld_in, *wo = w + (size_t)(o + lc) * F;
    for (int f = 0; f < F; f++) a = fmaf(wo[f], x[f], a);
  }
  out[i] = has_offset ? w[lda_cols - 1] + a : a;
}

// the same for the streams of a batched call: frame `local` of stream u is row row0 + local of its own buffer (clamped to that buffer: the first row is the stream's first frame
// or has its left context in the buffer, the last row is only reached when the stream has ended); which = 0: normalised frames -> packed xpost, 1: the statistics' features -> px
__global__ void ivec_splice_lda_multi_kernel(const BatchSeg *__restrict__ segs, int nseg, int which, int stats_from_cm, const float *__restrict__ lda,
    int lda_cols, int has_offset, int F, int D,
                                             int lc, int rc, float *__restrict__ xpost, int64_t total_frames) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= total_frames * D) return;
  const int64_t row = i / D; const int d = (int)(i % D);
  const BatchSeg &g = segs[seg_of_row(segs, nseg, row)];
  const int64_t local = row - g.out_off; const bool from_cm = which == 0 || stats_from_cm;
  const float *in = from_cm ? g.cm : g.raw; const int64_t e = from_cm ? g.cm_rows : g.raw_rows, r0 = from_cm ? g.row0_cm : g.row0_raw;
  const float *w = lda + (size_t)d * lda_cols; float a = 0.f;
  for (int o = -lc; o <= rc; o++) {
    int64_t t = r0 + local + o; t = t < 0 ? 0 : (t >= e ? e - 1 : t);
    const float *x = in + t * F, *wo = w + (size_t)(o + lc) * F;
    for (int f = 0; f < F; f++) a = fmaf(wo[f], x[f], a);
  }
  const float v = has_offset ? w[lda_cols - 1] + a : a;
  if (which == 0) xpost[i] = v; else g.px[local * D + d] = v;
}

// ---------------------------------------------------------------------------------------------------------------- posteriors
// One wave per frame.  The log-likelihoods of the G Gaussians go to LDS; the num_gselect best that pass the min-post cut are taken one at a time
// (a wave arg-max each), then pruned and renormalised by lane 0 exactly as VectorToPosteriorEntry does (float arithmetic, same order).
__global__ void __launch_bounds__(kBlock) ivec_posterior_kernel(const float *__restrict__ x, int D, int G, const float *__restrict__ gconsts, const float *__restrict__ miv_t,
                                                                const float *__restrict__ iv_t, int num_gselect, float min_post, float post_scale, int64_t total_frames,
                                                                int32_t *__restrict__ post_g, float *__restrict__ post_w, int32_t *__restrict__ post_n,
                                                                    const BatchSeg *__restrict__ segs = nullptr, int nseg = 0,
                                                                const float *__restrict__ frame_w = nullptr) {
  extern __shared__ float s_ll[];                             // [waves per block][G] log-likes, then [waves][2 * num_gselect] selections
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, nw = blockDim.x / kWave;
  const int64_t t = (int64_t)blockIdx.x * nw + wave;
  float *ll = s_ll + (size_t)wave * G; float *sel_p = s_ll + (size_t)nw * G + (size_t)wave * 2 * num_gselect; int *sel_g = (int *)(sel_p + num_gselect);
  if (t >= total_frames) return;                              // whole waves leave together; no block barrier below
  // frame weights (OnlineIvectorFeature::UpdateStatsForFrames, online-ivector-feature.cc:226-236): a frame of weight 0 has no posteriors; otherwise the pruning threshold is
  // GetMinPost(weight) = min(0.99, min_post / |weight|) (:188-199) and the posteriors are scaled by posterior_scale * weight
  if (frame_w) {
    const float w = frame_w[t];
    if (w == 0.0f) { if (lane == 0) post_n[t] = 0; return; }
    min_post = fminf(min_post / fabsf(w), 0.99f); post_scale *= w;
  }
  const float *xt = x + t * D;
  float mx = -INFINITY;
  for (int g = lane; g < G; g += kWave) {
    float a1 = 0.f, a2 = 0.f;
    for (int d = 0; d < D; d++) { const float v = xt[d]; a1 = fmaf(miv_t[(size_t)d * G + g], v, a1); a2 = fmaf(iv_t[(size_t)d * G + g], v * v, a2); }
    const float l = gconsts[g] + a1 - 0.5f * a2; ll[g] = l; mx = fmaxf(mx, l);
  }
  for (int o = kWave / 2; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  const float cut = min_post != 0.f ? mx + logf(min_post) : -INFINITY;
  int n = 0;
  for (int k = 0; k < num_gselect; k++) {
    float bl = -INFINITY; int bg = 0x7fffffff;
    for (int g = lane; g < G; g += kWave) { const float l = ll[g]; if (l > cut && (l > bl)) { bl = l; bg = g; } }      // lowest g among equal values of a lane
    for (int o = kWave / 2; o; o >>= 1) {
      const float ol = __shfl_xor(bl, o); const int og = __shfl_xor(bg, o);
      if (ol > bl || (ol == bl && og < bg)) { bl = ol; bg = og; }
    }
    if (bg == 0x7fffffff) break;
    if (lane == 0) { sel_p[n] = expf(bl - mx); sel_g[n] = bg; }
    if (bg % kWave == lane) ll[bg] = -INFINITY;             // its owner lane retires it
    n++; __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    float tot = 0.f; for (int k = 0; k < n; k++) tot += sel_p[k];
    const float cutoff = min_post * tot;
    while (n > 1 && sel_p[n - 1] < cutoff) { tot -= sel_p[n - 1]; n--; }
    const float inv = 1.0f / tot;
    int64_t tt = t;
    // batched streaming: the stream's own arrays
    if (segs) {
      const BatchSeg &g = segs[seg_of_row(segs, nseg, t)];
      tt = t - g.out_off;
      post_g = g.pg;
      post_w = g.pw;
      post_n = g.pn;
    }
    for (int k = 0; k < n; k++) { post_g[tt * num_gselect + k] = sel_g[k]; post_w[tt * num_gselect + k] = (sel_p[k] * inv) * post_scale; }
    post_n[tt] = n;
  }
}

// ---------------------------------------------------------------------------------------------------------------- statistics + solution
__device__ __forceinline__ double block_sum(double v, double *s_red) {          // all threads get the sum; fixed order, so every run gives the same bits
  for (int o = kWave / 2; o; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if (threadIdx.x % kWave == 0) s_red[threadIdx.x / kWave] = v;
  __syncthreads();
  double s = 0; for (int w = 0; w < kBlock / kWave; w++) s += s_red[w];
  return s;
}

// y = A x for symmetric A [R][R]: thread i reads column i (= row i), consecutive threads consecutive addresses
__device__ __forceinline__ double sym_row_dot(const double *A, const double *x, int R, int i) {
  double a = 0; for (int j = 0; j < R; j++) a += A[(size_t)j * R + i] * x[j]; return a;
}

// In-place Cholesky solve of A x = b on a scratch copy C of A (lower triangle used); "the exact optimisation" of LinearCgd's fall-back
// (matrix/optimization.cc:545-556) and the solution method of the GPU reference.  C may be any memory the whole block sees.
__device__ void chol_solve(double *C, const double *b, double *x, double *y, int R) {
  for (int k = 0; k < R; k++) {
    __syncthreads();
    if (threadIdx.x == 0) C[(size_t)k * R + k] = sqrt(C[(size_t)k * R + k]);
    __syncthreads();
    const double dkk = C[(size_t)k * R + k];
    for (int i = k + 1 + threadIdx.x; i < R; i += kBlock) C[(size_t)i * R + k] /= dkk;
    __syncthreads();
    const int m = R - k - 1;                                    // trailing update of the lower triangle
    for (int e = threadIdx.x; e < m * m; e += kBlock) {
      const int i = k + 1 + e / m, j = k + 1 + e % m;
      if (j <= i) C[(size_t)i * R + j] -= C[(size_t)i * R + k] * C[(size_t)j * R + k];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {                                       // R <= a few hundred: two serial triangular solves
    for (int i = 0; i < R; i++) { double a = b[i]; for (int j = 0; j < i; j++) a -= C[(size_t)i * R + j] * y[j]; y[i] = a / C[(size_t)i * R + i]; }
    for (int i = R - 1; i >= 0; i--) { double a = y[i]; for (int j = i + 1; j < R; j++) a -= C[(size_t)j * R + i] * x[j]; x[i] = a / C[(size_t)i * R + i]; }
  }
  __syncthreads();
}

struct EstParams {
  const float *xstats; const int64_t *frame_off; const int32_t *post_g; const float *post_w; const int32_t *post_n;
  const double *U, *SM; double *quad_g, *chol_g;              // per utterance [R][R] scratch (quad_g NULL = LDS)
  float *out; int64_t ld_out; const int64_t *out_off;
  int D, R, S, period, num_cg_iters, exact_solve; double prior, max_count;
  int acc_tail; const double *state_in; double *state_out;      // per utterance [1 + R + R*R]: num_frames, linear, quadratic (OnlineIvectorEstimationStats), nullable
  // streaming (k3_ivector_stream): the posterior-stage arrays start at frame t_base of the stream; estimates k_begin, k_begin + 1, ... while k * period < t_limit; x_io [R] = the
  // conjugate-gradient start (the previous estimate) in, the last estimate out.  t_limit < 0: a whole utterance (all of the above off)
  int64_t t_base, t_limit; int k_begin; double *x_io;
  const BatchSeg *segs;      // batched streaming: block u works on segs[u] (its arrays, state, scratch, range and output replace the fields above); null otherwise
};

__global__ void __launch_bounds__(kBlock) ivec_estimate_kernel(EstParams p_in) {
  extern __shared__ double s_dyn[];
  EstParams p = p_in; const int u = blockIdx.x; const BatchSeg *seg = p.segs ? p.segs + u : nullptr;
  if (seg) {
    if (seg->nk == 0) return;      // (the whole block: no barrier passed yet)
    p.xstats = seg->e_px; p.post_g = seg->e_pg; p.post_w = seg->e_pw; p.post_n = seg->e_pn; p.state_in = seg->est; p.state_out = seg->est; p.x_io = seg->x_io;
    p.t_base = seg->t_base; p.t_limit = seg->t_limit; p.k_begin = seg->k_begin; p.out = seg->rows; p.ld_out = p.R;
  }
  const int R = p.R, D = p.D, S = p.S, P = p.period, tid = threadIdx.x;
  double *s_x = s_dyn, *s_r = s_x + R, *s_p = s_r + R, *s_ap = s_p + R, *s_lin = s_ap + R, *s_xo = s_lin + R, *s_red = s_xo + R;    // 6R + 4 doubles
  const int max_ent = P * S;
  // gw != 0 marks a leader (weights may be negative)
  int *e_g = (int *)(s_red + kBlock / kWave);
  int *e_t = e_g + max_ent;
  float *e_w = (float *)(e_t + max_ent);
  float *e_gw = e_w + max_ent;
  double *A = seg ? (seg->quad ? seg->quad : (double *)(((uintptr_t)(e_gw + max_ent) + 7) & ~(uintptr_t)7)) : p.quad_g ? p.quad_g + (size_t)u * R * R :
      (double *)(((uintptr_t)(e_gw + max_ent) + 7) & ~(uintptr_t)7);
  double *C = seg ? seg->chol : p.chol_g + (size_t)u * R * R;
  __shared__ int s_n; __shared__ double s_tot;
  const int64_t fb = seg ? -p.t_base : p.frame_off[u] - (p.t_limit >= 0 ? p.t_base : 0); const int T = p.t_limit >= 0 ? (int)p.t_limit : (int)(p.frame_off[u + 1] - p.frame_off[u]);
  const int64_t out_base = seg ? 0 : p.out_off[u]; const size_t st_stride = seg ? 0 : (1 + R + (size_t)R * R);
  const double *st_in = p.state_in ? p.state_in + (size_t)u * st_stride : nullptr;      // the speaker's statistics so far (SetAdaptationState)
  // fresh: quadratic term of the prior I, linear term prior_offset e_0
  for (int i = tid; i < R * R; i += kBlock) A[i] = st_in ? st_in[1 + R + i] : ((i / R == i % R) ? 1.0 : 0.0);
  if (tid < R) { s_lin[tid] = st_in ? st_in[1 + tid] : (tid == 0 ? p.prior : 0.0); s_x[tid] = p.x_io ? p.x_io[tid] : (tid == 0 ? p.prior : 0.0); }
  double nframes = st_in ? st_in[0] : 0.0;
  __syncthreads();
  // OnlineIvectorFeature::UpdateStatsUntilFrame (online-ivector-feature.cc:248-277) for frames t_lo .. t_hi: every thread of the block calls it
  auto accumulate = [&](int t_lo, int t_hi) {
    if (tid == 0) {                                                                   // entries in frame order, then in the order VectorToPosteriorEntry left them
      int n = 0;
      for (int t = t_lo; t <= t_hi; t++) {
        const int c = p.post_n[fb + t];
        for (int j = 0; j < c; j++) {
          e_g[n] = p.post_g[(fb + t) * S + j];
          e_w[n] = p.post_w[(fb + t) * S + j];
          e_t[n] = t;
          n++;
        }
      }
      s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    for (int e = tid; e < n; e += kBlock) {                                           // total weight of a Gaussian: a float sum in frame order, like AccStats' Vector<BaseFloat>
      bool leader = true; for (int f = 0; f < e; f++) if (e_g[f] == e_g[e]) { leader = false; break; }
      float gw = 0.f; if (leader) for (int f = e; f < n; f++) if (e_g[f] == e_g[e]) gw += e_w[f];
      e_gw[e] = leader ? gw : 0.f;
    }
    __syncthreads();
    if (tid == 0) { double tot = 0; for (int e = 0; e < n; e++) tot += (double)e_gw[e]; s_tot = tot; }
    if (tid < R) {                                                                    // linear += sum_e w_e SM[g_e]^T x_stats[t_e]
      double a = 0;
      for (int e = 0; e < n; e++) {
        const double *sm = p.SM + (size_t)e_g[e] * D * R + tid; const float *x = p.xstats + (fb + e_t[e]) * D; double b = 0;
        for (int d = 0; d < D; d++) b += sm[(size_t)d * R] * (double)x[d];
        a += (double)e_w[e] * b;
      }
      s_lin[tid] += a;
    }
    for (int i = tid; i < R * R; i += kBlock) {                                       // quad += sum_g gw_g U_g
      double a = 0; for (int e = 0; e < n; e++) if (e_gw[e] != 0.f) a += (double)e_gw[e] * p.U[(size_t)e_g[e] * R * R + i];
      A[i] += a;
    }
    __syncthreads();
    const double tot = s_tot;
    if (p.max_count > 0.0) {                                                          // --max-count: the prior grows with the count beyond max_count (AccStats :650-668)
      const double change = fmax(nframes + tot, p.max_count) / p.max_count - fmax(nframes, p.max_count) / p.max_count;
      if (change != 0.0) { if (tid == 0) s_lin[0] += p.prior * change; if (tid < R) A[(size_t)tid * R + tid] += change; }
    }
    nframes += tot;
    __syncthreads();
  };
  int k_last = -1;
  const int k0 = p.t_limit >= 0 ? p.k_begin : 0;
  for (int k = k0; (int64_t)k * P < T; k++) {
    const int t_lo = k == 0 ? 0 : (k - 1) * P + 1, t_hi = k * P;                     // frames not yet in the statistics, up to and including frame k*P
    accumulate(t_lo, t_hi); k_last = k;
    if (nframes > 0.0) {
      if (tid == 0 && s_x[0] == 0.0) s_x[0] = p.prior;
      __syncthreads();
      if (p.exact_solve) {
        for (int i = tid; i < R * R; i += kBlock) C[i] = A[i];
        chol_solve(C, s_lin, s_x, s_ap, R);
      } else {                                                                        // LinearCgd<double>, defaults of LinearCgdOptions, warm start s_x
        double v = tid < R ? s_lin[tid] - sym_row_dot(A, s_x, R, tid) : 0.0;
        if (tid < R) { s_p[tid] = v; s_r[tid] = -v; s_xo[tid] = s_x[tid]; }
        double r_cur = block_sum(v * v, s_red); const double r_init = r_cur; double r_rec = r_cur; const double rf = 0.01 * 0.01;
        const int max_it = p.num_cg_iters;
        for (int it = 0; it < R + 5 && it != max_it; it++) {
          const double ap = tid < R ? sym_row_dot(A, s_p, R, tid) : 0.0;
          if (tid < R) s_ap[tid] = ap;
          const double pr = block_sum(tid < R ? s_p[tid] * s_r[tid] : 0.0, s_red), pap = block_sum(tid < R ? s_p[tid] * ap : 0.0, s_red);
          const double alpha = -pr / pap;
          double rn = 0;
          if (tid < R) { s_x[tid] += alpha * s_p[tid]; rn = s_r[tid] + alpha * ap; s_r[tid] = rn; }
          double r_next = block_sum(rn * rn, s_red);
          if (r_next < rf * r_rec || r_next > r_rec / rf) {                           // recompute the residual from scratch
            rn = tid < R ? sym_row_dot(A, s_x, R, tid) - s_lin[tid] : 0.0;
            __syncthreads(); if (tid < R) s_r[tid] = rn;
            r_next = block_sum(rn * rn, s_red); r_rec = r_next;
          }
          if (r_next <= 2.2250738585072014e-308) break;
          const double beta = r_next / r_cur;
          __syncthreads(); if (tid < R) s_p[tid] = beta * s_p[tid] - s_r[tid];
          __syncthreads();
          r_cur = r_next;
        }
        const double bb = block_sum(tid < R ? s_lin[tid] * s_lin[tid] : 0.0, s_red);
        if (r_cur > r_init && r_cur > r_init + 1.0e-10 * bb) {                        // the squared residual got worse: exact optimisation
          for (int i = tid; i < R * R; i += kBlock) C[i] = A[i];
          chol_solve(C, s_lin, s_x, s_ap, R);
        }
      }
    } else if (tid < R) s_x[tid] = tid == 0 ? p.prior : 0.0;
    __syncthreads();
    if (tid < R) {
      const float v = tid == 0 ? (float)s_x[0] - (float)p.prior : (float)s_x[tid];
      p.out[(out_base + k - k0) * p.ld_out + tid] = v;
      if (seg) seg->latest[tid] = v;      // (the last estimate of the call stays)
    }
  }
  if (p.x_io) { __syncthreads(); if (tid < R) p.x_io[tid] = s_x[tid]; }
  if (p.state_out) {                                                                  // GetAdaptationState: the statistics as they stand after the last estimate
    // (accumulate_tail: the reference's --repeat=true asks for the i-vector of the LAST frame, so its statistics hold every frame of the utterance when the state is taken:
    // ivector-extract-online2.cc:121-127 GetFrame(T - 1) -> UpdateStatsUntilFrame(T - 1))
    if (p.acc_tail && k_last >= 0 && k_last * P + 1 <= T - 1) accumulate(k_last * P + 1, T - 1);
    double *so = p.state_out + (size_t)u * st_stride;
    __syncthreads();
    if (tid == 0) so[0] = nframes;
    if (tid < R) so[1 + tid] = s_lin[tid];
    for (int i = tid; i < R * R; i += kBlock) so[1 + R + i] = A[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------- C ABI
extern "C" void k3_ivector_opts_default(k3_ivector_opts *o) {
  if (!o) return;
  o->left_context = 0; o->right_context = 0; o->num_gselect = 5; o->min_post = 0.025f; o->posterior_scale = 0.1f; o->max_count = 0.f;
  o->ivector_period = 10; o->num_cg_iters = 15; o->exact_solve = 0; o->online_cmvn_iextractor = 0; k3_online_cmvn_opts_default(&o->cmvn);
}

template <typename T> static int upload(T **dst, const std::vector<T> &h) {
  K3_HIP_CHECK(hipMalloc((void **)dst, h.size() * sizeof(T)));
  K3_HIP_CHECK(hipMemcpy(*dst, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return K3_OK;
}

extern "C" int k3_ivector_create(const k3_ivector_model *m, const k3_ivector_opts *opts, k3_ivector **out) {
  K3_REQUIRE(m && opts && out, "k3_ivector_create: null argument");
  K3_REQUIRE(m->feat_dim > 0 && m->lda && m->lda_rows > 0 && m->global_cmvn_stats && m->num_gauss > 0 && m->gconsts && m->means_invvars && m->inv_vars &&
      m->ivector_dim > 0 && m->M && m->sigma_inv,
             "k3_ivector_create: incomplete model");
  K3_REQUIRE(m->ivector_dim <= kBlock, "k3_ivector_create: i-vector dimension above 256");
  const int spl = opts->left_context + opts->right_context + 1;
  K3_REQUIRE(opts->left_context >= 0 && opts->right_context >= 0, "k3_ivector_create: negative splice context");
  // OnlineTransform, feat/online-feature.cc:520-535
  K3_REQUIRE(m->lda_cols == m->feat_dim * spl || m->lda_cols == m->feat_dim * spl + 1,
      "k3_ivector_create: the LDA matrix does not match the spliced feature dimension");
  // OnlineIvectorExtractionInfo::Check :100-121
  K3_REQUIRE(opts->num_gselect > 0 && opts->ivector_period > 0 && opts->min_post >= 0.f && opts->min_post < 1.f, "k3_ivector_create: bad option value");
  K3_REQUIRE(opts->posterior_scale > 0.f && opts->posterior_scale <= 1.f && opts->max_count >= 0.f, "k3_ivector_create: bad posterior-scale / max-count");
  std::unique_ptr<k3_ivector> iv(new k3_ivector);
  iv->o = *opts; iv->F = m->feat_dim; iv->D = m->lda_rows; iv->G = m->num_gauss; iv->R = m->ivector_dim; iv->splice = spl; iv->has_offset = m->lda_cols == m->feat_dim * spl + 1;
  iv->prior_offset = m->prior_offset;
  const int F = iv->F, D = iv->D, G = iv->G, R = iv->R;
  { std::vector<float> h(m->lda, m->lda + (size_t)D * m->lda_cols); const int rc = upload(&iv->lda, h); if (rc) return rc; }
  K3_REQUIRE(m->global_cmvn_stats[F] > 0.0, "k3_ivector_create: the global CMVN statistics hold no frames (OnlineCmvn raises 'Global CMVN stats are required')");
  { std::vector<double> h(m->global_cmvn_stats, m->global_cmvn_stats + 2 * (size_t)(F + 1)); const int rc = upload(&iv->global_stats, h); if (rc) return rc; }
  { std::vector<float> gc(G), a((size_t)D * G), b((size_t)D * G);
    for (int g = 0; g < G; g++) {
      gc[g] = (float)m->gconsts[g];
      for (int d = 0; d < D; d++) {
        a[(size_t)d * G + g] = (float)m->means_invvars[(size_t)g * D + d];
        b[(size_t)d * G + g] = (float)m->inv_vars[(size_t)g * D + d];
      }
    }
    int rc = upload(&iv->gconsts, gc); if (!rc) rc = upload(&iv->miv_t, a); if (!rc) rc = upload(&iv->iv_t, b); if (rc) return rc; }
  { double *dM = nullptr, *dS = nullptr;
    std::vector<double> hM(m->M, m->M + (size_t)G * D * R), hS(m->sigma_inv, m->sigma_inv + (size_t)G * (D * (D + 1) / 2));
    int rc = upload(&dM, hM); if (!rc) rc = upload(&dS, hS);
    if (!rc && hipMalloc((void **)&iv->SM, (size_t)G * D * R * 8) != hipSuccess) rc = K3_ERR_HIP;
    if (!rc && hipMalloc((void **)&iv->U, (size_t)G * R * R * 8) != hipSuccess) rc = K3_ERR_HIP;
    if (!rc) {
      const long n1 = (long)G * D * R, n2 = (long)G * R * R;
      hipLaunchKernelGGL(ivec_sigma_inv_m, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, 0, dM, dS, iv->SM, G, D, R);
      hipLaunchKernelGGL(ivec_u, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, dM, iv->SM, iv->U, G, D, R);
      if (hipDeviceSynchronize() != hipSuccess) rc = K3_ERR_HIP;
    }
    if (dM) (void)hipFree(dM);
    if (dS) (void)hipFree(dS);
    if (rc) { k3::set_error("k3_ivector_create: device allocation or the derived-variable kernels failed"); return rc; } }
  iv->quad_in_lds = (size_t)R * R * 8 <= 128 * 1024 && !getenv("K3_IVECTOR_QUAD_IN_HBM");
  *out = iv.release(); return K3_OK;
}
extern "C" void k3_ivector_destroy(k3_ivector *iv) { delete iv; }
extern "C" int k3_ivector_get_info(const k3_ivector *iv, k3_ivector_info *info) {
  K3_REQUIRE(iv && info, "k3_ivector_get_info: null argument");
  info->feat_dim = iv->F; info->lda_dim = iv->D; info->num_gauss = iv->G; info->ivector_dim = iv->R; info->ivector_period = iv->o.ivector_period; return K3_OK;
}
extern "C" int64_t k3_ivector_num_rows(const k3_ivector *iv, int32_t num_utts, const int64_t *h_frame_offsets, int64_t *h_row_offsets) {
  if (!iv || !h_frame_offsets || num_utts < 0) return -1;
  int64_t n = 0; const int P = iv->o.ivector_period;
  for (int u = 0; u < num_utts; u++) { if (h_row_offsets) h_row_offsets[u] = n; n += (h_frame_offsets[u + 1] - h_frame_offsets[u] + P - 1) / P; }
  if (h_row_offsets) h_row_offsets[num_utts] = n;
  return n;
}

extern "C" void k3_ivector_set_accumulate_tail(k3_ivector *iv, int32_t on) { if (iv) iv->acc_tail = on ? 1 : 0; }
extern "C" int k3_ivector_extract_batch_adapt(k3_ivector *iv, const float *d_feats, int64_t ld_feats, const int64_t *h_frame_offsets, int32_t num_utts,
    float *d_ivectors, int64_t ld_ivectors,
                                              const double *d_cmvn_speaker_stats, const double *d_stats_in, double *d_stats_out, void *stream_);
extern "C" int k3_ivector_extract_batch(k3_ivector *iv, const float *d_feats, int64_t ld_feats, const int64_t *h_frame_offsets, int32_t num_utts,
    float *d_ivectors, int64_t ld_ivectors,
                                        void *stream_) {
  return k3_ivector_extract_batch_adapt(iv, d_feats, ld_feats, h_frame_offsets, num_utts, d_ivectors, ld_ivectors, nullptr, nullptr, nullptr, stream_);
}
extern "C" int64_t k3_ivector_stats_size(const k3_ivector *iv) { return iv ? 1 + iv->R + (int64_t)iv->R * iv->R : -1; }
extern "C" int k3_ivector_extract_batch_adapt(k3_ivector *iv, const float *d_feats, int64_t ld_feats, const int64_t *h_frame_offsets, int32_t num_utts,
    float *d_ivectors, int64_t ld_ivectors,
                                              const double *d_cmvn_speaker_stats, const double *d_stats_in, double *d_stats_out, void *stream_) {
  return k3_ivector_extract_batch_weighted(iv, d_feats, ld_feats, h_frame_offsets, num_utts, nullptr, d_ivectors, ld_ivectors, d_cmvn_speaker_stats,
      d_stats_in, d_stats_out, stream_);
}
extern "C" int k3_ivector_extract_batch_weighted(k3_ivector *iv, const float *d_feats, int64_t ld_feats, const int64_t *h_frame_offsets, int32_t num_utts,
    const float *d_frame_weights,
                                                 float *d_ivectors, int64_t ld_ivectors, const double *d_cmvn_speaker_stats, const double *d_stats_in,
                                                     double *d_stats_out, void *stream_) {
  K3_REQUIRE(iv && d_feats && h_frame_offsets && d_ivectors && num_utts > 0, "k3_ivector_extract_batch: null or empty argument");
  K3_REQUIRE(ld_feats >= iv->F && ld_ivectors >= iv->R, "k3_ivector_extract_batch: leading dimension smaller than the row length");
  hipStream_t stream = (hipStream_t)stream_;
  const int F = iv->F, D = iv->D, G = iv->G, R = iv->R, S = iv->o.num_gselect, P = iv->o.ivector_period; const int64_t N = h_frame_offsets[num_utts] - h_frame_offsets[0];
  K3_REQUIRE(h_frame_offsets[0] == 0, "k3_ivector_extract_batch: frame offsets must start at 0");
  // the reference writes no i-vector for an empty utterance
  for (int u = 0; u < num_utts; u++) K3_REQUIRE(h_frame_offsets[u + 1] > h_frame_offsets[u], "k3_ivector_extract_batch: an utterance without frames");
  std::vector<int64_t> offs(2 * (size_t)(num_utts + 1));
  for (int u = 0; u <= num_utts; u++) offs[u] = h_frame_offsets[u];
  k3_ivector_num_rows(iv, num_utts, h_frame_offsets, offs.data() + num_utts + 1);
  int rc = iv->frame_off.reserve(offs.size() * 8); if (rc) return rc;
  if ((rc = iv->cmvn.reserve((size_t)N * F * 4)) || (rc = iv->xpost.reserve((size_t)N * D * 4)) || (rc = iv->xstats.reserve((size_t)N * D * 4)) ||
      (rc = iv->post_g.reserve((size_t)N * S * 4)) ||
      (rc = iv->post_w.reserve((size_t)N * S * 4)) || (rc = iv->post_n.reserve((size_t)N * 4)) || (rc = iv->state.reserve((size_t)num_utts * R * R * 8)))
    return rc;
  if (!iv->quad_in_lds && (rc = iv->quad.reserve((size_t)num_utts * R * R * 8))) return rc;
  K3_HIP_CHECK(hipMemcpyAsync(iv->frame_off.p, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, stream));
  K3_HIP_CHECK(hipStreamSynchronize(stream));                   // offs is a local
  const int64_t *d_off = (const int64_t *)iv->frame_off.p, *d_row_off = d_off + num_utts + 1;
  rc = k3_cmvn_online_batch(d_feats, ld_feats, (float *)iv->cmvn.p, F, F, d_off, num_utts, &iv->o.cmvn, iv->global_stats, d_cmvn_speaker_stats, nullptr, 0,
      stream_);
  if (rc) return rc;
  const unsigned nb = (unsigned)((N * D + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(ivec_splice_lda_kernel, dim3(nb), dim3(kBlock), 0, stream, (const float *)iv->cmvn.p, (int64_t)F, d_off, num_utts, iv->lda,
      F * iv->splice + iv->has_offset, iv->has_offset, F, D,
                     iv->o.left_context, iv->o.right_context, (float *)iv->xpost.p, N, (int64_t)0);
  hipLaunchKernelGGL(ivec_splice_lda_kernel, dim3(nb), dim3(kBlock), 0, stream, iv->o.online_cmvn_iextractor ? (const float *)iv->cmvn.p : d_feats,
      iv->o.online_cmvn_iextractor ? (int64_t)F : ld_feats, d_off, num_utts, iv->lda, F * iv->splice + iv->has_offset, iv->has_offset, F, D,
                     iv->o.left_context, iv->o.right_context, (float *)iv->xstats.p, N, (int64_t)0);
  const int nw = kBlock / kWave; const size_t lds_post = ((size_t)nw * G + (size_t)nw * 2 * S) * 4;
  K3_REQUIRE(lds_post <= 64 * 1024, "k3_ivector_extract_batch: too many Gaussians for the posterior kernel's LDS tile");
  const float min_post = iv->o.min_post < 0.99f ? iv->o.min_post : 0.99f;       // GetMinPost caps it, online-ivector-feature.cc:188-199
  hipLaunchKernelGGL(ivec_posterior_kernel, dim3((unsigned)((N + nw - 1) / nw)), dim3(kBlock), lds_post, stream, (const float *)iv->xpost.p, D, G, iv->gconsts,
      iv->miv_t, iv->iv_t, S, min_post,
                     iv->o.posterior_scale, N, (int32_t *)iv->post_g.p, (float *)iv->post_w.p, (int32_t *)iv->post_n.p, (const BatchSeg *)nullptr, 0, d_frame_weights);
  EstParams p;
  p.xstats = (const float *)iv->xstats.p;
  p.frame_off = d_off;
  p.post_g = (const int32_t *)iv->post_g.p;
  p.post_w = (const float *)iv->post_w.p;
  p.post_n = (const int32_t *)iv->post_n.p;
  p.U = iv->U;
  p.SM = iv->SM;
  p.quad_g = iv->quad_in_lds ? nullptr : (double *)iv->quad.p;
  p.chol_g = (double *)iv->state.p;
  p.out = d_ivectors;
  p.ld_out = ld_ivectors;
  p.out_off = d_row_off;
  p.D = D;
  p.R = R;
  p.S = S;
  p.period = P;
  p.num_cg_iters = iv->o.num_cg_iters;
  p.exact_solve = iv->o.exact_solve;
  p.prior = iv->prior_offset;
  p.max_count = iv->o.max_count;
  p.state_in = d_stats_in;
  p.state_out = d_stats_out;
  p.acc_tail = iv->acc_tail;
  p.t_base = 0;
  p.t_limit = -1;
  p.k_begin = 0;
  p.x_io = nullptr;
  p.segs = nullptr;
  size_t lds_est = (size_t)(6 * R + kBlock / kWave) * 8 + (size_t)P * S * 16 + 8 + (iv->quad_in_lds ? (size_t)R * R * 8 : 0);
  K3_REQUIRE(lds_est <= 156 * 1024, "k3_ivector_extract_batch: ivector_period * num_gselect too large for the estimation kernel's LDS");
  K3_HIP_CHECK(hipFuncSetAttribute((const void *)ivec_estimate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_est));
  hipLaunchKernelGGL(ivec_estimate_kernel, dim3((unsigned)num_utts), dim3(kBlock), lds_est, stream, p);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}

// ---------------------------------------------------------------------------------------------------------------- streaming
// One stream's extractor state between chunks: what OnlineIvectorFeature keeps (online2/online-ivector-feature.h:233-330: the statistics ivector_stats_, the
// frames already in them,
// the estimates made so far, and under it OnlineCmvn's window and OnlineSpliceFrames' context) and what BatchedIvectorExtractorCuda keeps per channel
// (cudafeat/feature-online-batched-ivector-cuda.h:30-61).  Every frame passes each stage once -- CMVN recursion continued from its carried window sums, splice + LDA + posteriors
// when the frame's right context exists (or the stream has ended), statistics and one estimate per period -- in the order and arithmetic of the whole-utterance
// kernels above, so the
// rows are bit-identical to k3_ivector_extract_batch on the whole utterance (tests/test_ivector_gpu.py).  Device memory per stream: max(cmn_window, splice) raw frames, the splice
// context of normalised frames, the posterior-stage output of at most one period (+ a chunk), and (3 F + 1 + 2 R + R^2) doubles.
struct k3_ivector_stream {
  k3_ivector *iv = nullptr;
  DevBuf raw[2], cm[2], px[2], pg[2], pw[2], pn[2], xpost, rows, state, offs, latest, chol, quad;
  int raw_i = 0, cm_i = 0, pend_i = 0;
  // frames accepted; stream index of row 0 of raw / cm / the pending posterior arrays; frames with posteriors; estimates made
  int64_t n_abs = 0, raw_s0 = 0, cm_c0 = 0, n_post = 0, a0 = 0, k_next = 0;
  bool finished = false, fresh = false;
  // a batched call failed after it had begun to move this stream's state (buffer switches, counters): the host state and the device buffers may be out of step
  // -- k3_ivector_stream_reset
  bool poisoned = false;
};

extern "C" int k3_ivector_stream_create(k3_ivector *iv, k3_ivector_stream **out) {
  K3_REQUIRE(iv && out, "k3_ivector_stream_create: null argument");
  k3_ivector_stream *s = new k3_ivector_stream(); s->iv = iv;
  const int rc = k3_ivector_stream_reset(s, nullptr);
  if (rc) { delete s; return rc; }
  *out = s; return K3_OK;
}
extern "C" void k3_ivector_stream_destroy(k3_ivector_stream *s) { delete s; }
extern "C" int64_t k3_ivector_stream_num_rows(const k3_ivector_stream *s) { return s ? s->k_next : -1; }

extern "C" int k3_ivector_stream_reset(k3_ivector_stream *s, void *stream_) {
  K3_REQUIRE(s, "k3_ivector_stream_reset: null stream");
  const k3_ivector *iv = s->iv; const int F = iv->F, R = iv->R; hipStream_t stream = (hipStream_t)stream_;
  // window sums 0; statistics of no frames: quadratic term I, linear term prior_offset e_0 (OnlineIvectorEstimationStats), estimate prior_offset e_0
  // (online-ivector-feature.cc:381-385)
  std::vector<double> h((size_t)3 * F + 1 + R + (size_t)R * R + R, 0.0);
  double *est = h.data() + 3 * F; est[1] = iv->prior_offset; for (int i = 0; i < R; i++) est[1 + R + (size_t)i * R + i] = 1.0; est[1 + R + (size_t)R * R] = iv->prior_offset;
  int rc = s->state.reserve(h.size() * 8); if (rc) return rc;
  if ((rc = s->latest.reserve((size_t)R * 4)) || (rc = s->chol.reserve((size_t)R * R * 8)) || (rc = s->offs.reserve(8 * 8))) return rc;
  if (!iv->quad_in_lds && (rc = s->quad.reserve((size_t)R * R * 8))) return rc;
  K3_HIP_CHECK(hipMemcpyAsync(s->state.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, stream));
  K3_HIP_CHECK(hipMemsetAsync(s->latest.p, 0, (size_t)R * 4, stream));
  K3_HIP_CHECK(hipStreamSynchronize(stream));                   // h is a local
  s->n_abs = s->raw_s0 = s->cm_c0 = s->n_post = s->a0 = s->k_next = 0; s->finished = false; s->fresh = true; s->poisoned = false;
  return K3_OK;
}

namespace {
// the other buffer of a pair becomes [rows keep_from .. keep_from + keep_rows of the current one | room up to total_rows]
int carry_over(DevBuf (&b)[2], int *cur, size_t row_bytes, int64_t keep_from, int64_t keep_rows, int64_t total_rows, hipStream_t stream) {
  DevBuf &dst = b[*cur ^ 1];
  const size_t need = (size_t)std::max<int64_t>(total_rows, 1) * row_bytes;
  if (need > dst.cap) { const int rc = dst.reserve(need + need / 2); if (rc) return rc; }
  if (keep_rows > 0) K3_HIP_CHECK(hipMemcpyAsync(dst.p, (const char *)b[*cur].p + (size_t)keep_from * row_bytes, (size_t)keep_rows * row_bytes, hipMemcpyDeviceToDevice, stream));
  *cur ^= 1; return K3_OK;
}
}  // namespace

namespace {
__global__ void ivec_set_offsets_kernel(int64_t *o, int64_t raw_rows, int64_t cm_rows, int64_t keep) {
  o[0] = 0;
  o[1] = raw_rows;
  o[2] = 0;
  o[3] = cm_rows;
  o[4] = keep;
  o[5] = 0;
  o[6] = 0;
  o[7] = 0;
}
}  // namespace

extern "C" int k3_ivector_stream_accept(k3_ivector_stream *s, const float *d_feats, int64_t ld_feats, int32_t num_frames, int32_t finished, float *d_new_rows, int64_t ld_rows,
                                        int32_t max_new_rows, int32_t *h_num_new_rows, float *d_latest, void *stream_) {
  K3_REQUIRE(s && num_frames >= 0 && (num_frames == 0 || (d_feats && ld_feats >= s->iv->F)), "k3_ivector_stream_accept: bad argument");
  K3_REQUIRE(!s->finished, "k3_ivector_stream_accept: the stream has ended (k3_ivector_stream_reset starts the next one)");
  K3_REQUIRE(!s->poisoned, "k3_ivector_stream_accept: an earlier batched call on this stream failed half-way, its state is lost (k3_ivector_stream_reset starts over)");
  K3_REQUIRE(!d_new_rows || ld_rows >= s->iv->R, "k3_ivector_stream_accept: leading dimension of the rows smaller than the i-vector dimension");
  k3_ivector *iv = s->iv; hipStream_t stream = (hipStream_t)stream_;
  const int F = iv->F, D = iv->D, G = iv->G, R = iv->R, S = iv->o.num_gselect, P = iv->o.ivector_period, lc = iv->o.left_context, rc_ = iv->o.right_context,
      W = iv->o.cmvn.cmn_window;
  const int64_t n_new = num_frames, keepN = std::max<int64_t>(W, (int64_t)lc + rc_ + 1);
  int rc;
  // ---- raw frames: [what the CMVN window and the splice still read | the new frames]
  const int64_t new_s0 = std::max<int64_t>(0, s->n_abs - keepN), keep = s->n_abs - new_s0, raw_rows = keep + n_new;
  if ((rc = carry_over(s->raw, &s->raw_i, (size_t)F * 4, new_s0 - s->raw_s0, keep, raw_rows, stream))) return rc;
  s->raw_s0 = new_s0;
  float *raw = (float *)s->raw[s->raw_i].p;
  if (n_new > 0) K3_HIP_CHECK(hipMemcpy2DAsync(raw + keep * F, (size_t)F * 4, d_feats, (size_t)ld_feats * 4, (size_t)F * 4, (size_t)n_new, hipMemcpyDeviceToDevice, stream));
  // ---- normalised frames: [the splice context of the frames without posteriors yet | the new frames]
  const int64_t new_c0 = std::max<int64_t>(0, s->n_post - lc), cm_keep = s->n_abs - new_c0, cm_rows = cm_keep + n_new;
  if ((rc = carry_over(s->cm, &s->cm_i, (size_t)F * 4, new_c0 - s->cm_c0, cm_keep, cm_rows, stream))) return rc;
  s->cm_c0 = new_c0;
  float *cm = (float *)s->cm[s->cm_i].p;
  // (values travel as kernel arguments: nothing on the host to keep alive, no wait)
  hipLaunchKernelGGL(ivec_set_offsets_kernel, dim3(1), dim3(1), 0, stream, (int64_t *)s->offs.p, raw_rows, cm_rows, keep);
  const int64_t *d_offs = (const int64_t *)s->offs.p;
  double *carry = (double *)s->state.p, *est = carry + 3 * F, *x_io = est + 1 + R + (size_t)R * R;
  if (n_new > 0) {      // rows keep .. raw_rows - 1 of raw -> rows cm_keep .. of cm
    rc = k3::cmvn_online_resume_async(raw, F, cm + (cm_keep - keep) * F, F, F, (const long long *)d_offs, 1, &iv->o.cmvn, iv->global_stats,
        (const long long *)(d_offs + 4), carry, stream_);
    if (rc) return rc;
  }
  s->n_abs += n_new; s->finished = finished != 0;
  // ---- splice + LDA + posteriors for the frames whose right context is there
  const int64_t P0 = s->n_post, P1 = finished ? s->n_abs : std::max<int64_t>(P0, s->n_abs - rc_), nP = P1 - P0, pend_rows = P0 - s->a0;
  if (nP > 0) {
    if ((rc = s->xpost.reserve((size_t)nP * D * 4))) return rc;
    int i0 = s->pend_i, i1 = s->pend_i, i2 = s->pend_i, i3 = s->pend_i;      // grow the pending arrays in step (they share pend_i)
    if ((size_t)(pend_rows + nP) * D * 4 > s->px[s->pend_i].cap || (size_t)(pend_rows + nP) * S * 4 > s->pg[s->pend_i].cap ||
        (size_t)(pend_rows + nP) * S * 4 > s->pw[s->pend_i].cap || (size_t)(pend_rows + nP) * 4 > s->pn[s->pend_i].cap) {
      if ((rc = carry_over(s->px, &i0, (size_t)D * 4, 0, pend_rows, pend_rows + nP, stream)) ||
          (rc = carry_over(s->pg, &i1, (size_t)S * 4, 0, pend_rows, pend_rows + nP, stream)) ||
          (rc = carry_over(s->pw, &i2, (size_t)S * 4, 0, pend_rows, pend_rows + nP, stream)) || (rc = carry_over(s->pn, &i3, 4, 0, pend_rows, pend_rows + nP, stream))) return rc;
      s->pend_i = i0;
    }
    float *px = (float *)s->px[s->pend_i].p + pend_rows * D;
    int32_t *pg = (int32_t *)s->pg[s->pend_i].p + pend_rows * S;
    float *pw = (float *)s->pw[s->pend_i].p + pend_rows * S;
    int32_t *pn = (int32_t *)s->pn[s->pend_i].p + pend_rows;
    const unsigned nb = (unsigned)((nP * D + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(ivec_splice_lda_kernel, dim3(nb), dim3(kBlock), 0, stream, (const float *)cm, (int64_t)F, d_offs + 2, 1, iv->lda,
        F * iv->splice + iv->has_offset, iv->has_offset, F, D, lc, rc_,
                       (float *)s->xpost.p, nP, P0 - s->cm_c0);
    if (iv->o.online_cmvn_iextractor)
      hipLaunchKernelGGL(ivec_splice_lda_kernel, dim3(nb), dim3(kBlock), 0, stream, (const float *)cm, (int64_t)F, d_offs + 2, 1, iv->lda,
          F * iv->splice + iv->has_offset, iv->has_offset, F, D, lc, rc_, px, nP, P0 - s->cm_c0);
    else
      hipLaunchKernelGGL(ivec_splice_lda_kernel, dim3(nb), dim3(kBlock), 0, stream, (const float *)raw, (int64_t)F, d_offs, 1, iv->lda,
          F * iv->splice + iv->has_offset, iv->has_offset, F, D, lc, rc_, px, nP, P0 - s->raw_s0);
    const int nw = kBlock / kWave; const size_t lds_post = ((size_t)nw * G + (size_t)nw * 2 * S) * 4;
    K3_REQUIRE(lds_post <= 64 * 1024, "k3_ivector_stream_accept: too many Gaussians for the posterior kernel's LDS tile");
    const float min_post = iv->o.min_post < 0.99f ? iv->o.min_post : 0.99f;
    hipLaunchKernelGGL(ivec_posterior_kernel, dim3((unsigned)((nP + nw - 1) / nw)), dim3(kBlock), lds_post, stream, (const float *)s->xpost.p, D, G,
        iv->gconsts, iv->miv_t, iv->iv_t, S, min_post,
                       iv->o.posterior_scale, nP, pg, pw, pn);
    K3_HIP_CHECK(hipGetLastError());
    s->n_post = P1;
  }
  // ---- one estimate per period among the frames with posteriors: k * period < n_post
  const int64_t k_end = (s->n_post + P - 1) / P, nk = k_end - s->k_next;
  if (h_num_new_rows) *h_num_new_rows = (int32_t)nk;
  if (nk > 0) {
    K3_REQUIRE(!d_new_rows || nk <= max_new_rows,
        "k3_ivector_stream_accept: more new rows than the caller's buffer holds ((num_frames + right_context) / ivector_period + 1 is enough)");
    if ((rc = s->rows.reserve((size_t)nk * R * 4))) return rc;
    EstParams p;
    p.xstats = (const float *)s->px[s->pend_i].p;
    p.frame_off = d_offs + 6;
    p.post_g = (const int32_t *)s->pg[s->pend_i].p;
    p.post_w = (const float *)s->pw[s->pend_i].p;
    p.post_n = (const int32_t *)s->pn[s->pend_i].p;
    p.U = iv->U;
    p.SM = iv->SM;
    p.quad_g = iv->quad_in_lds ? nullptr : (double *)s->quad.p;
    p.chol_g = (double *)s->chol.p;
    p.out = (float *)s->rows.p;
    p.ld_out = R;
    p.out_off = d_offs + 5;
    p.D = D; p.R = R; p.S = S; p.period = P; p.num_cg_iters = iv->o.num_cg_iters; p.exact_solve = iv->o.exact_solve; p.prior = iv->prior_offset; p.max_count = iv->o.max_count;
    p.state_in = est; p.state_out = est; p.acc_tail = 0; p.t_base = s->a0; p.t_limit = s->n_post; p.k_begin = (int)s->k_next; p.x_io = x_io; p.segs = nullptr;
    const size_t lds_est = (size_t)(6 * R + kBlock / kWave) * 8 + (size_t)P * S * 16 + 8 + (iv->quad_in_lds ? (size_t)R * R * 8 : 0);
    K3_REQUIRE(lds_est <= 156 * 1024, "k3_ivector_stream_accept: ivector_period * num_gselect too large for the estimation kernel's LDS");
    K3_HIP_CHECK(hipFuncSetAttribute((const void *)ivec_estimate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_est));
    hipLaunchKernelGGL(ivec_estimate_kernel, dim3(1), dim3(kBlock), lds_est, stream, p);
    K3_HIP_CHECK(hipGetLastError());
    K3_HIP_CHECK(hipMemcpyAsync(s->latest.p, (const float *)s->rows.p + (nk - 1) * R, (size_t)R * 4, hipMemcpyDeviceToDevice, stream));
    if (d_new_rows) K3_HIP_CHECK(hipMemcpy2DAsync(d_new_rows, (size_t)ld_rows * 4, s->rows.p, (size_t)R * 4, (size_t)R * 4, (size_t)nk, hipMemcpyDeviceToDevice, stream));
    s->k_next = k_end;
    // the statistics now hold the frames up to (k_next - 1) * period: the posterior-stage rows before that are done with
    const int64_t new_a0 = (s->k_next - 1) * P + 1, left = s->n_post - new_a0;
    int i0 = s->pend_i, i1 = s->pend_i, i2 = s->pend_i, i3 = s->pend_i;
    if ((rc = carry_over(s->px, &i0, (size_t)D * 4, new_a0 - s->a0, left, left, stream)) || (rc = carry_over(s->pg, &i1, (size_t)S * 4, new_a0 - s->a0, left, left, stream)) ||
        (rc = carry_over(s->pw, &i2, (size_t)S * 4, new_a0 - s->a0, left, left, stream)) || (rc = carry_over(s->pn, &i3, 4, new_a0 - s->a0, left, left, stream))) return rc;
    s->pend_i = i0; s->a0 = new_a0;
  }
  if (d_latest) K3_HIP_CHECK(hipMemcpyAsync(d_latest, s->latest.p, (size_t)R * 4, hipMemcpyDeviceToDevice, stream));
  return K3_OK;
}

// ---------------------------------------------------------------------------------------------------------------- streaming, many streams per call
namespace {
__global__ void __launch_bounds__(kBlock) ivec_batch_carry_kernel(const BatchSeg *__restrict__ segs, const float *__restrict__ feats, long long ld_feats, int F) {
  const BatchSeg &g = segs[blockIdx.x];
  for (long long i = threadIdx.x; i < g.keep * F; i += kBlock) g.raw[i] = g.raw_old[g.raw_from * F + i];
  for (long long i = threadIdx.x; i < g.n_new * F; i += kBlock) g.raw[g.keep * F + i] = feats[(g.feat_off + i / F) * ld_feats + i % F];
  for (long long i = threadIdx.x; i < g.cm_keep * F; i += kBlock) g.cm[i] = g.cm_old[g.cm_from * F + i];
}
__global__ void __launch_bounds__(kBlock) ivec_batch_shift_kernel(const BatchSeg *__restrict__ segs, int D, int S) {
  const BatchSeg &g = segs[blockIdx.x]; if (g.nk == 0) return;
  for (long long i = threadIdx.x; i < g.left * D; i += kBlock) g.px_alt[i] = g.e_px[g.shift_from * D + i];
  for (long long i = threadIdx.x; i < g.left * S; i += kBlock) { g.pg_alt[i] = g.e_pg[g.shift_from * S + i]; g.pw_alt[i] = g.e_pw[g.shift_from * S + i]; }
  for (long long i = threadIdx.x; i < g.left; i += kBlock) g.pn_alt[i] = g.e_pn[g.shift_from + i];
}
__global__ void ivec_batch_latest_kernel(const BatchSeg *__restrict__ segs, int R, float *__restrict__ out, long long ld_out) {
  if ((int)threadIdx.x < R) out[blockIdx.x * ld_out + threadIdx.x] = segs[blockIdx.x].latest[threadIdx.x];
}
}  // namespace

// k3_ivector_stream_accept for the streams of one batch in one launch per stage (BatchedIvectorExtractorCuda::GetIvectors per chunk,
// cudafeat/feature-online-batched-ivector-cuda.h:30-61):
// stream i takes feature rows h_frame_offsets[i] .. [i + 1] of d_feats; d_latest [num_streams x ld_latest] receives every stream's most recent estimate.  The per-stream results are
// those of k3_ivector_stream_accept (the same kernels' arithmetic; tests/test_ivector_gpu.py).  All streams must belong to one extractor; one batched call at a time per extractor.
extern "C" int k3_ivector_stream_accept_batch(k3_ivector_stream **streams, int32_t num_streams, const float *d_feats, int64_t ld_feats,
    const int64_t *h_frame_offsets, const int32_t *h_finished,
                                              float *d_latest, int64_t ld_latest, void *stream_) {
  K3_REQUIRE(streams && num_streams > 0 && streams[0] && h_frame_offsets && h_finished && h_frame_offsets[0] == 0, "k3_ivector_stream_accept_batch: bad argument");
  k3_ivector *iv = streams[0]->iv; hipStream_t stream = (hipStream_t)stream_;
  const int F = iv->F, D = iv->D, G = iv->G, R = iv->R, S = iv->o.num_gselect, P = iv->o.ivector_period, lc = iv->o.left_context, rc_ = iv->o.right_context,
      W = iv->o.cmvn.cmn_window;
  const int64_t keepN = std::max<int64_t>(W, (int64_t)lc + rc_ + 1);
  K3_REQUIRE(h_frame_offsets[num_streams] == 0 || (d_feats && ld_feats >= F), "k3_ivector_stream_accept_batch: features missing");
  K3_REQUIRE(!d_latest || ld_latest >= R, "k3_ivector_stream_accept_batch: leading dimension of the output smaller than the i-vector dimension");
  for (int i = 0; i < num_streams; i++) {
    K3_REQUIRE(streams[i] && streams[i]->iv == iv && h_frame_offsets[i + 1] >= h_frame_offsets[i],
        "k3_ivector_stream_accept_batch: streams of different extractors, or descending offsets");
    K3_REQUIRE(!streams[i]->finished, "k3_ivector_stream_accept_batch: a stream has ended (k3_ivector_stream_reset starts the next one)");
    K3_REQUIRE(!streams[i]->poisoned,
        "k3_ivector_stream_accept_batch: an earlier batched call on a stream failed half-way, its state is lost (k3_ivector_stream_reset starts over)");
    for (int j = 0; j < i; j++) K3_REQUIRE(streams[j] != streams[i], "k3_ivector_stream_accept_batch: a stream listed twice");
  }
  // everything that can be refused is refused BEFORE a stream's state moves: the kernels' LDS limits depend on the extractor only
  const size_t lds_post = ((size_t)(kBlock / kWave) * G + (size_t)(kBlock / kWave) * 2 * S) * 4,
      lds_est = (size_t)(6 * R + kBlock / kWave) * 8 + (size_t)P * S * 16 + 8 + (iv->quad_in_lds ? (size_t)R * R * 8 : 0);
  K3_REQUIRE(lds_post <= 64 * 1024, "k3_ivector_stream_accept_batch: too many Gaussians for the posterior kernel's LDS tile");
  K3_REQUIRE(lds_est <= 156 * 1024, "k3_ivector_stream_accept_batch: ivector_period * num_gselect too large for the estimation kernel's LDS");
  // The planning loop below switches buffers and advances counters stream by stream while it sizes the launches; an error after it has begun (an allocation that fails for a
  // later stream, a HIP error) would leave the streams visited so far out of step with their device buffers.  They are then marked: the next call on them says so instead of
  // producing wrong i-vectors, and k3_ivector_stream_reset starts them over.
  struct Poison { k3_ivector_stream **s; int n; bool ok = false; ~Poison() { if (!ok) for (int i = 0; i < n; i++) s[i]->poisoned = true; } } poison{streams, num_streams};
  std::vector<BatchSeg> segs((size_t)num_streams); std::vector<k3::CmvnSeg> cs((size_t)num_streams);
  int rc; int64_t total_nP = 0; bool any_new = false, any_est = false;
  for (int i = 0; i < num_streams; i++) {
    k3_ivector_stream *s = streams[i]; BatchSeg &g = segs[i]; g = BatchSeg{};
    const int64_t n_new = h_frame_offsets[i + 1] - h_frame_offsets[i]; const bool fin = h_finished[i] != 0;
    // raw and normalised frames: the other buffer of each pair becomes [what is still read | this call's rows] (ivec_batch_carry_kernel)
    const int64_t new_s0 = std::max<int64_t>(0, s->n_abs - keepN), keep = s->n_abs - new_s0, raw_rows = keep + n_new;
    const int64_t new_c0 = std::max<int64_t>(0, s->n_post - lc), cm_keep = s->n_abs - new_c0, cm_rows = cm_keep + n_new;
    g.raw_old = (const float *)s->raw[s->raw_i].p; g.cm_old = (const float *)s->cm[s->cm_i].p; g.raw_from = new_s0 - s->raw_s0; g.cm_from = new_c0 - s->cm_c0;
    // (capacity + switch; the kernel copies)
    if ((rc = carry_over(s->raw, &s->raw_i, (size_t)F * 4, 0, 0, raw_rows, stream)) ||
        (rc = carry_over(s->cm, &s->cm_i, (size_t)F * 4, 0, 0, cm_rows, stream))) return rc;
    g.raw = (float *)s->raw[s->raw_i].p;
    g.cm = (float *)s->cm[s->cm_i].p;
    g.keep = keep;
    g.n_new = n_new;
    g.feat_off = h_frame_offsets[i];
    g.cm_keep = cm_keep;
    g.raw_rows = raw_rows;
    g.cm_rows = cm_rows;
    s->raw_s0 = new_s0; s->cm_c0 = new_c0;
    double *carry = (double *)s->state.p, *est = carry + 3 * F;
    g.est = est;
    g.x_io = est + 1 + R + (size_t)R * R;
    g.chol = (double *)s->chol.p;
    g.quad = iv->quad_in_lds ? nullptr : (double *)s->quad.p;
    cs[i].in = g.raw; cs[i].out = g.cm + (cm_keep - keep) * F; cs[i].rows = raw_rows; cs[i].t_begin = keep; cs[i].carry = carry;
    any_new = any_new || n_new > 0;
    s->n_abs += n_new; s->finished = fin;
    // posterior stage
    const int64_t P0 = s->n_post, P1 = fin ? s->n_abs : std::max<int64_t>(P0, s->n_abs - rc_), nP = P1 - P0, pend_rows = P0 - s->a0;
    g.nP = nP; g.out_off = total_nP; g.row0_cm = P0 - s->cm_c0; g.row0_raw = P0 - s->raw_s0; total_nP += nP;
    if (nP > 0 && ((size_t)(pend_rows + nP) * D * 4 > s->px[s->pend_i].cap || (size_t)(pend_rows + nP) * S * 4 > s->pg[s->pend_i].cap ||
        (size_t)(pend_rows + nP) * S * 4 > s->pw[s->pend_i].cap ||
                   (size_t)(pend_rows + nP) * 4 > s->pn[s->pend_i].cap)) {
      int i0 = s->pend_i, i1 = s->pend_i, i2 = s->pend_i, i3 = s->pend_i;
      if ((rc = carry_over(s->px, &i0, (size_t)D * 4, 0, pend_rows, pend_rows + nP, stream)) ||
          (rc = carry_over(s->pg, &i1, (size_t)S * 4, 0, pend_rows, pend_rows + nP, stream)) ||
          (rc = carry_over(s->pw, &i2, (size_t)S * 4, 0, pend_rows, pend_rows + nP, stream)) || (rc = carry_over(s->pn, &i3, 4, 0, pend_rows, pend_rows + nP, stream))) return rc;
      s->pend_i = i0;
    }
    g.e_px = (const float *)s->px[s->pend_i].p;
    g.e_pg = (const int32_t *)s->pg[s->pend_i].p;
    g.e_pw = (const float *)s->pw[s->pend_i].p;
    g.e_pn = (const int32_t *)s->pn[s->pend_i].p;
    g.px = (float *)s->px[s->pend_i].p + pend_rows * D;
    g.pg = (int32_t *)s->pg[s->pend_i].p + pend_rows * S;
    g.pw = (float *)s->pw[s->pend_i].p + pend_rows * S;
    g.pn = (int32_t *)s->pn[s->pend_i].p + pend_rows;
    s->n_post = P1;
    // estimates
    const int64_t k_end = (s->n_post + P - 1) / P, nk = k_end - s->k_next;
    g.t_base = s->a0; g.t_limit = s->n_post; g.k_begin = (int)s->k_next; g.nk = (int)nk; g.latest = (float *)s->latest.p;
    if (nk > 0) {
      any_est = true;
      if ((rc = s->rows.reserve((size_t)nk * R * 4))) return rc;
      g.rows = (float *)s->rows.p; s->k_next = k_end;
      const int64_t new_a0 = (s->k_next - 1) * P + 1, left = s->n_post - new_a0;
      g.shift_from = new_a0 - s->a0; g.left = left;
      int i0 = s->pend_i, i1 = s->pend_i, i2 = s->pend_i, i3 = s->pend_i;      // (capacity + switch; ivec_batch_shift_kernel copies)
      if ((rc = carry_over(s->px, &i0, (size_t)D * 4, 0, 0, left, stream)) || (rc = carry_over(s->pg, &i1, (size_t)S * 4, 0, 0, left, stream)) ||
          (rc = carry_over(s->pw, &i2, (size_t)S * 4, 0, 0, left, stream)) || (rc = carry_over(s->pn, &i3, 4, 0, 0, left, stream))) return rc;
      s->pend_i = i0; s->a0 = new_a0;
      g.px_alt = (float *)s->px[s->pend_i].p; g.pg_alt = (int32_t *)s->pg[s->pend_i].p; g.pw_alt = (float *)s->pw[s->pend_i].p; g.pn_alt = (int32_t *)s->pn[s->pend_i].p;
    }
  }
  // the tables go up in one copy
  const size_t seg_bytes = segs.size() * sizeof(BatchSeg), cs_bytes = cs.size() * sizeof(k3::CmvnSeg);
  if ((rc = iv->frame_off.reserve(seg_bytes + cs_bytes))) return rc;
  if (total_nP > 0 && (rc = iv->xpost.reserve((size_t)total_nP * D * 4))) return rc;
  K3_HIP_CHECK(hipMemcpyAsync(iv->frame_off.p, segs.data(), seg_bytes, hipMemcpyHostToDevice, stream));
  K3_HIP_CHECK(hipMemcpyAsync((char *)iv->frame_off.p + seg_bytes, cs.data(), cs_bytes, hipMemcpyHostToDevice, stream));
  K3_HIP_CHECK(hipStreamSynchronize(stream));                   // the tables are locals
  const BatchSeg *d_segs = (const BatchSeg *)iv->frame_off.p; const k3::CmvnSeg *d_cs = (const k3::CmvnSeg *)((const char *)iv->frame_off.p + seg_bytes);
  hipLaunchKernelGGL(ivec_batch_carry_kernel, dim3((unsigned)num_streams), dim3(kBlock), 0, stream, d_segs, d_feats, (long long)ld_feats, F);
  if (any_new && (rc = k3::cmvn_online_resume_segs_async(d_cs, num_streams, F, &iv->o.cmvn, iv->global_stats, stream_))) return rc;
  if (total_nP > 0) {
    const unsigned nb = (unsigned)((total_nP * D + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(ivec_splice_lda_multi_kernel, dim3(nb), dim3(kBlock), 0, stream, d_segs, num_streams, 0, 0, iv->lda, F * iv->splice + iv->has_offset,
        iv->has_offset, F, D, lc, rc_, (float *)iv->xpost.p, total_nP);
    hipLaunchKernelGGL(ivec_splice_lda_multi_kernel, dim3(nb), dim3(kBlock), 0, stream, d_segs, num_streams, 1, iv->o.online_cmvn_iextractor ? 1 : 0, iv->lda,
        F * iv->splice + iv->has_offset, iv->has_offset, F, D, lc, rc_,
                       (float *)iv->xpost.p, total_nP);
    const int nw = kBlock / kWave;
    const float min_post = iv->o.min_post < 0.99f ? iv->o.min_post : 0.99f;
    hipLaunchKernelGGL(ivec_posterior_kernel, dim3((unsigned)((total_nP + nw - 1) / nw)), dim3(kBlock), lds_post, stream, (const float *)iv->xpost.p, D, G,
        iv->gconsts, iv->miv_t, iv->iv_t, S, min_post,
                       iv->o.posterior_scale, total_nP, (int32_t *)nullptr, (float *)nullptr, (int32_t *)nullptr, d_segs, num_streams);
  }
  if (any_est) {
    EstParams p{};
    p.U = iv->U;
    p.SM = iv->SM;
    p.D = D;
    p.R = R;
    p.S = S;
    p.period = P;
    p.num_cg_iters = iv->o.num_cg_iters;
    p.exact_solve = iv->o.exact_solve;
    p.prior = iv->prior_offset;
    p.max_count = iv->o.max_count;
    p.acc_tail = 0; p.t_limit = 0; p.segs = d_segs;
    K3_HIP_CHECK(hipFuncSetAttribute((const void *)ivec_estimate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_est));
    hipLaunchKernelGGL(ivec_estimate_kernel, dim3((unsigned)num_streams), dim3(kBlock), lds_est, stream, p);
    hipLaunchKernelGGL(ivec_batch_shift_kernel, dim3((unsigned)num_streams), dim3(kBlock), 0, stream, d_segs, D, S);
  }
  if (d_latest) hipLaunchKernelGGL(ivec_batch_latest_kernel, dim3((unsigned)num_streams), dim3(256), 0, stream, d_segs, R, d_latest, (long long)ld_latest);
  K3_HIP_CHECK(hipGetLastError());
  K3_HIP_CHECK(hipStreamSynchronize(stream));                   // the tables in iv->frame_off are reused by the next call
  poison.ok = true;
  return K3_OK;
}
