// k3_chain.hip -- LF-MMI denominator forward-backward on gfx950 (first slice of the chain-training row).  Paths relative to
// the reference's src/.
//
// Replaces chain::DenominatorComputation::Forward / Backward (chain/chain-denominator.cc:106-440; GPU reference: chain/chain-kernels.cu:108-296,
// one launch per frame and direction with grid = HMM states x sequence blocks, plus three cudamatrix calls per frame for the leaky-HMM terms:
// ~8 launches per frame, ~800 per minibatch of 50-frame sequences, each a few microseconds of work) and DenominatorGraph's constructor
// (chain/chain-den-graph.cc:29-143: transitions by source and by destination, initial probabilities by 100 steps of HMM propagation).
//
// MI355X design, not a port: the sequences of a minibatch are independent (they only share the graph), and one sequence's alpha / beta vector
// (one float per HMM state: 16 KB at 4 k states) fits the LDS of a CU with room to spare.  So ONE launch does the whole minibatch, one workgroup
// per sequence walking all frames forward and then backward:
//   * alpha-dash of the previous frame, the alpha being built, the frame's exp(nnet output) row (read where it lies: a sequence's frame is one
//     contiguous row of the [frames x sequences, pdfs] output matrix -- the reference transposes the whole matrix first) and, going backward,
//     beta / beta-dash and the frame's derivative row all live in LDS; the graph (12 B per transition) streams from L2, shared by every CU;
//   * HBM traffic is the alpha-dash rows (written once going forward, read once going backward), the output rows (read twice) and the
//     derivative rows (one read-modify-write each): the kernel is bound by the LDS gathers per transition, not by HBM;
//   * the work is spread over TRANSITIONS, not states (in- and out-degrees of a denominator graph are very uneven): a wavefront reads 64 consecutive
//     transitions (structure of arrays, coalesced), gathers the two LDS operands and adds the product into its state's accumulator with an LDS
//     atomic on a double -- the reference accumulates each state in double too, so the order of the additions is invisible after the final
//     rounding to float; the list is ordered by the state at the OTHER end, which scatters the 64 atomics of a wavefront.
// Arithmetic follows the CPU reference: per-state sums in double over float products, 'arbitrary scale' = 1 / (alpha-sum of the previous
// frame) folded into every transition, leaky-HMM terms as in AlphaDash / Beta (:200-248), exp limited to [-30, 30] (:91), the betas carrying
// 1 / total-prob (:320-336).  Summation ORDER differs (atomics; the derivative rows are float), so
// results agree with the reference to float rounding, not bit for bit: tests hold |objf difference| <= 1e-4 relative and derivatives <= 1e-5 absolute.
#include "k3_common.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace {
constexpr int kBlock = 1024, kWaves = kBlock / 64;
struct Tr { float p; int pdf; int st; };      // DenominatorGraphTransition (chain/chain-datastruct.h:41-46), host side only

// One transition list in two orders, structure of arrays (a wavefront reads 64 consecutive transitions: four coalesced loads).  Forward pass: ordered by
// SOURCE state, so that the 64 atomic adds of a wavefront go to scattered destinations; backward pass: ordered by DESTINATION state for the same reason.
struct Edges { const float *p; const int *pdf, *src, *dst; };
struct DenParams {
  Edges by_src, by_dst; int E; const float *init;
  int S, P, B, T; float leaky, deriv_weight;
  const float *out; long long ld; float *deriv; long long ld_deriv;
  float *alpha;            // [B][T + 1][S + 1] alpha-dash per frame, the alpha-sum in column S
  double *logprob;         // [B] log total prob + correction for the arbitrary scales
  float *check;            // [B][2] at t = 0: sum_h alpha-dash * beta-dash, sum of the derivative row (both 1 when all is well)
  int do_backward;
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ double block_sum_f64(double v, double *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum_f64(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < kWaves; w++) r += red[w];
  return r;
}

__global__ __launch_bounds__(kBlock) void k3_chain_den_kernel(DenParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[kWaves];
  const int S = p.S, P = p.P, E = p.E, s = blockIdx.x, tid = threadIdx.x;
  double *acc = reinterpret_cast<double *>(smem);                 // [S] per-state sums of the frame being built (double, like the reference's accumulators)
  // [P] backward: the frame's derivative row (double: ds_add_f64 runs at several times the rate of ds_add_f32 here, and the sums lose nothing)
  double *drow = acc + S;
  float *a_prev = reinterpret_cast<float *>(drow + P);            // [S + 1] alpha-dash of the previous frame (backward: beta of the next frame)
  float *probs = a_prev + (S + 1);                                // [P] exp of the frame's output row
  float *occ = probs + P;                                         // [S + 1] backward: alpha-dash of this frame / its alpha-sum
  float *alpha = p.alpha + (long long)s * (p.T + 1) * (S + 1);
  const float leaky = p.leaky;
  auto load_probs = [&](int t) {
    const float *row = p.out + ((long long)t * p.B + s) * p.ld;
    for (int k = tid; k < P; k += kBlock) { float x = row[k]; x = x < -30.0f ? -30.0f : (x > 30.0f ? 30.0f : x); probs[k] = expf(x); }      // ApplyExpLimited(-30, 30), :91
  };
  // ---- forward (:106-260)
  for (int h = tid; h < S; h += kBlock) acc[h] = (double)p.init[h];      // AlphaFirstFrame (scale 1 below)
  double corr = 0.0;      // sum over t < T of log(alpha-sum_t): ComputeTotLogLike's correction term
  for (int t = 0; t <= p.T; t++) {
    double scale = 1.0;
    if (t > 0) {
      load_probs(t - 1);
      for (int h = tid; h < S; h += kBlock) acc[h] = 0.0;
      __syncthreads();
      scale = (double)(float)(1.0 / (double)a_prev[S]);      // 'arbitrary scale' (:186-196)
      const Edges e = p.by_src;
#pragma unroll 8
      for (int k = tid; k < E; k += kBlock) atomicAdd(&acc[e.dst[k]], (double)(a_prev[e.src[k]] * e.p[k] * probs[e.pdf[k]]));
    }
    __syncthreads();
    // AlphaDash (:200-220): alpha-sum, then alpha += leaky * init * alpha-sum
    double part = 0.0;
    for (int h = tid; h < S; h += kBlock) { const float v = (float)(acc[h] * scale); occ[h] = v; part += (double)v; }
    const float asum = (float)block_sum_f64(part, red);
    for (int h = tid; h < S; h += kBlock) { const float v = occ[h] + leaky * p.init[h] * asum; a_prev[h] = v; alpha[(long long)t * (S + 1) + h] = v; }
    if (tid == 0) { a_prev[S] = asum; alpha[(long long)t * (S + 1) + S] = asum; }
    if (t < p.T) corr += log((double)asum);
    __syncthreads();
  }
  // ComputeTotLogLike (:262-300): the last alpha-dash summed over the states
  double part = 0.0;
  for (int h = tid; h < S; h += kBlock) part += (double)a_prev[h];
  const float tot_prob = (float)block_sum_f64(part, red);
  if (tid == 0) p.logprob[s] = log((double)tot_prob) + corr;
  if (!p.do_backward) return;
  // ---- backward (:304-440).  a_prev <- beta of frame t + 1; acc -> beta-dash of frame t.
  float *bdash = occ;      // beta-dash is only needed until the beta is made from it (occ is rebuilt every frame)
  auto beta_from_dash = [&]() {      // Beta(t) (:222-248): beta = beta-dash + leaky * sum_h beta-dash_h * init_h
    double pp = 0.0;
    for (int h = tid; h < S; h += kBlock) pp += (double)(bdash[h] * p.init[h]);
    const float bsum = leaky * (float)block_sum_f64(pp, red);
    for (int h = tid; h < S; h += kBlock) a_prev[h] = bdash[h] + bsum;
    __syncthreads();
  };
  const float inv_tot = 1.0f / tot_prob;
  __syncthreads();
  for (int h = tid; h < S; h += kBlock) bdash[h] = inv_tot;      // BetaDashLastFrame
  __syncthreads();
  beta_from_dash();
  for (int t = p.T - 1; t >= 0; t--) {
    load_probs(t);
    for (int k = tid; k < P; k += kBlock) drow[k] = 0.0;
    const float inv_scale = alpha[(long long)t * (S + 1) + S];
    for (int h = tid; h < S; h += kBlock) { acc[h] = 0.0; occ[h] = alpha[(long long)t * (S + 1) + h] / inv_scale; }      // occupation factor (:372)
    __syncthreads();
    {
      const Edges e = p.by_dst;
#pragma unroll 4
      for (int k = tid; k < E; k += kBlock) {
        const int sr = e.src[k], pd = e.pdf[k];
        const float vf = e.p[k] * a_prev[e.dst[k]] * probs[pd];
        atomicAdd(&acc[sr], (double)vf); atomicAdd(&drow[pd], (double)(vf * occ[sr]));
      }
    }
    __syncthreads();
    double ab = 0.0;
    // (occ is dead: each thread overwrites only the cells it read)
    for (int h = tid; h < S; h += kBlock) {
      const float bd = (float)(acc[h] / (double)inv_scale);
      if (t == 0) ab += (double)(occ[h] * inv_scale * bd);
      bdash[h] = bd;
    }
    if (t == 0) {      // BetaGeneralFrameDebug (:404-440): both sums are 1 per sequence when the computation is healthy
      double ds = 0.0;
      for (int k = tid; k < P; k += kBlock) ds += drow[k];
      ab = block_sum_f64(ab, red); ds = block_sum_f64(ds, red);
      if (tid == 0) { p.check[2 * s] = (float)ab; p.check[2 * s + 1] = (float)ds; }
    }
    float *drv = p.deriv + ((long long)t * p.B + s) * p.ld_deriv;
    for (int k = tid; k < P; k += kBlock) drv[k] += p.deriv_weight * (float)drow[k];      // nnet_output_deriv += deriv_weight * deriv (:318-330)
    beta_from_dash();
  }
}
}  // namespace

struct k3_chain_den {
  int S = 0, P = 0; int E = 0;
  // ep: [2][E] probabilities, ei: [2][3][E] pdf / src / dst, first half ordered by source, second by destination
  float *ep = nullptr;
  int *ei = nullptr;
  float *init = nullptr;
  std::vector<float> h_init;
  float *alpha = nullptr; size_t alpha_cap = 0; double *logprob = nullptr; float *check = nullptr; int b_cap = 0;
};

extern "C" void k3_chain_den_destroy(k3_chain_den *d) {
  if (!d) return;
  for (void *q : {(void *)d->ep, (void *)d->ei, (void *)d->init, (void *)d->alpha, (void *)d->logprob, (void *)d->check}) if (q) (void)hipFree(q);
  delete d;
}

extern "C" int k3_chain_den_create(int32_t num_states, int32_t start, int32_t num_pdfs, const int64_t *arc_offsets, const int32_t *ilabel, const int32_t *nextstate,
                                   const float *weight, const float *final_cost, k3_chain_den **out) {
  K3_REQUIRE(out && arc_offsets && ilabel && nextstate && weight && final_cost && num_states > 0 && num_pdfs > 0 && start >= 0 && start < num_states,
      "k3_chain_den_create: bad argument");
  const int S = num_states; const long long A = arc_offsets[S];
  K3_REQUIRE(A >= 0 && A < (1ll << 30), "k3_chain_den_create: bad arc count");
  // SetTransitions (chain-den-graph.cc:52-95): a transition = (probability exp(-weight), pdf-id = label - 1, the state at its other end); the reference keeps
  // them per source state and per destination state -- here the same two orders as flat lists
  std::vector<std::vector<Tr>> ins(S);
  std::vector<float> ep(2 * (size_t)A); std::vector<int> ei(6 * (size_t)A);
  int *pdf0 = ei.data(), *src0 = pdf0 + A, *dst0 = src0 + A, *pdf1 = dst0 + A, *src1 = pdf1 + A, *dst1 = src1 + A;
  for (int st = 0; st < S; st++)
    for (long long a = arc_offsets[st]; a < arc_offsets[st + 1]; a++) {
      K3_REQUIRE(ilabel[a] >= 1 && ilabel[a] <= num_pdfs && nextstate[a] >= 0 && nextstate[a] < S,
          "k3_chain_den_create: arc label must be pdf-id + 1 in [1, num_pdfs], next state in range");
      const float pr = expf(-weight[a]);
      ep[a] = pr; pdf0[a] = ilabel[a] - 1; src0[a] = st; dst0[a] = nextstate[a];
      ins[nextstate[a]].push_back(Tr{pr, ilabel[a] - 1, st});
    }
  { long long k = 0; for (int st = 0; st < S; st++) for (const Tr &t : ins[st]) { ep[A + k] = t.p; pdf1[k] = t.pdf; src1[k] = t.st; dst1[k] = st; k++; } }
  // SetInitialProbs (:97-143): mass on the start state, 100 steps of propagation through the per-state normalised HMM, averaged
  std::vector<double> norm(S), cur(S, 0.0), nxt(S, 0.0), avg(S, 0.0);
  for (int st = 0; st < S; st++) {
    double tot = std::exp(-(double)final_cost[st]);
    for (long long a = arc_offsets[st]; a < arc_offsets[st + 1]; a++) tot += std::exp(-(double)weight[a]);
    K3_REQUIRE(tot > 0.0 && tot < 100.0, "k3_chain_den_create: a state's outgoing probability mass must be in (0, 100)");
    norm[st] = 1.0 / tot;
  }
  cur[start] = 1.0;
  for (int it = 0; it < 100; it++) {
    for (int st = 0; st < S; st++) avg[st] += (1.0 / 100) * cur[st];
    for (int st = 0; st < S; st++) {
      const double pr = cur[st] * norm[st];
      for (long long a = arc_offsets[st]; a < arc_offsets[st + 1]; a++) nxt[nextstate[a]] += pr * std::exp(-(double)weight[a]);
    }
    cur.swap(nxt); std::fill(nxt.begin(), nxt.end(), 0.0);
    double sum = 0.0; for (double v : cur) sum += v;
    for (double &v : cur) v *= 1.0 / sum;
  }
  auto d = new k3_chain_den; d->S = S; d->P = num_pdfs; d->E = (int)A;
  d->h_init.resize(S); for (int st = 0; st < S; st++) d->h_init[st] = (float)avg[st];
#define K3_TRYC(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { k3_chain_den_destroy(d); k3::set_error("HIP error %s: %s", hipGetErrorName(e__), #e); return K3_ERR_HIP; } } while (0)
  K3_TRYC(hipMalloc(&d->ep, sizeof(float) * std::max<size_t>(1, ep.size())));
  K3_TRYC(hipMalloc(&d->ei, sizeof(int) * std::max<size_t>(1, ei.size())));
  K3_TRYC(hipMalloc(&d->init, sizeof(float) * S));
  K3_TRYC(hipMemcpy(d->ep, ep.data(), sizeof(float) * ep.size(), hipMemcpyHostToDevice)); K3_TRYC(hipMemcpy(d->ei, ei.data(), sizeof(int) * ei.size(), hipMemcpyHostToDevice));
  K3_TRYC(hipMemcpy(d->init, d->h_init.data(), sizeof(float) * S, hipMemcpyHostToDevice));
#undef K3_TRYC
  *out = d;
  return K3_OK;
}

extern "C" int k3_chain_den_num_states(const k3_chain_den *d) { return d ? d->S : 0; }
extern "C" int k3_chain_den_initial_probs(const k3_chain_den *d, float *h_probs) {
  K3_REQUIRE(d && h_probs, "k3_chain_den_initial_probs: null argument");
  std::copy(d->h_init.begin(), d->h_init.end(), h_probs);
  return K3_OK;
}

extern "C" int k3_chain_den_forward_backward(k3_chain_den *d, const float *d_nnet_output, int64_t ld, int32_t num_sequences, int32_t frames_per_sequence,
    float leaky_hmm_coefficient,
                                             float deriv_weight, float *d_nnet_output_deriv, int64_t ld_deriv, float *h_objf, int32_t *h_ok, void *stream) {
  K3_REQUIRE(d && d_nnet_output && h_objf && num_sequences > 0 && frames_per_sequence > 0 && ld >= d->P, "k3_chain_den_forward_backward: bad argument");
  K3_REQUIRE(leaky_hmm_coefficient > 0.0f && leaky_hmm_coefficient < 1.0f, "k3_chain_den_forward_backward: leaky-hmm-coefficient must be in (0, 1) (chain-denominator.cc:58)");
  K3_REQUIRE(!d_nnet_output_deriv || ld_deriv >= d->P, "k3_chain_den_forward_backward: bad derivative stride");
  const size_t lds = sizeof(double) * ((size_t)d->S + (size_t)d->P) + sizeof(float) * (2 * (size_t)(d->S + 1) + (size_t)d->P);
  if (lds > 150 * 1024) {
    k3::set_error("k3_chain_den_forward_backward: %d states x %d pdfs need %zu B of LDS per sequence (limit 150 KB)", d->S, d->P, lds);
    return K3_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t need = (size_t)num_sequences * (frames_per_sequence + 1) * (d->S + 1);
  if (need > d->alpha_cap) {
    if (d->alpha) (void)hipFree(d->alpha);
    d->alpha = nullptr;
    d->alpha_cap = 0;
    K3_HIP_CHECK(hipMalloc(&d->alpha, need * sizeof(float)));
    d->alpha_cap = need;
  }
  if (num_sequences > d->b_cap) {
    if (d->logprob) (void)hipFree(d->logprob); if (d->check) (void)hipFree(d->check); d->logprob = nullptr; d->check = nullptr; d->b_cap = 0;
    K3_HIP_CHECK(hipMalloc(&d->logprob, sizeof(double) * num_sequences)); K3_HIP_CHECK(hipMalloc(&d->check, sizeof(float) * 2 * num_sequences)); d->b_cap = num_sequences;
  }
  DenParams p{};
  const size_t E = (size_t)d->E;
  p.by_src = Edges{d->ep, d->ei, d->ei + E, d->ei + 2 * E}; p.by_dst = Edges{d->ep + E, d->ei + 3 * E, d->ei + 4 * E, d->ei + 5 * E}; p.E = d->E; p.init = d->init;
  p.S = d->S; p.P = d->P; p.B = num_sequences; p.T = frames_per_sequence; p.leaky = leaky_hmm_coefficient; p.deriv_weight = deriv_weight;
  p.out = d_nnet_output; p.ld = ld; p.deriv = d_nnet_output_deriv; p.ld_deriv = ld_deriv; p.alpha = d->alpha; p.logprob = d->logprob; p.check = d->check;
  p.do_backward = d_nnet_output_deriv ? 1 : 0;
  K3_HIP_CHECK(hipFuncSetAttribute((const void *)k3_chain_den_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k3_chain_den_kernel, dim3(num_sequences), dim3(kBlock), lds, st, p);
  K3_HIP_CHECK(hipGetLastError());
  std::vector<double> lp(num_sequences); std::vector<float> chk(2 * (size_t)num_sequences, 1.0f);
  K3_HIP_CHECK(hipMemcpyAsync(lp.data(), d->logprob, sizeof(double) * num_sequences, hipMemcpyDeviceToHost, st));
  if (p.do_backward) K3_HIP_CHECK(hipMemcpyAsync(chk.data(), d->check, sizeof(float) * 2 * num_sequences, hipMemcpyDeviceToHost, st));
  K3_HIP_CHECK(hipStreamSynchronize(st));
  double tot = 0.0; for (double v : lp) tot += v;
  *h_objf = (float)tot;
  if (h_ok) {      // BetaGeneralFrameDebug's verdict (:419-438): abandon the minibatch when either total is off by more than 2
    double ab = 0.0, ds = 0.0; for (int s = 0; s < num_sequences; s++) { ab += chk[2 * s]; ds += chk[2 * s + 1]; }
    *h_ok = (std::isfinite(tot) && std::fabs(ab - num_sequences) <= 2.0 && std::fabs(ds - num_sequences) <= 2.0) ? 1 : 0;
  }
  return K3_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// Numerator of the LF-MMI objective: forward-backward over the supervision FSTs (chain::NumeratorComputation, chain/chain-numerator.cc:115-213) and
// the objective function that puts numerator, denominator and regularisers together (chain::ComputeChainObjfAndDeriv, chain/chain-training.cc:
// 242-330, the non-e2e branch).
// The reference runs the numerator on the CPU over ONE merged FST (the minibatch's supervisions concatenated: a chain of sequences x frames
// levels walked serially in log space, in double).  The sequences of a minibatch are independent: here every sequence keeps its own FST and gets
// its own wavefront, which walks its frames_per_sequence levels forward and backward with all arcs of a level in flight (the merged FST's
// boundary states only differ by the final weights of the previous sequence, which factor out of the posteriors and add up in the total).
// Per level the sums are taken in the linear domain against the level's largest term, accumulated in double by LDS atomics: terms more than
// 700 nats below it vanish, as they do in the reference's LogAdd (which drops everything below -36.7 nats, base/kaldi-math.h:187-205).
namespace {
struct NumParams {
  const int *state_off, *layer_off;      // [B + 1] first state of a sequence (global numbering); [B][T + 2] first state (local) of each time level, level T + 1 = number of states
  const long long *arc_off;              // [total states + 1]
  const int *arc_src, *arc_dst, *arc_pdf; const float *arc_w; const float *final_cost;      // arcs: local source / destination state, pdf-id, weight; per state final cost
  int B, T; float weight;
  const float *out; long long ld; float *deriv; long long ld_deriv; float *xent; long long ld_xent;
  double *logprob;                       // [B]
  int max_states;
};

__global__ __launch_bounds__(64) void k3_chain_num_kernel(NumParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *alpha = reinterpret_cast<double *>(smem), *beta = alpha + p.max_states, *acc = beta + p.max_states;      // log alpha, log beta, linear-domain accumulators
  const int n = blockIdx.x, lane = threadIdx.x, T = p.T;
  const int s0 = p.state_off[n], S = p.state_off[n + 1] - s0; const int *lo = p.layer_off + (long long)n * (T + 2);
  const long long *aoff = p.arc_off + s0; const float *fin = p.final_cost + s0;
  const double kNegInf = -__builtin_inf();
  for (int i = lane; i < S; i += 64) { alpha[i] = i == 0 ? 0.0 : kNegInf; acc[i] = 0.0; }
  __syncthreads();
  auto wave_max = [](double v) { for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o); v = t > v ? t : v; } return v; };
  auto loglike = [&](int t, int pdf) { return (double)p.out[((long long)t * p.B + n) * p.ld + pdf]; };
  // ---- forward (:115-160): level t's arcs consume frame t
  for (int t = 0; t < T; t++) {
    const long long a0 = aoff[lo[t]], a1 = aoff[lo[t + 1]];
    double m = kNegInf;
    for (long long a = a0 + lane; a < a1; a += 64) { const double x = alpha[p.arc_src[a]] + loglike(t, p.arc_pdf[a]) - (double)p.arc_w[a]; m = x > m ? x : m; }
    m = wave_max(m);
    if (m != kNegInf) for (long long a = a0 + lane; a < a1; a += 64) {
      const double x = alpha[p.arc_src[a]] + loglike(t, p.arc_pdf[a]) - (double)p.arc_w[a];
      atomicAdd(&acc[p.arc_dst[a]], exp(x - m));
    }
    __syncthreads();
    for (int i = lo[t + 1] + lane; i < lo[t + 2]; i += 64) { alpha[i] = acc[i] > 0.0 ? m + log(acc[i]) : kNegInf; acc[i] = 0.0; }
    __syncthreads();
  }
  // total log-prob over the final states (:150-156)
  double m = kNegInf;
  for (int i = lo[T] + lane; i < lo[T + 1]; i += 64) { const double x = alpha[i] - (double)fin[i]; m = x > m ? x : m; }
  m = wave_max(m);
  double sum = 0.0;
  if (m != kNegInf) for (int i = lo[T] + lane; i < lo[T + 1]; i += 64) sum += exp(alpha[i] - (double)fin[i] - m);
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const double tot = m != kNegInf ? m + log(sum) : kNegInf;
  if (lane == 0) p.logprob[n] = tot;
  if (!p.deriv && !p.xent) return;
  // ---- backward (:163-213): log beta, occupation probabilities into the derivative rows
  for (int i = lo[T] + lane; i < lo[T + 1]; i += 64) beta[i] = -(double)fin[i];
  __syncthreads();
  for (int t = T - 1; t >= 0; t--) {
    const long long a0 = aoff[lo[t]], a1 = aoff[lo[t + 1]];
    double mb = kNegInf;
    for (long long a = a0 + lane; a < a1; a += 64) { const double y = loglike(t, p.arc_pdf[a]) - (double)p.arc_w[a] + beta[p.arc_dst[a]]; mb = y > mb ? y : mb; }
    mb = wave_max(mb);
    for (long long a = a0 + lane; a < a1; a += 64) {
      const int src = p.arc_src[a], pdf = p.arc_pdf[a];
      const double y = loglike(t, pdf) - (double)p.arc_w[a] + beta[p.arc_dst[a]];
      if (mb != kNegInf) atomicAdd(&acc[src], exp(y - mb));
      const float occ = (float)exp(alpha[src] + y - tot);      // occupation_logprob (:196-199)
      if (occ != 0.0f) {
        if (p.xent) atomicAdd(&p.xent[((long long)t * p.B + n) * p.ld_xent + pdf], p.weight * occ);
        else atomicAdd(&p.deriv[((long long)t * p.B + n) * p.ld_deriv + pdf], p.weight * occ);
      }
    }
    __syncthreads();
    for (int i = lo[t] + lane; i < lo[t + 1]; i += 64) { beta[i] = acc[i] > 0.0 ? mb + log(acc[i]) : kNegInf; acc[i] = 0.0; }
    __syncthreads();
  }
}

// PenalizeOutOfRange (chain-training.cc:49-93) and the sum of squares of the l2 term
__global__ void k3_chain_penalize_kernel(const float *in, long long ld, float *out, long long ld_out, int rows, int cols, float limit, float scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (long long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols); const float v = in[r * ld + c];
  if (v < -limit) out[r * ld_out + c] -= scale * (v + limit); else if (v > limit) out[r * ld_out + c] -= scale * (v - limit);
}
__global__ void k3_chain_sumsq_kernel(const float *in, long long ld, int rows, int cols, double *result) {
  __shared__ double red[4];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)rows * cols; i += (long long)gridDim.x * blockDim.x) {
    const float v = in[(i / cols) * ld + (i % cols)];
    acc += (double)v * v;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(result, red[0] + red[1] + red[2] + red[3]);
}
}  // namespace

struct k3_chain_supervision {
  int B = 0, T = 0, P = 0, max_states = 0; float weight = 1.0f;
  int *ints = nullptr; long long *arc_off = nullptr; float *floats = nullptr; double *logprob = nullptr; double *scratch = nullptr;
  const int *state_off = nullptr, *layer_off = nullptr, *arc_src = nullptr, *arc_dst = nullptr, *arc_pdf = nullptr; const float *arc_w = nullptr, *final_cost = nullptr;
  // end-to-end ("generic numerator") supervisions: transitions by destination and by pdf, the per-sequence offset of state 0's arcs, alpha rows
  bool e2e = false; int max_pdfs = 0; float *alpha = nullptr;
  const long long *in_off = nullptr, *pt_off = nullptr;
  const int *in_src = nullptr, *in_pdf = nullptr, *pdf_off = nullptr, *pdf_id = nullptr, *pt_src = nullptr, *pt_dst = nullptr;
  const float *in_tp = nullptr, *out_tp = nullptr, *pt_tp = nullptr, *seq_offset = nullptr;
};
extern "C" void k3_chain_supervision_destroy(k3_chain_supervision *s) {
  if (!s) return;
  for (void *q : {(void *)s->ints, (void *)s->arc_off, (void *)s->floats, (void *)s->logprob, (void *)s->scratch, (void *)s->alpha}) if (q) (void)hipFree(q);
  delete s;
}
extern "C" int k3_chain_supervision_create(int32_t num_sequences, int32_t frames_per_sequence, int32_t label_dim, float weight, const int32_t *state_offsets,
    const int64_t *arc_offsets,
                                           const int32_t *ilabel, const int32_t *nextstate, const float *arc_weight, const float *final_cost, k3_chain_supervision **out) {
  K3_REQUIRE(out && state_offsets && arc_offsets && ilabel && nextstate && arc_weight && final_cost && num_sequences > 0 && frames_per_sequence > 0 &&
      label_dim > 0, "k3_chain_supervision_create: bad argument");
  const int B = num_sequences, T = frames_per_sequence; const int NS = state_offsets[B]; const long long NA = arc_offsets[NS];
  std::vector<int> layer((size_t)B * (T + 2)), src(NA), dst(NA), pdf(NA); int max_states = 0;
  // ComputeFstStateTimes (chain-supervision.cc:663-700): start state 0, every arc advances one frame, states sorted by time, all paths T arcs long
  for (int n = 0; n < B; n++) {
    const int s0 = state_offsets[n], S = state_offsets[n + 1] - s0; K3_REQUIRE(S > 0, "k3_chain_supervision_create: empty supervision FST");
    max_states = std::max(max_states, S);
    std::vector<int> time(S, -1); time[0] = 0;
    for (int st = 0; st < S; st++) {
      K3_REQUIRE(time[st] >= 0 && (st == 0 || time[st] >= time[st - 1]),
          "k3_chain_supervision_create: the FST's states must be reachable and sorted by path length from state 0 (ComputeFstStateTimes)");
      for (long long a = arc_offsets[s0 + st]; a < arc_offsets[s0 + st + 1]; a++) {
        K3_REQUIRE(ilabel[a] >= 1 && ilabel[a] <= label_dim && nextstate[a] >= 0 && nextstate[a] < S,
            "k3_chain_supervision_create: arc label must be pdf-id + 1 in [1, label_dim], next state in range");
        K3_REQUIRE(time[nextstate[a]] == -1 || time[nextstate[a]] == time[st] + 1, "k3_chain_supervision_create: all paths to a state must have the same length");
        time[nextstate[a]] = time[st] + 1; src[a] = st; dst[a] = nextstate[a]; pdf[a] = ilabel[a] - 1;
      }
      if (final_cost[s0 + st] != __builtin_inff()) K3_REQUIRE(time[st] == T, "k3_chain_supervision_create: final states must be frames_per_sequence arcs from the start state");
      else K3_REQUIRE(time[st] < T, "k3_chain_supervision_create: a state at the last level must be final");
    }
    int *lo = &layer[(size_t)n * (T + 2)]; int st = 0;
    for (int t = 0; t <= T; t++) { lo[t] = st; while (st < S && time[st] == t) st++; K3_REQUIRE(st > lo[t], "k3_chain_supervision_create: a time level has no states"); }
    lo[T + 1] = S; K3_REQUIRE(st == S, "k3_chain_supervision_create: states beyond the last level");
  }
  auto s = new k3_chain_supervision; s->B = B; s->T = T; s->P = label_dim; s->weight = weight; s->max_states = max_states;
  std::vector<int> ints; ints.insert(ints.end(), state_offsets, state_offsets + B + 1); ints.insert(ints.end(), layer.begin(), layer.end());
  ints.insert(ints.end(), src.begin(), src.end()); ints.insert(ints.end(), dst.begin(), dst.end()); ints.insert(ints.end(), pdf.begin(), pdf.end());
  std::vector<float> fl(arc_weight, arc_weight + NA); fl.insert(fl.end(), final_cost, final_cost + NS);
  std::vector<long long> ao(arc_offsets, arc_offsets + NS + 1);
#define K3_TRYS(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { k3_chain_supervision_destroy(s); k3::set_error("HIP error %s: %s", hipGetErrorName(e__), #e); return K3_ERR_HIP; } } while (0)
  K3_TRYS(hipMalloc(&s->ints, sizeof(int) * ints.size()));
  K3_TRYS(hipMalloc(&s->floats, sizeof(float) * std::max<size_t>(1, fl.size())));
  K3_TRYS(hipMalloc(&s->arc_off, sizeof(long long) * ao.size()));
  K3_TRYS(hipMalloc(&s->logprob, sizeof(double) * B)); K3_TRYS(hipMalloc(&s->scratch, sizeof(double)));
  K3_TRYS(hipMemcpy(s->ints, ints.data(), sizeof(int) * ints.size(), hipMemcpyHostToDevice));
  K3_TRYS(hipMemcpy(s->floats, fl.data(), sizeof(float) * fl.size(), hipMemcpyHostToDevice));
  K3_TRYS(hipMemcpy(s->arc_off, ao.data(), sizeof(long long) * ao.size(), hipMemcpyHostToDevice));
#undef K3_TRYS
  s->state_off = s->ints; s->layer_off = s->state_off + B + 1; s->arc_src = s->layer_off + (size_t)B * (T + 2); s->arc_dst = s->arc_src + NA; s->arc_pdf = s->arc_dst + NA;
  s->arc_w = s->floats; s->final_cost = s->floats + NA;
  *out = s;
  return K3_OK;
}

// ---- the end-to-end ("generic") numerator: chain::GenericNumeratorComputation (chain/chain-generic-numerator.cc:30-463), the forward-backward of flat-start chain training
// over per-sequence FSTs that may have self-loops and several final states.  The reference runs it on the CPU, sequence by sequence in a few host threads, in the log domain
// with LogAdd in float (base/kaldi-math.h:187-205) and a per-frame normaliser kept in an extra column of alpha.  Here one workgroup per sequence in one launch; the arithmetic
// is the reference's, transition by transition in its order: alpha(t, h) = LogAdd over the in-transitions of h in (source state, arc) order (:196-207), minus the previous
// frame's normaliser, normaliser = LogSumExp of the row (:208-216); beta(t, h) over the out-transitions (:330-350); the occupation log-probabilities of a pdf are LogAdd-ed
// over the transitions that carry it in (state, arc) order -- a thread per pdf instead of the reference's serial sweep, same order.  exp / log1p are the device's, so results
// agree to float rounding (tests: 1e-5), not bit for bit.  The offset of state 0's arcs (:84-93; only that state's, as the reference's loop scopes it) is kept.
namespace {
struct E2eParams {
  const int *state_off; const long long *arc_off, *in_off, *pt_off; const int *arc_dst, *arc_pdf, *in_src, *in_pdf, *pdf_off, *pdf_id, *pt_src, *pt_dst;
  const float *out_tp, *in_tp, *pt_tp, *final_cost, *seq_offset;
  int B, T, max_states; float weight;
  const float *out; long long ld; float *deriv; long long ld_deriv;
  float *alpha;              // [B][T + 1][max_states + 1]
  double *logprob;           // [B]
};
__device__ __forceinline__ float k3_logadd(float x, float y) {      // kaldi::LogAdd (float): the smaller term is dropped below log(FLT_EPSILON)
  float diff; if (x < y) { diff = x - y; x = y; } else diff = y - x;
  return diff >= -15.9423847198486328125f ? x + log1pf(expf(diff)) : x;
}
__global__ __launch_bounds__(256) void k3_chain_e2e_num_kernel(E2eParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *beta = reinterpret_cast<float *>(smem);      // [2][max_states]
  __shared__ float redf[4]; __shared__ double redd[4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, T = p.T, B = p.B;
  const int s0 = p.state_off[n], S = p.state_off[n + 1] - s0, ld_a = p.max_states + 1;
  float *A = p.alpha + (long long)n * (T + 1) * ld_a; const float *fin = p.final_cost + s0; const float kNegInf = -__builtin_inff();
  auto probs = [&](int t, int pdf) { return p.out[((long long)t * B + n) * p.ld + pdf]; };
  // MatrixBase::LogSumExp over a row of S floats (max, then exp(x - max) summed in double); every thread gets the result
  auto row_logsumexp = [&](const float *row) -> float {
    float m = kNegInf; for (int h = tid; h < S; h += 256) m = fmaxf(m, row[h]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) redf[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    double sum = 0.0; for (int h = tid; h < S; h += 256) sum += (double)expf(row[h] - m);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) redd[wave] = sum;
    __syncthreads();
    sum = redd[0] + redd[1] + redd[2] + redd[3];
    __syncthreads();
    return m + (float)log(sum);
  };
  // ---- forward (AlphaFirstFrame :122-131, AlphaRemainingFrames :163-236)
  for (int h = tid; h <= S; h += 256) A[h] = (h == 0 || h == S) ? 0.0f : kNegInf;
  __syncthreads();
  double log_scale_product = 0.0;
  for (int t = 1; t <= T; t++) {
    const float *a_tm1 = A + (long long)(t - 1) * ld_a; float *a_t = A + (long long)t * ld_a; const float prev_norm = a_tm1[S];
    for (int h = tid; h < S; h += 256) {
      float a = kNegInf;
      for (long long k = p.in_off[s0 + h]; k < p.in_off[s0 + h + 1]; k++) a = k3_logadd(a, a_tm1[p.in_src[k]] + p.in_tp[k] + probs(t - 1, p.in_pdf[k]));
      a_t[h] = a + -prev_norm;
    }
    __syncthreads();
    const float norm = row_logsumexp(a_t);
    if (tid == 0) a_t[S] = norm;
    log_scale_product += (double)norm;
    __syncthreads();
  }
  float *a_T = A + (long long)T * ld_a;
  log_scale_product -= (double)a_T[S];
  __syncthreads();
  for (int h = tid; h < S; h += 256) a_T[h] += fin[h] == __builtin_inff() ? kNegInf : -fin[h];      // last_alpha.AddVecToRows(final_probs)
  __syncthreads();
  const float tot = row_logsumexp(a_T);
  if (tid == 0) { a_T[S] = tot; p.logprob[n] = ((double)tot - (double)p.seq_offset[n]) + log_scale_product; }
  if (!p.deriv) return;
  // ---- backward (BetaLastFrame :303-320, BetaRemainingFrames :322-363) with the derivative rows (AddSpecificPdfsIndirect :366-401: += weight * exp(log occupation))
  for (int h = tid; h < S; h += 256) beta[(T & 1) * p.max_states + h] = -tot + (fin[h] == __builtin_inff() ? kNegInf : -fin[h]);
  __syncthreads();
  const int j0 = p.pdf_off[n], NP = p.pdf_off[n + 1] - j0;
  for (int t = T - 1; t >= 0; t--) {
    const float *a_t = A + (long long)t * ld_a, *b_tp1 = beta + ((t + 1) & 1) * p.max_states; float *b_t = beta + (t & 1) * p.max_states; const float inv = a_t[S];
    for (int h = tid; h < S; h += 256) {
      float totv = kNegInf;
      for (long long k = p.arc_off[s0 + h]; k < p.arc_off[s0 + h + 1]; k++) totv = k3_logadd(totv, p.out_tp[k] + b_tp1[p.arc_dst[k]] + probs(t, p.arc_pdf[k]) - inv);
      b_t[h] = totv;
    }
    for (int j = tid; j < NP; j += 256) {
      const int pdf = p.pdf_id[j0 + j]; const float pr = probs(t, pdf); float ld_ = kNegInf;
      for (long long k = p.pt_off[j0 + j]; k < p.pt_off[j0 + j + 1]; k++) ld_ = k3_logadd(ld_, (p.pt_tp[k] + b_tp1[p.pt_dst[k]] + pr - inv) + a_t[p.pt_src[k]]);
      p.deriv[((long long)t * B + n) * p.ld_deriv + pdf] += p.weight * expf(ld_);
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" int k3_chain_supervision_create_e2e(int32_t num_sequences, int32_t frames_per_sequence, int32_t label_dim, float weight,
    const int32_t *state_offsets, const int64_t *arc_offsets,
                                               const int32_t *ilabel, const int32_t *nextstate, const float *arc_weight, const float *final_cost, k3_chain_supervision **out) {
  K3_REQUIRE(out && state_offsets && arc_offsets && ilabel && nextstate && arc_weight && final_cost && num_sequences > 0 && frames_per_sequence > 0 &&
      label_dim > 0, "k3_chain_supervision_create_e2e: bad argument");
  const int B = num_sequences, T = frames_per_sequence; const int NS = state_offsets[B]; const long long NA = arc_offsets[NS];
  std::vector<long long> in_off(NS + 1, 0), pt_off;
  std::vector<int> in_src(NA), in_pdf(NA), pdf_off(B + 1, 0), pdf_id, pt_src(NA), pt_dst(NA), dst(NA), pdf(NA);
  std::vector<float> in_tp(NA), out_tp(NA), pt_tp(NA), seq_offset(B, 0.0f);
  int max_states = 0, max_pdfs = 0; bool any_final = true;
  for (int n = 0; n < B; n++) {
    const int s0 = state_offsets[n], S = state_offsets[n + 1] - s0; K3_REQUIRE(S > 0, "k3_chain_supervision_create_e2e: empty supervision FST");
    max_states = std::max(max_states, S);
    float offset = 0.0f; for (long long a = arc_offsets[s0]; a < arc_offsets[s0 + 1]; a++) if (arc_weight[a] > offset) offset = arc_weight[a];      // :84-93
    seq_offset[n] = offset;
    bool fin_any = false;
    for (int st = 0; st < S; st++) {
      if (final_cost[s0 + st] != __builtin_inff()) fin_any = true;
      for (long long a = arc_offsets[s0 + st]; a < arc_offsets[s0 + st + 1]; a++) {
        K3_REQUIRE(ilabel[a] >= 1 && ilabel[a] <= label_dim && nextstate[a] >= 0 && nextstate[a] < S,
            "k3_chain_supervision_create_e2e: arc label must be pdf-id + 1 in [1, label_dim] (epsilon-free), next state in range");
        dst[a] = nextstate[a]; pdf[a] = ilabel[a] - 1; out_tp[a] = -(arc_weight[a] - (st == 0 ? offset : 0.0f));
        in_off[s0 + nextstate[a] + 1]++;
      }
    }
    any_final = any_final && fin_any;
  }
  K3_REQUIRE(any_final, "k3_chain_supervision_create_e2e: a supervision FST without a final state");
  for (int i = 0; i < NS; i++) in_off[i + 1] += in_off[i];
  { std::vector<long long> cur(in_off.begin(), in_off.end() - 1);
    for (int n = 0; n < B; n++) { const int s0 = state_offsets[n], S = state_offsets[n + 1] - s0;
      for (int st = 0; st < S; st++) for (long long a = arc_offsets[s0 + st]; a < arc_offsets[s0 + st + 1]; a++) {
        const long long k = cur[s0 + dst[a]]++;
        in_src[k] = st;
        in_pdf[k] = pdf[a];
        in_tp[k] = out_tp[a];
      }
      }
      }
  // transitions by pdf, per sequence (pdfs in order of first use like index_to_pdf_, transitions of a pdf in (state, arc) order)
  pt_off.push_back(0);
  for (int n = 0; n < B; n++) {
    const int s0 = state_offsets[n], S = state_offsets[n + 1] - s0; std::vector<int> slot(label_dim, -1); std::vector<std::vector<long long>> lists;
    for (int st = 0; st < S; st++) for (long long a = arc_offsets[s0 + st]; a < arc_offsets[s0 + st + 1]; a++) {
      if (slot[pdf[a]] < 0) { slot[pdf[a]] = (int)lists.size(); lists.emplace_back(); pdf_id.push_back(pdf[a]); }
      lists[slot[pdf[a]]].push_back(a);
    }
    for (size_t j = 0; j < lists.size(); j++) {
      long long k = pt_off.back();
      for (long long a : lists[j]) {
        int st = (int)(std::upper_bound(arc_offsets + s0, arc_offsets + s0 + S + 1, (int64_t)a) - (arc_offsets + s0)) - 1;      // the arc's source state
        pt_src[k] = st; pt_dst[k] = dst[a]; pt_tp[k] = out_tp[a]; k++;
      }
      pt_off.push_back(k);
    }
    pdf_off[n + 1] = (int)pdf_id.size(); max_pdfs = std::max(max_pdfs, (int)lists.size());
  }
  auto s = new k3_chain_supervision; s->B = B; s->T = T; s->P = label_dim; s->weight = weight; s->max_states = max_states; s->e2e = true; s->max_pdfs = max_pdfs;
  std::vector<int> ints; auto put_i = [&](const std::vector<int> &v) { const size_t o = ints.size(); ints.insert(ints.end(), v.begin(), v.end()); return o; };
  std::vector<int> so(state_offsets, state_offsets + B + 1);
  const size_t o_so = put_i(so), o_dst = put_i(dst), o_pdf = put_i(pdf), o_isrc = put_i(in_src), o_ipdf = put_i(in_pdf), o_poff = put_i(pdf_off),
      o_pid = put_i(pdf_id), o_psrc = put_i(pt_src), o_pdst = put_i(pt_dst);
  std::vector<float> fl; auto put_f = [&](const float *b, size_t n_) { const size_t o = fl.size(); fl.insert(fl.end(), b, b + n_); return o; };
  const size_t o_otp = put_f(out_tp.data(), NA), o_itp = put_f(in_tp.data(), NA), o_ptp = put_f(pt_tp.data(), NA), o_fin = put_f(final_cost, NS),
      o_off = put_f(seq_offset.data(), B);
  std::vector<long long> lls(arc_offsets, arc_offsets + NS + 1);
  const size_t o_in = lls.size();
  lls.insert(lls.end(), in_off.begin(), in_off.end());
  const size_t o_pt = lls.size();
  lls.insert(lls.end(), pt_off.begin(), pt_off.end());
#define K3_TRYS(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { k3_chain_supervision_destroy(s); k3::set_error("HIP error %s: %s", hipGetErrorName(e__), #e); return K3_ERR_HIP; } } while (0)
  K3_TRYS(hipMalloc(&s->ints, sizeof(int) * std::max<size_t>(1, ints.size())));
  K3_TRYS(hipMalloc(&s->floats, sizeof(float) * std::max<size_t>(1, fl.size())));
  K3_TRYS(hipMalloc(&s->arc_off, sizeof(long long) * lls.size()));
  K3_TRYS(hipMalloc(&s->logprob, sizeof(double) * B));
  K3_TRYS(hipMalloc(&s->scratch, sizeof(double)));
  K3_TRYS(hipMalloc(&s->alpha, sizeof(float) * (size_t)B * (T + 1) * (max_states + 1)));
  K3_TRYS(hipMemcpy(s->ints, ints.data(), sizeof(int) * ints.size(), hipMemcpyHostToDevice));
  K3_TRYS(hipMemcpy(s->floats, fl.data(), sizeof(float) * fl.size(), hipMemcpyHostToDevice));
  K3_TRYS(hipMemcpy(s->arc_off, lls.data(), sizeof(long long) * lls.size(), hipMemcpyHostToDevice));
#undef K3_TRYS
  s->state_off = s->ints + o_so;
  s->arc_dst = s->ints + o_dst;
  s->arc_pdf = s->ints + o_pdf;
  s->in_src = s->ints + o_isrc;
  s->in_pdf = s->ints + o_ipdf;
  s->pdf_off = s->ints + o_poff;
  s->pdf_id = s->ints + o_pid;
  s->pt_src = s->ints + o_psrc;
  s->pt_dst = s->ints + o_pdst;
  s->out_tp = s->floats + o_otp;
  s->in_tp = s->floats + o_itp;
  s->pt_tp = s->floats + o_ptp;
  s->final_cost = s->floats + o_fin;
  s->seq_offset = s->floats + o_off;
  s->in_off = s->arc_off + o_in; s->pt_off = s->arc_off + o_pt;
  *out = s;
  return K3_OK;
}

// GenericNumeratorComputation::ForwardBackward / ComputeObjf: *h_logprob = the total log-probability as the reference returns it (NOT multiplied by the
// supervision weight, whatever
// the comment at chain-training.cc:147 says: :270 assigns the plain sum); the derivative gets weight * occupation probabilities
static int chain_numerator_e2e(k3_chain_supervision *s, const float *d_out, int64_t ld, float *d_deriv, int64_t ld_deriv, float *h_logprob, hipStream_t st) {
  const size_t lds = 2 * sizeof(float) * (size_t)s->max_states;
  if (lds > 150 * 1024) {
    k3::set_error("k3_chain e2e numerator: a supervision FST of %d states needs %zu B of LDS (limit 150 KB)", s->max_states, lds);
    return K3_ERR_UNSUPPORTED;
  }
  E2eParams p{};
  p.state_off = s->state_off;
  p.arc_off = s->arc_off;
  p.in_off = s->in_off;
  p.pt_off = s->pt_off;
  p.arc_dst = s->arc_dst;
  p.arc_pdf = s->arc_pdf;
  p.in_src = s->in_src;
  p.in_pdf = s->in_pdf;
  p.pdf_off = s->pdf_off;
  p.pdf_id = s->pdf_id;
  p.pt_src = s->pt_src;
  p.pt_dst = s->pt_dst;
  p.out_tp = s->out_tp;
  p.in_tp = s->in_tp;
  p.pt_tp = s->pt_tp;
  p.final_cost = s->final_cost;
  p.seq_offset = s->seq_offset;
  p.B = s->B;
  p.T = s->T;
  p.max_states = s->max_states;
  p.weight = s->weight;
  p.out = d_out;
  p.ld = ld;
  p.deriv = d_deriv;
  p.ld_deriv = ld_deriv;
  p.alpha = s->alpha;
  p.logprob = s->logprob;
  K3_HIP_CHECK(hipFuncSetAttribute((const void *)k3_chain_e2e_num_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 16)));
  hipLaunchKernelGGL(k3_chain_e2e_num_kernel, dim3(s->B), dim3(256), std::max<size_t>(lds, 16), st, p);
  K3_HIP_CHECK(hipGetLastError());
  std::vector<double> lp(s->B);
  K3_HIP_CHECK(hipMemcpyAsync(lp.data(), s->logprob, sizeof(double) * s->B, hipMemcpyDeviceToHost, st)); K3_HIP_CHECK(hipStreamSynchronize(st));
  float tot = 0.0f; for (double v : lp) tot += (float)v;      // (BaseFloat partial sums, :262-268)
  *h_logprob = tot;
  return K3_OK;
}

// NumeratorComputation::Forward (+ Backward when a derivative matrix is given): *h_logprob_weighted = weight * total log-prob; deriv (or xent) += weight * occupation probabilities
static int chain_numerator(k3_chain_supervision *s, const float *d_out, int64_t ld, float *d_deriv, int64_t ld_deriv, float *d_xent, int64_t ld_xent,
    float *h_logprob_weighted, hipStream_t st) {
  if (s->e2e) return d_xent ? chain_numerator_e2e(s, d_out, ld, d_xent, ld_xent, h_logprob_weighted, st) :
      chain_numerator_e2e(s, d_out, ld, d_deriv, ld_deriv, h_logprob_weighted, st);
  const size_t lds = 3 * sizeof(double) * (size_t)s->max_states;
  if (lds > 150 * 1024) { k3::set_error("k3_chain numerator: a supervision FST of %d states needs %zu B of LDS (limit 150 KB)", s->max_states, lds); return K3_ERR_UNSUPPORTED; }
  NumParams p{};
  p.state_off = s->state_off;
  p.layer_off = s->layer_off;
  p.arc_off = s->arc_off;
  p.arc_src = s->arc_src;
  p.arc_dst = s->arc_dst;
  p.arc_pdf = s->arc_pdf;
  p.arc_w = s->arc_w;
  p.final_cost = s->final_cost;
  p.B = s->B;
  p.T = s->T;
  p.weight = s->weight;
  p.out = d_out;
  p.ld = ld;
  p.deriv = d_deriv;
  p.ld_deriv = ld_deriv;
  p.xent = d_xent;
  p.ld_xent = ld_xent;
  p.logprob = s->logprob;
  p.max_states = s->max_states;
  K3_HIP_CHECK(hipFuncSetAttribute((const void *)k3_chain_num_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k3_chain_num_kernel, dim3(s->B), dim3(64), lds, st, p);
  K3_HIP_CHECK(hipGetLastError());
  std::vector<double> lp(s->B);
  K3_HIP_CHECK(hipMemcpyAsync(lp.data(), s->logprob, sizeof(double) * s->B, hipMemcpyDeviceToHost, st)); K3_HIP_CHECK(hipStreamSynchronize(st));
  double tot = 0.0; for (double v : lp) tot += v;
  *h_logprob_weighted = (float)(tot * (double)s->weight);
  return K3_OK;
}
extern "C" int k3_chain_numerator(k3_chain_supervision *sup, const float *d_nnet_output, int64_t ld, float *d_nnet_output_deriv, int64_t ld_deriv,
    float *h_logprob_weighted, void *stream) {
  K3_REQUIRE(sup && d_nnet_output && h_logprob_weighted && ld >= sup->P && (!d_nnet_output_deriv || ld_deriv >= sup->P), "k3_chain_numerator: bad argument");
  return chain_numerator(sup, d_nnet_output, ld, d_nnet_output_deriv, ld_deriv, nullptr, 0, h_logprob_weighted, (hipStream_t)stream);
}

extern "C" int k3_chain_objf_and_deriv(k3_chain_den *den, k3_chain_supervision *sup, const k3_chain_training_opts *opts, const float *d_nnet_output, int64_t ld,
                                       float *d_nnet_output_deriv, int64_t ld_deriv, float *d_xent_output_deriv, int64_t ld_xent, float *h_objf,
                                           float *h_l2_term, float *h_weight, void *stream) {
  K3_REQUIRE(den && sup && opts && d_nnet_output && h_objf && h_l2_term && h_weight && sup->P == den->P && ld >= den->P, "k3_chain_objf_and_deriv: bad argument");
  K3_REQUIRE((!d_nnet_output_deriv || ld_deriv >= den->P) && (!d_xent_output_deriv || ld_xent >= den->P), "k3_chain_objf_and_deriv: bad derivative stride");
  hipStream_t st = (hipStream_t)stream; const int rows = sup->B * sup->T, P = den->P; const float w = sup->weight;
  auto zero = [&](float *m, int64_t l) -> int { K3_HIP_CHECK(hipMemset2DAsync(m, (size_t)l * 4, 0, (size_t)P * 4, (size_t)rows, st)); return K3_OK; };
  if (d_nnet_output_deriv) { const int rc = zero(d_nnet_output_deriv, ld_deriv); if (rc) return rc; }                       // :258-259
  float den_logprob = 0.0f; int32_t ok = 1;
  // :261-271
  {
    const int rc = k3_chain_den_forward_backward(den, d_nnet_output, ld, sup->B, sup->T, opts->leaky_hmm_coefficient, -w, d_nnet_output_deriv, ld_deriv,
        &den_logprob, &ok, stream);
    if (rc) return rc;
  }
  const float den_logprob_weighted = w * den_logprob;
  // :273-277 (the reference applies it on a coin flip; here the caller decides)
  if (d_nnet_output_deriv && opts->apply_out_of_range_penalty && opts->out_of_range_regularize != 0.0f) {
    const long long n = (long long)rows * P;
    hipLaunchKernelGGL(k3_chain_penalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_nnet_output, (long long)ld, d_nnet_output_deriv,
        (long long)ld_deriv, rows, P, 30.0f, 2.0f * opts->out_of_range_regularize);
  }
  if (d_xent_output_deriv) { const int rc = zero(d_xent_output_deriv, ld_xent); if (rc) return rc; }                       // :279-282
  float num_logprob_weighted = 0.0f;
  // :285-297
  {
    const int rc = chain_numerator(sup, d_nnet_output, ld, d_xent_output_deriv ? nullptr : d_nnet_output_deriv, ld_deriv, d_xent_output_deriv, ld_xent,
        &num_logprob_weighted, st);
    if (rc) return rc;
  }
  if (d_xent_output_deriv && d_nnet_output_deriv) {
    const int rc = k3_mat_add_mat(1.0f, d_xent_output_deriv, ld_xent, 0, d_nnet_output_deriv, ld_deriv, rows, P, stream);
    if (rc) return rc;
  }
  *h_objf = num_logprob_weighted - den_logprob_weighted; *h_weight = w * sup->B * sup->T;                                    // :299-301
  if (!(*h_objf - *h_objf == 0.0f) || !ok) {                                                                                // :302-314: abandon the minibatch
    if (d_nnet_output_deriv) { const int rc = zero(d_nnet_output_deriv, ld_deriv); if (rc) return rc; }
    if (d_xent_output_deriv) { const int rc = zero(d_xent_output_deriv, ld_xent); if (rc) return rc; }
    *h_objf = -10.0f * *h_weight;
  }
  *h_l2_term = 0.0f;
  // end-to-end supervisions (ComputeChainObjfAndDerivE2e): the l2 term and its derivative only when the numerator computation was fine (`if (opts.l2_regularize != 0.0 &&
  // numerator_ok)`, chain-training.cc:203) -- an abandoned minibatch keeps its zero derivatives; the regular branch (:329-337) applies it unconditionally
  const bool numerator_ok = num_logprob_weighted - num_logprob_weighted == 0.0f;
  if (opts->l2_regularize != 0.0f && (!sup->e2e || numerator_ok)) {                                                         // :329-337
    const float scale = w * opts->l2_regularize; double sumsq = 0.0;
    K3_HIP_CHECK(hipMemsetAsync(sup->scratch, 0, sizeof(double), st));
    hipLaunchKernelGGL(k3_chain_sumsq_kernel, dim3(512), dim3(256), 0, st, d_nnet_output, (long long)ld, rows, P, sup->scratch);
    K3_HIP_CHECK(hipMemcpyAsync(&sumsq, sup->scratch, sizeof(double), hipMemcpyDeviceToHost, st)); K3_HIP_CHECK(hipStreamSynchronize(st));
    *h_l2_term = (float)(-0.5 * (double)scale * sumsq);
    if (d_nnet_output_deriv) { const int rc = k3_mat_add_mat(-scale, d_nnet_output, ld, 0, d_nnet_output_deriv, ld_deriv, rows, P, stream); if (rc) return rc; }
  }
  K3_HIP_CHECK(hipStreamSynchronize(st));
  return K3_OK;
}
