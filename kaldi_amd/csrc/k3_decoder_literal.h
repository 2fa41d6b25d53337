// k3_decoder_literal.h -- token passing that reproduces LatticeFasterDecoder's SERIAL algorithm bit for bit (k3_decoder_config.literal_order).
// Included by k3_decoder.hip inside its anonymous namespace (shares DecParams, Table, Shared, the wave helpers and finish_frame).
//
// The reference walks the previous frame's tokens in HashList order and tightens next_cutoff while it goes
// (decoder/lattice-faster-decoder.cc:779-797), so which arcs it accepts depends on that order, and the order depends on the order in which
// tokens were inserted into the hash (util/hash-list-inl.h:125-165), including the LIFO order of ProcessNonemitting's queue (:845-896).
// The default kernel applies the frame's FINAL bound instead (order-free, a sub-lattice).  This kernel computes the serial result without a
// serial walk; the phase structure is the one oracle/lattice_faster_oracle.cc proves equal to the serial code (its modes 2 / 3):
//   * visit order      = tokens sorted by (creation rank of their bucket's first occupant, own creation rank), bucket = state % hash_size
//                        (PossiblyResizeHash :227-233 tracked per lane): dense creation ranks by a bitmap prefix count, per-bucket first /
//                        count by atomics, a prefix sum over the bucket leaders -- no key sort;
//   * acceptance       = tot < min(pre-pass bound, min over EARLIER arcs of (tot + adaptive_beam)): a rejected arc never lowers the bound
//                        (tot >= bound => tot + beam >= bound), so the bound in force at arc j is an exclusive prefix-min over the arc sequence
//                        (tokens in visit order, arcs in FST order): per 64-token chunk minima, a scan over the chunks, a wave scan inside;
//   * creation time    = min sequence number over the accepted arcs into a state (atomicMin);
//   * eps closure      = the order-free fixpoint of the default kernel (costs, tokens, links), then a REPLAY of the LIFO queue by one
//                        wavefront on the closure sub-graph in token space, which only recovers the order in which tokens are created;
//   * final frame      = PruneForwardLinksFinal's in-place sweeps in token-list order with its 1e-5 ApproxEqual stop rule (:385-467), emulated
//                        token by token (k3_decode_prune_kernel, literal branch).
// Capacities: frame_tokens_cap <= 65536; arcs expanded + tokens created on one frame <= 32 * seq_words_cap.

constexpr int kRN = kHL / 2, kRA = 2048, kRS = 1024;      // replay fully in LDS: closure ids (cost + meta alias the level-1 table: kHL x 12 B) / passing arcs / stack entries
// (the hash-order passes use the same segment between replays: kHoLds below)      // replay in LDS: tokens / closure arcs / stack entries of a frame (larger
// frames: HBM scratch)
constexpr size_t kLitDynLds = (size_t)kRA * 8 + (size_t)kRS * 4;
constexpr unsigned kLabelNone = 0xFFFFFFFFu;
enum { kRfHasEps = 1, kRfExists = 2 };

// -DK3_LIT_PROF=1: cycles per phase of a frame (K3_LT);  =2: cycles per sub-phase of the two hash-order passes and the component replay (K3_LS)
// -DK3_LIT_PROF_MINTOK=a -DK3_LIT_PROF_MAXTOK=b: only frames built from a..b tokens are counted (sh.prof_n = the frame's n_cur)
#ifndef K3_LIT_PROF_MINTOK
#define K3_LIT_PROF_MINTOK 0
#endif
#ifndef K3_LIT_PROF_MAXTOK
#define K3_LIT_PROF_MAXTOK 0x7FFFFFFF
#endif
#define K3_LP_ON (sh.prof_n >= K3_LIT_PROF_MINTOK && sh.prof_n <= K3_LIT_PROF_MAXTOK)
#if defined(K3_LIT_PROF) && K3_LIT_PROF == 3      // =3: a finer split of the frame's phases (K3_LQ) instead of K3_LT's; a general frame's first phase also holds the LDS-path frames before it
#define K3_LT(i) do { } while (0)
#define K3_LS(i) do { } while (0)
#define K3_LQ(i) do { if (threadIdx.x == 0) { const long long now__ = (long long)__builtin_readcyclecounter(); if (K3_LP_ON) sh.prof[i] += now__ - lt_last__; lt_last__ = now__; } } while (0)
#elif defined(K3_LIT_PROF) && K3_LIT_PROF == 2
#define K3_LT(i) do { if (threadIdx.x == 0) lt_last__ = (long long)__builtin_readcyclecounter(); } while (0)
#define K3_LS(i) do { if (threadIdx.x == 0) { const long long now__ = (long long)__builtin_readcyclecounter(); if (K3_LP_ON) sh.prof[i] += now__ - lt_last__; lt_last__ = now__; } } while (0)
#elif defined(K3_LIT_PROF)
#define K3_LT(i) do { if (threadIdx.x == 0) { const long long now__ = (long long)__builtin_readcyclecounter(); if (K3_LP_ON) sh.prof[i] += now__ - lt_last__; lt_last__ = now__; } } while (0)
#define K3_LS(i) do { } while (0)
#else
#define K3_LT(i) do { } while (0)
#define K3_LS(i) do { } while (0)
#endif
#ifndef K3_LQ
#define K3_LQ(i) do { } while (0)
#endif

#ifndef K3_LIT_BASE_PRIO
#define K3_LIT_BASE_PRIO 2      // wave priority of the token-passing kernel (0 .. 3); A/B with -DK3_LIT_BASE_PRIO=0
#endif
#ifndef K3_COLD_INLINE
#define K3_COLD_INLINE __forceinline__      // (register-pressure experiments: -DK3_COLD_INLINE=__noinline__ makes the large-frame forms of the hash-order pass real calls)
#endif
// inclusive prefix sum over the wavefront with DPP row shifts / row broadcasts (no LDS-crossbar round trips)
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
  return v;
}

// exclusive prefix sum over n values produced by in(i), written to out[i]; returns the total.  All threads of the block call it.
template <typename In>
__device__ __forceinline__ int block_excl_scan(In &&in, unsigned *out, int n, int *redi) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (int)blockDim.x >> 6;
  const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x;
  if (per <= 4) {      // the usual frame: every thread takes `per` consecutive items (their loads are in flight together), one pass, two barriers
    const int b = tid * per; int x[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { x[k] = (k < per && b + k < n) ? (int)in(b + k) : 0; sum += x[k]; }
    const int incl = wave_incl_scan(sum);
    __syncthreads();
    if (lane == 63) redi[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < nw; w++) { const int s = redi[w]; if (w < wave) woff += s; tot += s; }
    int run = woff + incl - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) if (k < per && b + k < n) { out[b + k] = (unsigned)run; run += x[k]; }
    __syncthreads();
    return tot;
  }
  int carry = 0;
  for (int i0 = 0; i0 < n; i0 += (int)blockDim.x) {
    const int i = i0 + tid; const int x = i < n ? (int)in(i) : 0;
    const int incl = wave_incl_scan(x);
    __syncthreads();
    if (lane == 63) redi[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < nw; w++) { const int s = redi[w]; if (w < wave) woff += s; tot += s; }
    if (i < n) out[i] = (unsigned)(carry + woff + incl - x);
    carry += tot;
  }
  __syncthreads();
  return carry;
}

// the same over four counters at once: in(i) -> int4, out(i, exclusive sums); returns the totals.  red4: LDS, one int4 per wavefront.
template <typename In, typename Out>
__device__ __forceinline__ int4 block_excl_scan4(In &&in, Out &&out, int n, int4 *red4) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (int)blockDim.x >> 6;
  const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x;
  auto add = [](int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
  auto sub = [](int4 a, int4 b) { return make_int4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
  auto wscan = [](int4 a) { return make_int4(wave_incl_scan(a.x), wave_incl_scan(a.y), wave_incl_scan(a.z), wave_incl_scan(a.w)); };
  const int4 zero = make_int4(0, 0, 0, 0);
  if (per <= 4) {
    const int b = tid * per; int4 x[4], sum = zero;
#pragma unroll
    for (int k = 0; k < 4; k++) { x[k] = (k < per && b + k < n) ? in(b + k) : zero; sum = add(sum, x[k]); }
    const int4 incl = wscan(sum);
    __syncthreads();
    if (lane == 63) red4[wave] = incl;
    __syncthreads();
    int4 woff = zero, tot = zero;
    for (int w = 0; w < nw; w++) { const int4 s = red4[w]; if (w < wave) woff = add(woff, s); tot = add(tot, s); }
    int4 run = sub(add(woff, incl), sum);
#pragma unroll
    for (int k = 0; k < 4; k++) if (k < per && b + k < n) { out(b + k, run); run = add(run, x[k]); }
    __syncthreads();
    return tot;
  }
  int4 carry = zero;
  for (int i0 = 0; i0 < n; i0 += (int)blockDim.x) {
    const int i = i0 + tid; const int4 x = i < n ? in(i) : zero;
    const int4 incl = wscan(x);
    __syncthreads();
    if (lane == 63) red4[wave] = incl;
    __syncthreads();
    int4 woff = zero, tot = zero;
    for (int w = 0; w < nw; w++) { const int4 s = red4[w]; if (w < wave) woff = add(woff, s); tot = add(tot, s); }
    if (i < n) out(i, sub(add(add(carry, woff), incl), x));
    carry = add(carry, tot);
  }
  __syncthreads();
  return carry;
}

// wave_expand with the arc's position j in the wave's arc sequence (owner lanes in order, arcs of an owner in FST order); returns the number of arcs
template <typename F>
__device__ __forceinline__ int wave_expand_seq(const ArcRec *arcs, int beg, int deg, F &&f) {
  const int lane = threadIdx.x & 63;
  int incl = deg;
  incl = wave_incl_sum_i32(incl);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - deg;
  const unsigned long long has_arcs = __ballot(deg > 0);
  for (int j0 = 0; j0 < total; j0 += 64) {
    const int j = j0 + lane;
    const int lo = wave_owner_of(j0, deg, excl, has_arcs);
    const int a = __shfl(beg - excl, lo) + j; ArcRec r{};
    if (j < total) r = arcs[a];
    f(j < total, j, a, lo, r);
  }
  return total;
}

// wave_expand_seq for the frames that live in HBM (general path), software-pipelined: the arc records are requested TWO groups of 64 ahead and the log-likelihood of an arc's pdf ONE
// group ahead, so that a group's body (a chain of table round trips) starts with both in registers instead of behind two more dependent round trips.  f gets ll[r.pdf] as `llv`.
template <typename F>
__device__ __forceinline__ int wave_expand_seq_pf(const ArcRec *arcs, const float *ll, int beg, int deg, F &&f) {
  const int lane = threadIdx.x & 63;
  int incl = deg;
  incl = wave_incl_sum_i32(incl);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - deg, rel = beg - excl;
  const unsigned long long has_arcs = __ballot(deg > 0);
  auto locate = [&](int j0, int &a, int &lo) { lo = j0 < total ? wave_owner_of(j0, deg, excl, has_arcs) : 0; a = __shfl(rel, lo) + j0 + lane; };
  int a0, lo0, a1, lo1; ArcRec r0{}, r1{};
  locate(0, a0, lo0); if (lane < total) r0 = arcs[a0];
  locate(64, a1, lo1); if (64 + lane < total) r1 = arcs[a1];
  float l0 = lane < total ? ll[r0.pdf] : 0.0f;
  for (int j0 = 0; j0 < total; j0 += 64) {
    int a2, lo2; ArcRec r2{};
    locate(j0 + 128, a2, lo2); if (j0 + 128 + lane < total) r2 = arcs[a2];
    const float l1 = j0 + 64 + lane < total ? ll[r1.pdf] : 0.0f;
    f(j0 + lane < total, j0 + lane, a0, lo0, r0, l0);
    r0 = r1; a0 = a1; lo0 = lo1; l0 = l1; r1 = r2; a1 = a2; lo1 = lo2;
  }
  return total;
}

struct LitLane {      // this lane's slices of the literal_order scratch
  int *order[2], *by_ins, *dense, *grp, *ccnt, *cdst, *rflag, *rown, *stack;
  unsigned *label, *lead, *bm, *wpre, *cmin;
  float *c0, *cw, *rcost;
  int2 *crng, *arcs2;
  int4 *meta;
  int *iq, *c2t;
  int *par, *rtmp; int2 *rlist, *rinfo; int4 *cinfo, *coffs, *wrec;      // component replay (below)
  int4 *btab;     // bucket table of the hash-order pass of large frames
  int4 *vis;      // per visit position of the frame being expanded: (token, cost, first emitting arc, emitting arcs)
  __device__ LitLane(const DecParams &p, int L) {
    const long long cap = p.frame_tokens_cap, nch = cap / 64 + 2; const long long lb = p.lt_lane_bytes * L;
    auto at = [lb](auto *base) { return reinterpret_cast<decltype(base)>(reinterpret_cast<char *>(base) + lb); };
    order[0] = at(p.lt_order); order[1] = order[0] + cap; by_ins = at(p.lt_by_ins); dense = at(p.lt_dense); grp = at(p.lt_grp);
    label = at(p.lt_label); lead = at(p.lt_lead); bm = at(p.lt_bm); wpre = at(p.lt_wpre);
    cmin = at(p.lt_cmin); ccnt = at(p.lt_ccnt); c0 = at(p.lt_c0); crng = at(p.lt_crng); (void)nch;
    cdst = at(p.lt_cdst); cw = at(p.lt_cw); rcost = at(p.lt_rcost); rflag = at(p.lt_rflag); rown = at(p.lt_rown);
    stack = at(p.lt_stack); arcs2 = at(p.lt_arcs2); iq = at(p.lt_iq); meta = at(p.lt_meta); c2t = at(p.lt_c2t);
    par = at(p.lt_par);
    rtmp = at(p.lt_rtmp);
    rlist = at(p.lt_rlist);
    wrec = at(p.lt_wrec);
    cinfo = at(p.lt_cinfo);
    coffs = at(p.lt_coffs);
    rinfo = at(p.lt_rinfo);
    vis = at(p.lt_vis);
    btab = at(p.lt_btab);
  }
};

#ifdef K3_LIT_FORWARD      // the token-passing kernel and its helpers: compiled by k3_decoder_lit.hip (its own block size)
// HashList order of n tokens with unique creation labels < M: order_out[position] = token.  by_ins[d] = token with creation rank d.
// (The path of frames too large for lit_hash_order_lds.)  The buckets -- hash_size is unbounded, a frame touches at most n of them -- live in an
// open-addressing table of 16 B records {bucket, smallest creation rank, members, fill cursor} sized to the frame (2n .. 4n slots): one
// cache line per token and pass in a region that scales with the frame, where per-bucket arrays cost three lines spread over hash_size entries.
// dense_in (every form below): the labels already are dense creation ranks 0 .. n-1 (the frame's second pass), no ranking needed; relabel (the first pass): leave the
// tokens' dense ranks in q.label for that second pass.
__device__ K3_COLD_INLINE void lit_hash_order(const LitLane &q, Shared &sh, int n, unsigned M, const int *st, unsigned hash_size, int *order_out, bool dense_in, bool relabel,
                                              long long &lt_last__) {
  const int tid = threadIdx.x;
  const int W = dense_in ? 0 : (int)((M + 31u) >> 5);
  unsigned tsize = 1024; while (tsize < 2u * (unsigned)n) tsize <<= 1;      // <= 2 * next_pow2(cap) = the table's capacity
  const unsigned tmask = tsize - 1; int4 *tab = q.btab; int *slot_of = q.rtmp;
  for (unsigned s_ = tid; s_ < tsize; s_ += kBlock) tab[s_] = make_int4(-1, -1, 0, 0);
  if (!dense_in) for (int i = tid; i < n; i += kBlock) { const unsigned l = K3_ALD(&q.label[i]); k3a_or(&q.bm[l >> 5], 1u << (l & 31)); }
  __syncthreads();
  K3_LS(0);
  if (!dense_in) block_excl_scan([&](int w) { return __popc(K3_ALD(&q.bm[w])); }, q.wpre, W, sh.redi);
  K3_LS(1);
  for (int i = tid; i < n; i += kBlock) {
    const unsigned l = K3_ALD(&q.label[i]); int d = (int)l;
    if (!dense_in) { const unsigned wd = K3_ALD(&q.bm[l >> 5]); d = (int)(q.wpre[l >> 5] + (unsigned)__popc(wd & ((1u << (l & 31)) - 1u))); }
    q.dense[i] = d; q.by_ins[d] = i;
    const unsigned b = (unsigned)st[i] % hash_size; unsigned h = (b * 2654435761u) & tmask;
    for (;;) {
      unsigned *key = reinterpret_cast<unsigned *>(&tab[h]);
      const unsigned old = k3a_cas(key, 0xFFFFFFFFu, b);
      if (old == 0xFFFFFFFFu || old == b) break;
      h = (h + 1) & tmask;
    }
    slot_of[i] = (int)h;
    unsigned *rec = reinterpret_cast<unsigned *>(&tab[h]); k3a_min(&rec[1], (unsigned)d); k3a_add(&rec[2], 1u);
  }
  __syncthreads();
  K3_LS(2);
  block_excl_scan([&](int d) { const unsigned *rec = reinterpret_cast<const unsigned *>(&tab[slot_of[q.by_ins[d]]]); return K3_ALD(&rec[1]) == (unsigned)d ?
      K3_ALD(&rec[2]) : 0u; }, q.lead, n, sh.redi);
  K3_LS(3);
  for (int i = tid; i < n; i += kBlock) {
    unsigned *rec = reinterpret_cast<unsigned *>(&tab[slot_of[i]]);
    if (K3_ALD(&rec[2]) > 1u) { const unsigned lp = q.lead[K3_ALD(&rec[1])]; const unsigned s_ = k3a_add(&rec[3], 1u); q.grp[lp + s_] = q.dense[i]; }
  }
  __syncthreads();
  K3_LS(4);
  for (int i = tid; i < n; i += kBlock) {
    const unsigned *rec = reinterpret_cast<const unsigned *>(&tab[slot_of[i]]); const unsigned cnt = K3_ALD(&rec[2]), lp = q.lead[K3_ALD(&rec[1])];
    unsigned rank = 0;
    if (cnt > 1u) { const int d = q.dense[i]; for (unsigned k = 0; k < cnt; k++) rank += q.grp[lp + k] < d; }
    order_out[lp + rank] = i;
  }
  __syncthreads();
  K3_LS(5);
  if (!dense_in) for (int i = tid; i < n; i += kBlock) { const unsigned l = K3_ALD(&q.label[i]); K3_AST(&q.bm[l >> 5], 0u); }      // the label bitmap back to its idle pattern
  __syncthreads();
  if (relabel) { for (int i = tid; i < n; i += kBlock) K3_AST(&q.label[i], (unsigned)q.dense[i]); __syncthreads(); }
  K3_LS(6);
}

// The same for the usual frame (n <= kHoN tokens, labels < kHoM), entirely in LDS: label bitmap, leader offsets and bucket groups in `arena`
// (kHoLds bytes of the dynamic segment), and the buckets themselves -- hash_size is unbounded, a frame touches at most n of them -- in an
// open-addressing table keyed by the bucket number in `tab` (the level-1 state table's 3 x kHL words, dead at both call sites): key, smallest
// creation rank, member count.  A thread keeps its <= 4 tokens in registers across the phases; global memory is touched for the labels and
// states (one stream in) and the result (one stream out).  (Per-bucket arrays in HBM cost two cache-line round trips per token and call: they
// were half of the kernel's HBM traffic.)
constexpr int kHoN = 4 * kBlock < 2048 ? 4 * kBlock : 2048, kHoM = 32 * kBlock < 16384 ? 32 * kBlock : 16384;      // (four tokens per thread in registers, one bitmap word per thread)
constexpr size_t kHoLds = (size_t)(kHoM / 32) * 4 + (size_t)(kHoM / 32) * 2 + 2 * (size_t)kHoN * 2;
__device__ __forceinline__ void lit_hash_order_lds(const LitLane &q, Shared &sh, int n, unsigned M, const int *st, unsigned hash_size, int *order_out, bool write_by_ins,
                                                   bool dense_in, bool relabel, char *arena, int *tab, long long &lt_last__) {
  static_assert(kHoN <= 4 * kBlock && kHoM / 32 <= kBlock && 2 * kHoN <= kHL, "one pass per phase; table at most half full");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6; constexpr int nw = kBlock / 64;
  const int W = dense_in ? 0 : (int)((M + 31u) >> 5);
  unsigned *s_bm = reinterpret_cast<unsigned *>(arena);
  unsigned short *s_wpre = reinterpret_cast<unsigned short *>(s_bm + kHoM / 32), *s_lead = s_wpre + kHoM / 32, *s_grp = s_lead + kHoN;
  unsigned *b_key = reinterpret_cast<unsigned *>(tab), *b_first = b_key + kHL, *b_cnt = b_first + kHL;
  unsigned lab[4], bkt[4], lf[4], cnt[4]; int d[4], slot[4];
  for (int i = tid; i < kHL; i += kBlock) { b_key[i] = 0xFFFFFFFFu; b_first[i] = 0xFFFFFFFFu; b_cnt[i] = 0u; }
  if (tid < W) s_bm[tid] = 0u;
#pragma unroll
  for (int k = 0; k < 4; k++) { const int i = tid + k * kBlock; if (i < n) { lab[k] = K3_ALD(&q.label[i]); bkt[k] = (unsigned)st[i] % hash_size; } }
  __syncthreads();
  if (!dense_in) {
#pragma unroll
    for (int k = 0; k < 4; k++) { const int i = tid + k * kBlock; if (i < n) atomicOr(&s_bm[lab[k] >> 5], 1u << (lab[k] & 31)); }
    __syncthreads();
    K3_LS(0);
    {      // exclusive prefix count over the bitmap words (W <= kBlock: one word per thread)
      const int x = tid < W ? __popc(s_bm[tid]) : 0; const int incl = wave_incl_scan(x);
      if (lane == 63) sh.redi[wave] = incl;
      __syncthreads();
      int woff = 0;
      for (int w = 0; w < nw; w++) if (w < wave) woff += sh.redi[w];
      if (tid < W) s_wpre[tid] = (unsigned short)(woff + incl - x);
      __syncthreads();
    }
  }
  K3_LS(1);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = tid + k * kBlock;
    if (i < n) {
      const unsigned l = lab[k]; d[k] = dense_in ? (int)l : (int)s_wpre[l >> 5] + __popc(s_bm[l >> 5] & ((1u << (l & 31)) - 1u));
      if (relabel) K3_AST(&q.label[i], (unsigned)d[k]);
      unsigned h = (bkt[k] * 2654435761u) >> 20;      // 12 bits: kHL = 4096 slots
      for (;;) { const unsigned old = atomicCAS(&b_key[h], 0xFFFFFFFFu, bkt[k]); if (old == 0xFFFFFFFFu || old == bkt[k]) break; h = (h + 1) & (kHL - 1); }
      slot[k] = (int)h;
      atomicMin(&b_first[h], (unsigned)d[k]); atomicAdd(&b_cnt[h], 1u);
      if (write_by_ins) q.by_ins[d[k]] = i;
    }
  }
  __syncthreads();
  K3_LS(2);
  bool multi = false;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = tid + k * kBlock;
    if (i < n) {
      lf[k] = b_first[slot[k]];
      cnt[k] = b_cnt[slot[k]];
      s_lead[d[k]] = lf[k] == (unsigned)d[k] ? (unsigned short)cnt[k] : (unsigned short)0;
      multi |= cnt[k] > 1u;
    }
  }
  multi = __syncthreads_or(multi);
  {      // exclusive scan of the leaders' bucket sizes in creation order, in place (4 consecutive ranks per thread)
    const int b0 = tid * 4; int x[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { x[k] = b0 + k < n ? (int)s_lead[b0 + k] : 0; sum += x[k]; }
    const int incl = wave_incl_scan(sum);
    if (lane == 63) sh.redi[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < nw; w++) if (w < wave) run += sh.redi[w];
#pragma unroll
    for (int k = 0; k < 4; k++) if (b0 + k < n) { s_lead[b0 + k] = (unsigned short)run; run += x[k]; }
    __syncthreads();
  }
  K3_LS(3);
  if (multi) {      // buckets with several tokens: their members' ranks, grouped behind the leader's offset (b_first now counts up from the leader's rank: a cursor)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = tid + k * kBlock;
      if (i < n && cnt[k] > 1u) {
        const unsigned s_ = atomicAdd(&b_first[slot[k]], 1u) - lf[k];
        s_grp[s_lead[lf[k]] + s_] = (unsigned short)d[k];
      }
    }
    __syncthreads();
  }
  K3_LS(4);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = tid + k * kBlock;
    if (i < n) {
      const unsigned lp = s_lead[lf[k]]; unsigned rank = 0;
      if (cnt[k] > 1u) for (unsigned t = 0; t < cnt[k]; t++) rank += (int)s_grp[lp + t] < d[k];
      order_out[lp + rank] = i;
    }
  }
  __syncthreads();
  K3_LS(5);
}

static_assert(kHoLds <= kLitDynLds, "the hash-order arena lives in the replay's dynamic segment");
constexpr size_t kLitTabBytes = (size_t)3 * kHL * 4, kLitMarkBytes = (size_t)3 * (kHL / 32) * 4, kLitAuxBytes = (size_t)2 * 1024 * 4;
#include "k3_decoder_fast.h"
// HashList order for the frames between the two LDS-only forms above and the HBM form (lit_hash_order): up to kHmN tokens, labels below kHmM, hash_size <= 65535 -- 95 % of
// the frames the general path sees.  The structure of fast_hash_order (k3_decoder_fast.h: packed {bucket, smallest creation rank} table, members counted at the leader's
// rank) with every per-token value in an LDS array instead of registers (the general path runs at 128 registers per thread); `arena` = the 77.5 KB of the general path, all of it
// dead at both call sites.  Returns false when the frame does not fit (the caller takes lit_hash_order).
constexpr int kHmN = 3072, kHmM = 32768, kHmB = 4096;
constexpr size_t kHmLds = (size_t)kHmB * 4 + 7 * ((size_t)kHmN * 2 + 8) + (size_t)kHmM / 8 + (size_t)kHmM / 32 * 2;
__device__ K3_COLD_INLINE bool lit_hash_order_mid(const LitLane &q, Shared &sh, char *arena, int n, unsigned M, const int *st, unsigned hash_size,
    int *order_out, bool write_by_ins, bool dense_in, bool relabel) {
  if (n > kHmN || M > (unsigned)kHmM || hash_size > 65535u) return false;
  const int tid = threadIdx.x; const int W = dense_in ? 0 : (int)((M + 31u) >> 5);
  unsigned *btab = reinterpret_cast<unsigned *>(arena); char *a_ = arena + (size_t)kHmB * 4; constexpr size_t kCol = (size_t)kHmN * 2 + 8;
  unsigned short *lab16 = reinterpret_cast<unsigned short *>(a_), *bkt16 = reinterpret_cast<unsigned short *>(a_ + kCol),
      *dense16 = reinterpret_cast<unsigned short *>(a_ + 2 * kCol),
                 *lf16 = reinterpret_cast<unsigned short *>(a_ + 3 * kCol), *lead = reinterpret_cast<unsigned short *>(a_ + 4 * kCol),
                     *grp = reinterpret_cast<unsigned short *>(a_ + 5 * kCol),
                 *curs = reinterpret_cast<unsigned short *>(a_ + 6 * kCol);
  unsigned *bm = reinterpret_cast<unsigned *>(a_ + 7 * kCol); unsigned short *wpre = reinterpret_cast<unsigned short *>(a_ + 7 * kCol + (size_t)kHmM / 8);
  for (int i = tid; i < kHmB; i += kBlock) btab[i] = 0xFFFFFFFFu;
  for (int i = tid; i < W; i += kBlock) bm[i] = 0u;
  for (int i = tid; i < (n + 1) / 2 + 1; i += kBlock) { reinterpret_cast<unsigned *>(lead)[i] = 0u; reinterpret_cast<unsigned *>(curs)[i] = 0u; }
  for (int i = tid; i < n; i += kBlock) { lab16[i] = (unsigned short)K3_ALD(&q.label[i]); bkt16[i] = (unsigned short)((unsigned)st[i] % hash_size); }
  __syncthreads();
  if (!dense_in) {
    for (int i = tid; i < n; i += kBlock) { const unsigned l = lab16[i]; k3a_or(&bm[l >> 5], 1u << (l & 31)); }
    __syncthreads();
    block_excl_scan_f([&](int w) { return __popc(bm[w]); }, [&](int w, int ex) { wpre[w] = (unsigned short)ex; }, W, sh.redi);
  }
  for (int i = tid; i < n; i += kBlock) {
    const unsigned l = lab16[i], b = bkt16[i]; const unsigned d = dense_in ? l : (unsigned)wpre[l >> 5] + (unsigned)__popc(bm[l >> 5] & ((1u << (l & 31)) - 1u));
    dense16[i] = (unsigned short)d; if (write_by_ins) q.by_ins[d] = i;
    if (relabel) K3_AST(&q.label[i], d);
    const unsigned mine = (b << 16) | d; unsigned h = (b * 2654435761u) >> 20;      // 12 bits: kHmB = 4096
    for (;;) {
      unsigned w = lds_ld(&btab[h]);
      if (w == 0xFFFFFFFFu) { const unsigned old = k3a_cas(&btab[h], 0xFFFFFFFFu, mine); if (old == 0xFFFFFFFFu) break; w = old; }
      if ((w >> 16) == b) { k3a_min(&btab[h], mine); break; }
      h = (h + 1) & (kHmB - 1);
    }
    lf16[i] = (unsigned short)h;      // (the slot for now; the leader's rank once every member has arrived)
  }
  __syncthreads();
  for (int i = tid; i < n; i += kBlock) { const unsigned lf = btab[lf16[i]] & 0xFFFFu; lf16[i] = (unsigned short)lf; add16(lead, (int)lf, 1u); }
  __syncthreads();
  bool multi = false;
  for (int i = tid; i < n; i += kBlock) { const unsigned c = lead[lf16[i]]; bkt16[i] = (unsigned short)c; multi |= c > 1u; }      // (bkt16 is free now: the bucket's member count)
  multi = __syncthreads_or(multi);
  block_excl_scan_f([&](int r) { return (int)lead[r]; }, [&](int r, int ex) { lead[r] = (unsigned short)ex; }, n, sh.redi);
  if (multi) {
    for (int i = tid; i < n; i += kBlock) if (bkt16[i] > 1u) { const unsigned lf = lf16[i]; const unsigned s_ = add16(curs, (int)lf, 1u); grp[lead[lf] + s_] = dense16[i]; }
    __syncthreads();
  }
  for (int i = tid; i < n; i += kBlock) {
    const unsigned lp = lead[lf16[i]], c = bkt16[i]; unsigned rank = 0;
    if (c > 1u) { const unsigned d = dense16[i]; for (unsigned t = 0; t < c; t++) rank += grp[lp + t] < d; }
    order_out[lp + rank] = i;
  }
  __syncthreads();
  return true;
}
constexpr size_t kLitGeneralLds = kLitTabBytes + kLitMarkBytes + kLitAuxBytes + kLitDynLds;
static_assert(kHmLds <= kLitTabBytes + kLitMarkBytes + kLitAuxBytes + kLitDynLds, "the mid-size hash order works in the general path's arena");
constexpr size_t kLitArena = kLitGeneralLds > (size_t)kFastArena ? kLitGeneralLds : (size_t)kFastArena;
struct LitShared { int n_csr, use_lds, n_created, m_e; unsigned final_cut; };

// HashList order for the frames above lit_hash_order_mid (the head of an utterance: 3 k .. 25 k tokens on every lane at the same time) WITHOUT global atomics.  The HBM form
// (lit_hash_order) spends ~6 read-modify-write operations per token at the L2, and with all lanes in their large frames together the chip's atomic rate -- not latency --
// is what those frames wait for (tools/prof_frames.py: the same frame takes half the cycles with 128 lanes resident instead of 512).  Here every atomic is an LDS atomic:
//   * creation ranks: the label bitmap of a RANGE of kHbW x 32 = 32 768 labels in LDS, prefix counts over its words, ranks to q.dense (the head frames of an utterance take two to
//     four ranges, so the range loop is exercised by every decode of the bench configuration; a range costs two coalesced sweeps over the labels);
//   * the tokens' records {bucket, rank | token} are grouped into P = 2^k PARTITIONS by the top bits of the bucket's hash (<= 2048 tokens each, ~1000 on average): a histogram and
//     per-partition cursors in LDS, one 8-byte store per token;
//   * a partition at a time: its records in registers (<= 4 per thread), its buckets {key, smallest rank, members | cursor} in an LDS table of kHbT slots, the ranks of the
//     members of shared buckets in an LDS list; the leader publishes the bucket's size at its rank, every token gets {leader's rank, members created before it};
//   * a scan over the leaders' sizes in creation order gives the buckets' offsets; position = offset of the leader + place inside the bucket.
// Global memory sees streams plus, per token, the record store, the leader's size (one store per bucket), one load of the leader's offset and the result.
// Returns false when a partition does not fit (the caller takes lit_hash_order): the result is then untouched.
constexpr int kHbT = 4096, kHbW = 1024, kHbMaxPart = 4 * kBlock < 2048 ? 4 * kBlock : 2048, kHbPart = kHbMaxPart * 3 / 4, kHbMaxP = 64;
constexpr size_t kHbLdsA = (size_t)kHbW * 4 + (size_t)kHbW * 2, kHbLdsB = (size_t)kHbT * 12 + (size_t)kHbT * 2 + (size_t)kHbMaxPart * 2,
    kHbLdsP = kHbLdsB > kHbLdsA ? kHbLdsB : kHbLdsA;
static_assert(kHbLdsP + 2 * kHbMaxP * 4 <= kLitGeneralLds, "the large-frame hash order works in the general path's arena");
static_assert(kHbMaxPart <= 4 * kBlock, "a partition's records fit the registers of one pass");
__device__ K3_COLD_INLINE bool lit_hash_order_big(const LitLane &q, Shared &sh, char *arena, int n, unsigned M, const int *st, unsigned hash_size,
    int *order_out, bool write_by_ins, bool dense_in, bool relabel,
                                                  long long &lt_last__) {
  const int tid = threadIdx.x;
  int *dense = q.dense, *bkt = q.grp, *lr = q.rtmp; unsigned *lead = q.lead; int2 *rec = q.rlist;
  int *pcnt = reinterpret_cast<int *>(arena + kHbLdsP), *pbase = pcnt + kHbMaxP;      // partition sizes -> cursors; first record of each partition
  int lgp = 0; while ((n >> lgp) > kHbPart) lgp++;
  if (lgp > 6) return false;
  const int P = 1 << lgp;
  auto part_of = [&](unsigned h) { return lgp ? (int)(h >> (32 - lgp)) : 0; };
  auto start_of = [&](unsigned h) { return (h >> (20 - lgp)) & (unsigned)(kHbT - 1); };
  if (tid < kHbMaxP) pcnt[tid] = 0;
  if (dense_in) {      // ---- the labels are the creation ranks: one sweep for the ranks, the buckets and the partition counts
    __syncthreads();
    for (int i0 = tid; i0 < n; i0 += 4 * kBlock) {
      unsigned l[4]; int s_[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { const int i = i0 + k * kBlock; l[k] = 0u; s_[k] = 0; if (i < n) { l[k] = K3_ALD(&q.label[i]); s_[k] = st[i]; } }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = i0 + k * kBlock;
        if (i < n) {
          dense[i] = (int)l[k]; if (write_by_ins) q.by_ins[l[k]] = i;
          const unsigned b = (unsigned)s_[k] % hash_size; bkt[i] = (int)b; lead[i] = 0u; k3a_add(&pcnt[part_of(b * 2654435761u)], 1);
        }
      }
    }
    __syncthreads();
  } else {      // ---- creation ranks (label ranges of kHbW words); the first range's sweep also computes the buckets and counts the partitions
    unsigned *bm = reinterpret_cast<unsigned *>(arena); unsigned short *wpre = reinterpret_cast<unsigned short *>(bm + kHbW);
    int base = 0;
    for (unsigned l0 = 0; l0 < M; l0 += (unsigned)kHbW * 32u) {
      const unsigned span = M - l0 < (unsigned)kHbW * 32u ? M - l0 : (unsigned)kHbW * 32u; const int W = (int)((span + 31u) >> 5);
      for (int i = tid; i < W; i += kBlock) bm[i] = 0u;
      __syncthreads();
      for (int i0 = tid; i0 < n; i0 += 4 * kBlock) {
        unsigned l[4]; int s_[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int i = i0 + k * kBlock; l[k] = 0xFFFFFFFFu; s_[k] = 0; if (i < n) { l[k] = K3_ALD(&q.label[i]) - l0; if (l0 == 0u) s_[k] = st[i]; } }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int i = i0 + k * kBlock;
          if (i < n) {
            if (l[k] < span) k3a_or(&bm[l[k] >> 5], 1u << (l[k] & 31));
            if (l0 == 0u) { const unsigned b = (unsigned)s_[k] % hash_size; bkt[i] = (int)b; lead[i] = 0u; k3a_add(&pcnt[part_of(b * 2654435761u)], 1); }
          }
        }
      }
      __syncthreads();
      const int tot = block_excl_scan_f([&](int w) { return __popc(bm[w]); }, [&](int w, int ex) { wpre[w] = (unsigned short)ex; }, W, sh.redi);
      for (int i0 = tid; i0 < n; i0 += 4 * kBlock) {
        unsigned l[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int i = i0 + k * kBlock; l[k] = i < n ? K3_ALD(&q.label[i]) - l0 : 0xFFFFFFFFu; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int i = i0 + k * kBlock;
          if (l[k] < span) { const int d = base + (int)wpre[l[k] >> 5] + __popc(bm[l[k] >> 5] & ((1u << (l[k] & 31)) - 1u)); dense[i] = d; if (write_by_ins) q.by_ins[d] = i; }
        }
      }
      base += tot;
      __syncthreads();
    }
  }
  K3_LS(0);
  // ---- records grouped by partition
  if (tid < 64) {      // (one wavefront: exclusive scan of the <= 64 partition sizes; a partition beyond the register budget of a pass -> the HBM form)
    const int c = tid < P ? pcnt[tid] : 0; const int incl = wave_incl_scan(c);
    pbase[tid] = incl - c; pcnt[tid] = incl - c;      // (pcnt becomes the cursor)
    if (c > kHbMaxPart) sh.flag = 1;
  }
  __syncthreads();
  if (sh.flag) { __syncthreads(); if (tid == 0) sh.flag = 0; __syncthreads(); return false; }
  for (int i0 = tid; i0 < n; i0 += 4 * kBlock) {
    int b[4], d[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int i = i0 + k * kBlock; b[k] = 0; d[k] = 0; if (i < n) { b[k] = bkt[i]; d[k] = dense[i]; } }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = i0 + k * kBlock;
      if (i < n) {
        const int pos = k3a_add(&pcnt[part_of((unsigned)b[k] * 2654435761u)], 1);
        rec[pos] = make_int2(b[k], (int)(((unsigned)d[k] << 16) | (unsigned)i));
      }
    }
  }
  __syncthreads();
  K3_LS(1);
  // ---- buckets, partition by partition
  unsigned *key = reinterpret_cast<unsigned *>(arena), *mind = key + kHbT, *cnt = mind + kHbT;
  unsigned short *moff = reinterpret_cast<unsigned short *>(cnt + kHbT), *mem = moff + kHbT;
  for (int pt = 0; pt < P; pt++) {
    const int pb = pbase[pt], np_ = (pt + 1 < P ? pbase[pt + 1] : n) - pb;
    int2 r_[4]; unsigned slot[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int j = tid + k * kBlock; r_[k] = make_int2(0, 0); if (j < np_) r_[k] = rec[pb + j]; }
    for (int i = tid; i < kHbT; i += kBlock) { key[i] = 0xFFFFFFFFu; mind[i] = 0xFFFFFFFFu; cnt[i] = 0u; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {      // the partition's buckets: {key, smallest creation rank, members}
      if (tid + k * kBlock < np_) {
        const unsigned b = (unsigned)r_[k].x; unsigned s_ = start_of(b * 2654435761u);
        // (<= 2048 keys in 4096 slots: always ends)
        for (;;) {
          const unsigned old = k3a_cas(&key[s_], 0xFFFFFFFFu, b);
          if (old == 0xFFFFFFFFu || old == b) break;
          s_ = (s_ + 1) & (unsigned)(kHbT - 1);
        }
        slot[k] = s_; k3a_min(&mind[s_], (unsigned)r_[k].y >> 16); k3a_add(&cnt[s_], 1u);
      }
    }
    __syncthreads();
    const int tm = block_excl_scan_f([&](int s_) { const unsigned c = cnt[s_]; return c > 1u ? (int)c : 0; },
        [&](int s_, int ex) { moff[s_] = (unsigned short)ex; }, kHbT, sh.redi);
#pragma unroll
    for (int k = 0; k < 4; k++) {      // the leader publishes its bucket's size; members of shared buckets line up behind their bucket's offset
      if (tid + k * kBlock < np_) {
        const unsigned s_ = slot[k], d = (unsigned)r_[k].y >> 16; const unsigned c = lds_ld(&cnt[s_]) & 0xFFFFu, lf = mind[s_];
        if (lf == d) lead[lf] = c;
        if (c > 1u) { const unsigned pos = k3a_add(&cnt[s_], 0x10000u) >> 16; mem[moff[s_] + pos] = (unsigned short)d; }
        else lr[pb + tid + k * kBlock] = (int)(lf << 16);
      }
    }
    __syncthreads();
    if (tm > 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) {      // position inside a shared bucket = members created earlier
        if (tid + k * kBlock < np_) {
          const unsigned s_ = slot[k], d = (unsigned)r_[k].y >> 16, c = cnt[s_] & 0xFFFFu;
          if (c > 1u) {
            const unsigned short *m_ = mem + moff[s_];
            unsigned rank = 0;
            for (unsigned t = 0; t < c; t++) rank += (unsigned)m_[t] < d;
            lr[pb + tid + k * kBlock] = (int)((mind[s_] << 16) | rank);
          }
        }
      }
    }
    __syncthreads();
  }
  K3_LS(2);
  // ---- leaders' bucket sizes in creation order -> offsets; position = leader's offset + own place in the bucket
  block_excl_scan([&](int r) { return lead[r]; }, lead, n, sh.redi);
  K3_LS(3);
  for (int j0 = tid; j0 < n; j0 += 4 * kBlock) {
    int v[4], t_[4]; unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int j = j0 + k * kBlock; v[k] = 0; t_[k] = 0; if (j < n) { v[k] = lr[j]; t_[k] = rec[j].y & 0xFFFF; } }
#pragma unroll
    for (int k = 0; k < 4; k++) { const int j = j0 + k * kBlock; o[k] = j < n ? lead[(unsigned)v[k] >> 16] : 0u; }
#pragma unroll
    for (int k = 0; k < 4; k++) { const int j = j0 + k * kBlock; if (j < n) order_out[o[k] + ((unsigned)v[k] & 0xFFFFu)] = t_[k]; }
  }
  __syncthreads();
  if (relabel) { for (int i = tid; i < n; i += kBlock) K3_AST(&q.label[i], (unsigned)dense[i]); __syncthreads(); }
  K3_LS(5);
  return true;
}

// The replay loop.  MODE 0: costs, meta, arcs, stack and the creation list in LDS; MODE 1: costs, stack and list in LDS, meta / arcs read-only
// in HBM; MODE 2: everything in HBM.  rcost[id] = the cost the serial code has seen for the token so far (+inf: not created yet);
// meta[id] = {first passing arc, passing arcs, destination and weight of the first one}; clist[k] = id of the k-th token the closure creates.
#define K3_U(x) __builtin_amdgcn_readfirstlane(x)      /* a wave-uniform int: pin it to a scalar register so that its arithmetic and the branches on it run on the scalar unit */
__device__ __forceinline__ float k3_uf(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
template <int MODE>
__device__ __forceinline__ void lit_replay(float *rcost, const int4 *meta, const int2 *AR, int *stack, int stack_cap, unsigned *clist, const int *iq, int n_iq, float accept,
                                           int *s_own, int *out_created, int *out_err, int *out_pops) {
  const int lane = threadIdx.x & 63; const float kInf = __builtin_inff();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int sp = 0, kpos = K3_U(n_iq), created = 0, err = 0, e_fwd = -1, cbase = -64, cache = 0, iters = 0;      // all wave-uniform except `cache`
  accept = k3_uf(accept);
  for (;; iters++) {
    if (iters > (1 << 24)) { err = 2; break; }      // cannot happen (every pop follows a cost decrease); keeps a bug from hanging the GPU
    int e;
    if (e_fwd >= 0) { e = e_fwd; e_fwd = -1; }
    else if (sp > 0) { sp = sp - 1; e = K3_U(stack[sp]); }
    else if (kpos > 0) {
      kpos = kpos - 1;
      if ((kpos & ~63) != cbase) { cbase = kpos & ~63; cache = cbase + lane < n_iq ? iq[cbase + lane] : 0; }
      e = __builtin_amdgcn_readlane(cache, kpos & 63);
    } else break;
    const float c = k3_uf(rcost[e]); const int4 mt4 = meta[e];
    const int pc = K3_U(mt4.y), abeg = K3_U(mt4.x);
    if (!(c < accept) || pc == 0) continue;
    if (pc == 1) {
      // ---- one passing arc (the common case): the whole pop is scalar work; every lane computes the same values, lane 0 stores
      const int d = K3_U(mt4.z); const float tot = c + k3_uf(__int_as_float(mt4.w));
      if (tot < accept) {
        const float old = k3_uf(rcost[d]);
        if (old > tot) {
          if (old == kInf) { if (lane == 0) clist[created] = (unsigned)d; created++; }
          if (lane == 0) rcost[d] = tot;
          if (K3_U(meta[d].y) > 0) e_fwd = d;      // pushed and popped again at once
          if (MODE == 2) __threadfence_block();
        }
      }
      continue;
    }
    for (int k0 = 0; k0 < pc; k0 += 64) {
      if (e_fwd >= 0) { if (sp + 1 > stack_cap) { err = 1; break; } if (lane == 0) stack[sp] = e_fwd; sp++; e_fwd = -1; if (MODE == 2) __threadfence_block(); }
      const int k = k0 + lane; const bool v = k < pc;
      const int2 ar = v ? AR[abeg + k] : make_int2(0, 0);
      const int d = ar.x; const float tot = c + __int_as_float(ar.y); const bool ok = v && tot < accept;
      const unsigned long long okm = __ballot(ok);
      if (okm == 0ull) continue;
      bool clash = false;
      if (__popcll(okm) >= 2) {      // two arcs of this batch into the same token must be applied one after the other
        volatile int *own = s_own;      // volatile: the compiler must not forward a lane's own store to its load (another lane's store may have landed in between)
        if (ok) own[d & 1023] = lane;
        __builtin_amdgcn_wave_barrier();
        clash = __ballot(ok && own[d & 1023] != lane) != 0ull;      // (LDS executes a wavefront's accesses in order)
      }
      if (!clash) {
        const float old = ok ? rcost[d] : 0.0f; const int dpc = ok ? meta[d].y : 0;
        const bool isnew = ok && old == kInf, changed = ok && old > tot;      // a token's cost is < cutoff < inf once it exists
        const unsigned long long nm = __ballot(isnew);
        if (isnew) clist[created + __popcll(nm & lt_mask)] = (unsigned)d;
        created += __popcll(nm);
        if (changed) rcost[d] = tot;
        const unsigned long long pm = __ballot(changed && dpc > 0);      // only tokens that can expand are queued
        if (pm) {
          const int np = __popcll(pm), top = 63 - __clzll((long long)pm);
          if (sp + np > stack_cap) { err = 1; break; }
          if ((pm >> lane & 1ull) && lane != top) stack[sp + __popcll(pm & lt_mask)] = d;
          sp += np - 1; e_fwd = __builtin_amdgcn_readlane(d, top);
        }
      } else {
        for (unsigned long long rest = okm; rest; rest &= rest - 1) {
          const int jl = __ffsll((long long)rest) - 1;
          const int dj = __builtin_amdgcn_readlane(d, jl);
          const float tj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), jl));
          const float old = k3_uf(rcost[dj]);
          const bool isnew = old == kInf, changed = old > tj;
          if (isnew) { if (lane == 0) clist[created] = (unsigned)dj; created++; }
          if (changed && lane == 0) rcost[dj] = tj;
          if (changed && K3_U(meta[dj].y) > 0) { if (sp + 1 > stack_cap) { err = 1; break; } if (lane == 0) stack[sp] = dj; sp++; }
          if (MODE == 2) __threadfence_block();
        }
        if (err) break;
      }
      if (MODE == 2) __threadfence_block();
    }
    if (err) break;
  }
  *out_created = created; *out_err = err; *out_pops = iters;
}

// ---- The replay split by weakly connected COMPONENTS of the closure sub-graph (oracle mode 4).  The serial code consumes its queue root by
// root (a root = an entry of the initial queue; the stack is back at its initial level before the next root is taken) and a root's cascade
// reads and writes only the costs of tokens it can reach, so cascades in different components commute.  Every component replays its own
// roots in queue order on ONE THREAD with a stack of its own (hundreds of components per frame, the longest chain a dozen pops), and the
// creation labels are handed out afterwards root by root in queue order by a prefix sum over the roots' creation counts.
// words updated by atomics (at L2): not through the L1
__device__ __forceinline__ int4 k3_ald4(const int4 *p_) {
  const int *w = reinterpret_cast<const int *>(p_);
  return make_int4(K3_ALD(&w[0]), K3_ALD(&w[1]), K3_ALD(&w[2]), K3_ALD(&w[3]));
}
template <typename P> __device__ __forceinline__ int uf_find(P par, int x) { for (;;) { const int q_ = K3_ALD(&par[x]); if (q_ == x) return x; x = q_; } }
// lock-free union: the larger root hooks under the smaller one (parents only ever decrease: no cycles)
template <typename P> __device__ __forceinline__ void uf_union(P par, int a, int b) {
  for (;;) {
    a = uf_find(par, a); b = uf_find(par, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    if (k3a_cas(&par[a], a, b) == a) return;
  }
}

// One worker = one component, on one thread.  rcost / meta / AR / clist as in lit_replay (LDS or HBM by instantiation); stk: this component's
// slice of the stack pool.  Returns false when the stack slice overflows (the frame then takes the serial replay).
template <typename RC, typename MT, typename AT, typename CL>
__device__ __forceinline__ bool lit_replay_component(RC rcost, MT meta, AT AR, CL clist, const int2 *rlist, int2 *rinfo, int *stk, int scap,
                                                     int r0, int rcnt, int cpos, float accept) {
  const float kInf = __builtin_inff();
  int2 root = rlist[r0];      // {position in the initial queue, closure id}
  for (int j = 0; j < rcnt; j++) {
    const int k = root.x; int cur = root.y; const int seg0 = cpos; int sp = 0;
    if (j + 1 < rcnt) root = rlist[r0 + j + 1];      // (in flight while this root's cascade runs)
    for (;;) {
      const float cc = rcost[cur]; const int4 mt = meta[cur];
      int nxt = -1;      // the newest push stays in a register: it is the next pop
      if (cc < accept) {
        for (int a = 0; a < mt.y; a++) {
          int d = mt.z, wb = mt.w;
          if (a > 0) { const int2 ar = AR[mt.x + a]; d = ar.x; wb = ar.y; }
          const float tot = cc + __int_as_float(wb);
          if (tot < accept) {
            const float old = rcost[d]; const int dpc = meta[d].y;      // (both behind `d`, requested together)
            if (old > tot) {
              if (old == kInf) clist[cpos++] = (unsigned)d;
              rcost[d] = tot;
              if (dpc > 0) { if (nxt >= 0) { if (sp >= scap) return false; stk[sp++] = nxt; } nxt = d; }
            }
          }
        }
      }
      if (nxt >= 0) cur = nxt; else if (sp > 0) cur = stk[--sp]; else break;
    }
    rinfo[k] = make_int2(seg0, cpos - seg0);
  }
  return true;
}

// All phases of the component replay; every thread of the block calls it.  par: n_cid ints (LDS for small frames), par[c] = c and cinfo[c] = 0
// on entry (the caller's step 2 sets them).  Returns the number of tokens created, or -1 when the frame has to take the serial replay (rcost /
// clist are then in an undefined state).  Loops over roots / components keep up to four items per thread in flight (a phase costs its chain of
// dependent round trips once, not once per item).
template <typename RC, typename MT, typename AT, typename CL, typename P>
__device__ __forceinline__ int lit_replay_components(const DecParams &p, const LitLane &q, Shared &sh, int *s_flag, RC rcost, MT meta, AT AR, CL clist, P par,
                                                     int n_cid, int n_arc, int n_iq, unsigned m_e, float accept, long long &lt_last__) {
  const int tid = threadIdx.x; const float kInf = __builtin_inff();
  int4 *red4 = reinterpret_cast<int4 *>(sh.hist);
  if (n_arc > p.stack_cap) return -1;
  for (int c = tid; c < n_cid; c += kBlock) {
    const int4 mt = meta[c];
    for (int a = 0; a < mt.y; a++) uf_union(par, c, a == 0 ? mt.z : AR[mt.x + a].x);
  }
  __syncthreads();
  K3_LS(8);
  // per component (indexed by its root id): roots of the initial queue, tokens the closure creates, passing arcs
  for (int c = tid; c < n_cid; c += kBlock) {
    const int r = uf_find(par, c); if (r != c) K3_AST(&par[c], r);
    int *ci = reinterpret_cast<int *>(&q.cinfo[r]);
    if (rcost[c] == kInf) k3a_add(&ci[1], 1);
    const int pc = meta[c].y; if (pc > 0) k3a_add(&ci[2], pc);
  }
  for (int k0 = tid; k0 < n_iq; k0 += 4 * kBlock) {
    int e[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int k = k0 + u * kBlock; if (k < n_iq) e[u] = q.iq[k]; }
#pragma unroll
    for (int u = 0; u < 4; u++) { const int k = k0 + u * kBlock; if (k < n_iq) k3a_add(reinterpret_cast<int *>(&q.cinfo[uf_find(par, e[u])]), 1); }
  }
  __syncthreads();
  K3_LS(9);
  // offsets per component; a component with roots is a worker: its record {first root slot, roots, first creation slot, first stack slot | stack slots}
  const int4 tot = block_excl_scan4([&](int c) { const int4 ci = k3_ald4(&q.cinfo[c]); return make_int4(ci.x, ci.y, ci.z, ci.x > 0 ? 1 : 0); },
                                    [&](int c, int4 ex) { q.coffs[c] = ex; }, n_cid, red4);
  const int n_workers = tot.w;
#if K3_LIT_CAPTURE
  if (p.cap && tid == 0) p.cap[6] = n_workers;
#endif
  K3_LS(10);
  // a component's roots in queue order (the queue is consumed from its back: descending k).  A component with one root (the usual case) is done
  // in one step; the others collect their roots, then rank them.
  bool multi = false;
  for (int k0 = tid; k0 < n_iq; k0 += 2 * kBlock) {
    int e[2], r[2]; int4 ci[2], co[2];
#pragma unroll
    for (int u = 0; u < 2; u++) { const int k = k0 + u * kBlock; if (k < n_iq) e[u] = q.iq[k]; }
#pragma unroll
    for (int u = 0; u < 2; u++) { const int k = k0 + u * kBlock; if (k < n_iq) { r[u] = uf_find(par, e[u]); ci[u] = k3_ald4(&q.cinfo[r[u]]); co[u] = q.coffs[r[u]]; } }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int k = k0 + u * kBlock;
      if (k < n_iq) {
        if (ci[u].x == 1) q.rlist[co[u].x] = make_int2(k, e[u]);
        else { multi = true; const int pos = k3a_add(reinterpret_cast<int *>(&q.cinfo[r[u]]) + 3, 1); q.rtmp[co[u].x + pos] = k; }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {      // the worker's record (a 1-root component: by its root; the others: by their first root in queue order, below)
      const int k = k0 + u * kBlock;
      if (k < n_iq && ci[u].x == 1) { q.wrec[2 * co[u].w] = make_int4(co[u].x, 1, co[u].y, co[u].z); q.wrec[2 * co[u].w + 1] = make_int4(ci[u].z, 0, 0, 0); }
    }
  }
  multi = __syncthreads_or(multi);
  if (multi) {
    for (int k = tid; k < n_iq; k += kBlock) {
      const int e = q.iq[k]; const int r = uf_find(par, e); const int4 ci = k3_ald4(&q.cinfo[r]), co = q.coffs[r];
      if (ci.x > 1) {
        int rank = 0;
        for (int t = 0; t < ci.x; t++) rank += q.rtmp[co.x + t] > k;
        q.rlist[co.x + rank] = make_int2(k, e);
        if (rank == 0) { q.wrec[2 * co.w] = make_int4(co.x, ci.x, co.y, co.z); q.wrec[2 * co.w + 1] = make_int4(ci.z, 0, 0, 0); }
      }
    }
    __syncthreads();
  }
  K3_LS(11);
  for (int w = tid; w < n_workers; w += kBlock) {
    const int4 w0 = q.wrec[2 * w], w1 = q.wrec[2 * w + 1];
    // (literal_order = 3: no stack slices, a test hook for the fall-back)
    if (!lit_replay_component(rcost, meta, AR, clist, q.rlist, q.rinfo, q.stack + w0.w, p.literal == 3 ? 0 : w1.x, w0.x, w0.y, w0.z, accept)) *s_flag = 1;
  }
  __syncthreads();
  K3_LS(12);
  if (*s_flag) return -1;
  // creation labels: roots in queue order (j-th root processed = position n_iq - 1 - j), tokens of a root in the order it created them
  const int created = block_excl_scan([&](int j) { return (unsigned)q.rinfo[n_iq - 1 - j].y; }, reinterpret_cast<unsigned *>(q.dense), n_iq, sh.redi);
  for (int j0 = tid; j0 < n_iq; j0 += 4 * kBlock) {
    int2 ri[4]; unsigned base[4]; int t0[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int j = j0 + u * kBlock; ri[u] = make_int2(0, 0); if (j < n_iq) { ri[u] = q.rinfo[n_iq - 1 - j]; base[u] = m_e + (unsigned)q.dense[j]; } }
#pragma unroll
    for (int u = 0; u < 4; u++) if (ri[u].y > 0) t0[u] = q.c2t[clist[ri[u].x]];      // (most roots create one token)
#pragma unroll
    for (int u = 0; u < 4; u++) for (int t = 0; t < ri[u].y; t++) K3_AST(&q.label[t == 0 ? t0[u] : q.c2t[clist[ri[u].x + t]]], base[u] + (unsigned)t);
  }
  K3_LS(13);
  return created;
}

// InitDecoding from the decoder's template (DecParams::tpl_*), launched in front of the token-passing kernel: a fresh lane (one workgroup each) gets exactly what that
// kernel's f == -1 pass would leave -- tokens, links, visit order (in the half of lt_order that pass writes), creation order, offsets, counters -- as the state of a lane that
// continues after zero frames, and its `fresh` flag is cleared, so the token-passing kernel resumes it like any chunked AdvanceDecoding call.  A lane whose previous utterance
// failed (its scratch needs the token-passing kernel's clean-up) or whose pools are smaller than the template stays fresh.  (A kernel of its own, not a branch of the
// token-passing kernel: that kernel lives at 128 registers with ~50 spilled, and the branch cost its frames 1.7 %.)
__global__ __launch_bounds__(256) void k3_decode_init_from_template_kernel(DecParams p, int *fresh) {
  const int L = p.q_lanes ? p.q_lanes[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
  if (!fresh[L] || p.tpl_n <= 0) return;
  LaneInfo &li = p.info[L];
  const LanePool lp = p.pools[L];
  if (li.status < 0 || lp.tcap < p.tpl_n || lp.lcap < p.tpl_nl) return;      // (uniform over the workgroup)
  const LitLane q(p, L);
  const int n_t = p.tpl_n, n_l = p.tpl_nl; int *ord_t = q.order[1];      // (a fresh lane starts with order_sel = 0 and its InitDecoding pass writes the other half)
  for (int i = tid; i < n_t; i += 256) { lp.tok_state[i] = p.tpl_state[i]; lp.tok_cost[i] = p.tpl_cost[i]; ord_t[i] = p.tpl_order[i]; q.by_ins[i] = p.tpl_by_ins[i]; }
  for (int l = tid; l < n_l; l += 256) { lp.links[l] = p.tpl_links[l]; lp.link_arc[l] = p.tpl_arc[l]; }
  __syncthreads();
  if (tid == 0) {
    long long *tok_off = p.tok_off + L * p.fstride, *loff_e = p.link_off_e + L * p.fstride, *loff_n = p.link_off_n + L * p.fstride;
    tok_off[0] = 0; tok_off[1] = n_t; loff_n[0] = 0; loff_e[0] = n_l;
    li.n_tokens = n_t; li.n_links = n_l; li.n_cands = 0; li.n_eps = p.tpl_eps; li.max_frame_tokens = n_t; li.status = kStOk; li.num_frames = 0; li.reached_final = 0;
    li.out_states = 0; li.out_arcs = 0; li.cur_base = 0; li.n_cur = n_t; li.n_order_sensitive = 0; li.hash_size = 1000; li.order_sel = 1;
    __threadfence();
    fresh[L] = 0;
  }
}

// Frame 0 from the decoder's template (DecParams::t0_*), launched between the InitDecoding kernel above and the token-passing kernel: a lane that stands right behind
// InitDecoding (no frame decoded, the template's tokens) and has at least one frame in this call gets its first frame WITHOUT the passes that only depend on the graph --
// candidate walk, state table, the closure's link bookkeeping, the closure sub-graph, union-find, root lists -- which are the same for every utterance when the adaptive beam
// of frame 0 is +inf (see DecParams).  What is left is what the log-likelihoods decide:
//   costs of the emitting links and tokens (the reference's own arithmetic: (cost + (offset - loglike)) + graph, :779-797), the closure's costs as the least fixpoint over the
//   template's epsilon links (:830-897 relaxes the same arcs until nothing changes: the same minimum), the links at their final costs, the REPLAY of the LIFO queue on the
//   template's components (lit_replay_component: the order the closure creates tokens in depends on which cost improves when), the creation ranks and the frame's HashList order.
// The token-passing kernel then resumes the lane at frame 1 (row_skip = 1).  Bit-identical to that kernel's own frame 0 (tests/test_decoder_literal_gpu.py).
__global__ __launch_bounds__(kBlock) void k3_decode_frame0_from_template_kernel(DecParams p) {
  extern __shared__ __attribute__((aligned(16))) char arena[];
  __shared__ Shared sh; __shared__ int s_bad;
  const int tid = threadIdx.x;
  const int L = p.q_lanes ? __builtin_amdgcn_readfirstlane(p.q_lanes[blockIdx.x]) : (int)blockIdx.x;
  const long long r0 = p.row_off[L]; const int T = (int)(p.row_off[L + 1] - r0);
  LaneInfo &li = p.info[L];
  if (tid == 0) p.row_skip[L] = 0;
  if (p.fresh[L] || T < 1 || p.t0_n <= 0 || li.status != kStOk || li.num_frames != 0 || li.cur_base != 0 || li.n_cur != p.tpl_n || li.n_links != p.tpl_nl || li.order_sel != 1) return;
  const LanePool lp = k3_uniform_pool(p.pools[L]);
  const int n = p.t0_n, n_e = p.t0_n_e, nle = p.t0_nle, nlx = p.t0_nlx, ncid = p.t0_ncid, niq = p.t0_niq; const long long nb = p.tpl_n, l0 = p.tpl_nl;
  if (lp.tcap < nb + n || lp.lcap < l0 + nle + nlx) return;      // (the token-passing kernel grows the pools and decodes the frame itself)
  const LitLane q(p, L);
  const float *ll = p.lane_rows ? p.lane_rows[L] : p.loglikes + r0 * p.ld;
  const float kInf = __builtin_inff(), co = -p.t0_best;
  int *tok_state = lp.tok_state; unsigned *tok_cost = lp.tok_cost; Link *links = lp.links; int *link_arc = lp.link_arc;
  if (tid == 0) { sh.err = 0; sh.flag = 0; s_bad = 0; }
  // ---- the frame's tokens; the emitting links: cost of a link = (source cost + (offset - loglike)) + graph cost, cost of a token = the minimum over its links
  for (int i = tid; i < n; i += kBlock) { tok_state[nb + i] = p.t0_state[i]; K3_AST(&tok_cost[nb + i], kEncMax); }
  __syncthreads();
  for (int l = tid; l < nle; l += kBlock) {
    const int4 e = p.t0_elinks[l]; const int s_ = (int)((unsigned)e.x >> 16), d_ = e.x & 0xFFFF;
    const float oc = dec(K3_ALD(&tok_cost[s_])), ac = co - ll[e.z]; const float tot = oc + ac + __int_as_float(e.w);
    k3a_min(&tok_cost[nb + d_], enc(tot));
    store_link(&links[l0 + l], Link{(unsigned)s_, (unsigned)(nb + d_), tot, ac}); store_stream(&link_arc[l0 + l], e.y);
  }
  __syncthreads();
  if (p.t0_pad == 1) return;
  // ---- the replay's starting costs (the tokens right after ProcessEmitting; +inf: not created yet) and the emitting tokens' creation ranks
  unsigned *clist = reinterpret_cast<unsigned *>(q.rflag);
  for (int c = tid; c < ncid; c += kBlock) { const int i = p.t0_c2t[c]; q.rcost[c] = i < n_e ? dec(K3_ALD(&tok_cost[nb + i])) : kInf; }
  for (int i = tid; i < n_e; i += kBlock) K3_AST(&q.label[i], (unsigned)p.t0_rank1[i]);
  // ---- ProcessNonemitting's costs: relax the template's epsilon links until nothing moves (every one of them passes: the cutoff is +inf)
  for (int round = 0; round < 100000; round++) {
    int changed = 0;
    for (int l = tid; l < nlx; l += kBlock) {
      const int4 x = p.t0_xlinks[l]; const int s_ = (int)((unsigned)x.x >> 16), d_ = x.x & 0xFFFF;
      const unsigned cs = K3_ALD(&tok_cost[nb + s_]);
      if (cs != kEncMax) { const unsigned e = enc(dec(cs) + __int_as_float(x.w)); if (e < k3a_min(&tok_cost[nb + d_], e)) changed = 1; }
    }
    if (!__syncthreads_or(changed)) break;
  }
  // ---- the closure's forward links at their sources' final costs (Link::ac of an epsilon link = the source cost it was made at: the live-link stamp)
  for (int l = tid; l < nlx; l += kBlock) {
    const int4 x = p.t0_xlinks[l]; const int s_ = (int)((unsigned)x.x >> 16), d_ = x.x & 0xFFFF;
    const unsigned cs = K3_ALD(&tok_cost[nb + s_]);
    store_link(&links[l0 + nle + l], Link{(unsigned)(nb + s_), (unsigned)(nb + d_), dec(cs) + __int_as_float(x.w), __uint_as_float(cs)}); store_stream(&link_arc[l0 + nle + l], x.y);
  }
  __syncthreads();
  if (p.t0_pad == 2) return;
  // ---- replay of the LIFO queue, one thread per component of the template (lit_replay_components' worker phase and label phase)
  for (int w = tid; w < p.t0_nworkers; w += kBlock) {
    const int4 w0 = p.t0_wrec[2 * w], w1 = p.t0_wrec[2 * w + 1];
    if (!lit_replay_component(q.rcost, p.t0_meta, p.t0_ar, clist, p.t0_rlist, q.rinfo, q.stack + w0.w, w1.x, w0.x, w0.y, w0.z, kInf)) {
      if (p.t0_pad == 9 && atomicAdd(&s_bad, 1) < 3) { const int2 r_ = p.t0_rlist[w0.x]; const int4 m_ = p.t0_meta[r_.y];
        printf("frame0 lane %d worker %d: w0 %d %d %d %d scap %d | first root k %d cid %d meta %d %d %d %d rcost %g\n", L, w, w0.x, w0.y, w0.z, w0.w, w1.x, r_.x, r_.y, m_.x, m_.y, m_.z, m_.w, (double)q.rcost[r_.y]); }
      s_bad = 1;
    }
  }
  __syncthreads();
  if (p.t0_pad == 3) return;
  const int created = block_excl_scan([&](int j) { return (unsigned)q.rinfo[niq - 1 - j].y; }, reinterpret_cast<unsigned *>(q.dense), niq, sh.redi);
  if (s_bad) {      // a component's stack slice overflowed (the token-passing kernel falls back to its one-wavefront replay then): the frame is left to that kernel -- nothing
    // it reads has been changed but the creation labels, which go back to their idle pattern
    for (int i = tid; i < n; i += kBlock) K3_AST(&q.label[i], kLabelNone);
    if (tid == 0 && p.t0_pad == 9) printf("frame0 lane %d: a component stack overflowed, frame left to the token-passing kernel\n", L);
    return;
  }
  if (n_e + created != n) { if (tid == 0) li.status = K3_ERR_HIP; return; }      // (cannot happen: every token of the template is reachable and the cutoff is +inf)
  for (int j = tid; j < niq; j += kBlock) {
    const int2 ri = q.rinfo[niq - 1 - j]; const unsigned base = (unsigned)n_e + (unsigned)q.dense[j];
    for (int t = 0; t < ri.y; t++) K3_AST(&q.label[p.t0_c2t[clist[ri.x + t]]], base + (unsigned)t);
  }
  __syncthreads();
  if (p.t0_pad == 4) return;
  // ---- the frame's HashList order (the next frame's visit order) and creation order
  const unsigned hash_size = p.t0_hash_size; int *ord_nxt = q.order[0]; long long lt_last__ = 0; (void)lt_last__;
  int *const s_tab = reinterpret_cast<int *>(arena); char *const smem_raw = arena + kLitTabBytes + kLitMarkBytes + kLitAuxBytes;
  if (n <= kHoN) lit_hash_order_lds(q, sh, n, (unsigned)n, tok_state + nb, hash_size, ord_nxt, true, true, false, smem_raw, s_tab, lt_last__);
  else if (!lit_hash_order_mid(q, sh, arena, n, (unsigned)n, tok_state + nb, hash_size, ord_nxt, true, true, false) &&
           (p.lit_force_hbm_order || !lit_hash_order_big(q, sh, arena, n, (unsigned)n, tok_state + nb, hash_size, ord_nxt, true, true, false, lt_last__)))
    lit_hash_order(q, sh, n, (unsigned)n, tok_state + nb, hash_size, ord_nxt, true, false, lt_last__);
  for (int i = tid; i < n; i += kBlock) K3_AST(&q.label[i], kLabelNone);
  __syncthreads();
  if (p.t0_pad == 5) return;
  if (tid == 0) {
    long long *tok_off = p.tok_off + L * p.fstride, *loff_e = p.link_off_e + L * p.fstride, *loff_n = p.link_off_n + L * p.fstride;
    tok_off[2] = nb + n; loff_e[0] = l0; loff_n[1] = l0 + nle; loff_e[1] = l0 + nle + nlx;
    (p.st_ntoks + L * p.fstride)[0] = (int)nb; (p.st_cur + L * p.fstride)[0] = kInf; (p.st_ab + L * p.fstride)[0] = kInf; (p.st_next + L * p.fstride)[0] = kInf;
    (p.st_co + L * p.fstride)[0] = co;
    li.n_tokens = nb + n; li.n_links = l0 + nle + nlx; li.n_cands += p.t0_m_e; li.n_eps += p.t0_eps; li.max_frame_tokens = n > (int)nb ? n : (int)nb; li.num_frames = 1;
    li.cur_base = nb; li.n_cur = n; li.hash_size = (int)hash_size; li.order_sel = 0;
    __threadfence();
    p.row_skip[L] = 1;
  }
}

__global__ __launch_bounds__(kBlock, K3_LIT_WPE) void k3_decode_forward_literal_kernel(DecParams p) {
  // One dynamic LDS arena (kLitArena bytes).  The general path below carves its level-1 state table, mark bits, work-lists / clash bins / union-find
  // parents and the replay segment out of its first 77.5 KB; a frame of the LDS-resident path (k3_decoder_fast.h) uses all of it, so the two
  // never run at the same time and the general path re-initialises its table after a fast frame.
  extern __shared__ __attribute__((aligned(16))) char arena[];
  __shared__ Shared sh; __shared__ LitShared ls; __shared__ FastShared fs;
  // A lane is a latency chain that issues in a sixth of its cycles; the workgroups that share its CU in the tail of a batch are the next batch's GEMMs, which
  // would issue every cycle.
  // Raised priority lets the lane's few instructions go first (the GEMM loses nothing it could use: its MFMA work is fixed).
  __builtin_amdgcn_s_setprio(K3_LIT_BASE_PRIO);
  int *const s_tab = reinterpret_cast<int *>(arena);      // level-1 table {key, cost, token}; between the closure and the end of a frame: the replay's records
  int *const s_lkey = s_tab; unsigned *const s_lcost = reinterpret_cast<unsigned *>(s_tab + kHL); int *const s_ltok = s_tab + 2 * kHL;
  unsigned *const s_lmark = reinterpret_cast<unsigned *>(arena + kLitTabBytes);
  // 8 KB shared by three users with disjoint lifetimes: the eps rounds' work-lists (first half), the serial replay's clash bins (second half:
  // which lane of a batch targets a token; a false clash only takes the one-by-one path), the component replay's union-find parents (all of it)
  int *const s_aux = reinterpret_cast<int *>(arena + kLitTabBytes + kLitMarkBytes);
  char *const smem_raw = arena + kLitTabBytes + kLitMarkBytes + kLitAuxBytes;       // replay arrays of a small frame
  static_assert(2 * kWlLds * sizeof(unsigned short) <= 1024 * sizeof(int) && kHL / 2 <= 2 * 1024, "s_aux layout");
  unsigned short (*const s_lwl)[kWlLds] = reinterpret_cast<unsigned short (*)[kWlLds]>(s_aux); int *const s_own = s_aux + 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6; constexpr int nw = kBlock / 64;
  __shared__ LanePool s_pool; __shared__ int s_qi;
  // Which lane this workgroup decodes: entry blockIdx.x of the call's lane list -- the lanes sorted by their number of frames, longest first, so that the two workgroups of a
  // CU (b and b + 256 of a 512-lane launch) are a long and a short utterance and the long one has the CU to itself once the short one is done (a ragged batch in
  // generation order: 144 ms per launch, sorted: 83 ms) -- or, builds with -DK3_LIT_QUEUE=1 and fewer workgroups than lanes (k3_decoder_config.resident_lanes), entry after
  // entry of that list through an atomic cursor until it is exhausted.  (The loop costs the kernel 165 spilled vector registers, +14 % time: measured, not shipped.)
#if K3_LIT_QUEUE
#define K3_LANE_DONE continue
  for (int q_it = 0;; q_it++) {
  int L;
  if (p.q_head) {
    __syncthreads();      // (the previous lane's last readers of sh / s_qi are done)
    if (tid == 0) s_qi = __hip_atomic_fetch_add(p.q_head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int k_ = s_qi;
    if (k_ >= (int)p.q_n) break;
    L = __builtin_amdgcn_readfirstlane(p.q_lanes[k_]);
  } else { if (q_it) break; L = p.q_lanes ? __builtin_amdgcn_readfirstlane(p.q_lanes[blockIdx.x]) : (int)blockIdx.x; }
#else
#define K3_LANE_DONE return
  {
  const int L = p.q_lanes ? __builtin_amdgcn_readfirstlane(p.q_lanes[blockIdx.x]) : (int)blockIdx.x; (void)s_qi;
#endif
  // (row_skip: the frames of this call that k3_decode_frame0_from_template_kernel has already decoded for the lane -- 0 or 1)
  const int skip = p.row_skip ? __builtin_amdgcn_readfirstlane(p.row_skip[L]) : 0;
  const long long r0 = p.row_off[L] + skip; const int T = (int)(p.row_off[L + 1] - r0);
  LanePool lp = k3_uniform_pool(p.pools[L]);      // this lane's token / link pools (grow_lane_pools moves them when the lane outgrows its reservation)
  int *tok_state = lp.tok_state; unsigned *tok_cost = lp.tok_cost; Link *links = lp.links; int *link_arc = lp.link_arc;
  Slot *hash = p.hash + (long long)L * (p.hash_mask + 1);
  int *tok_slot = p.tok_slot + (long long)L * p.frame_tokens_cap, *wl = p.wl + 2ll * L * p.frame_tokens_cap;
  long long *tok_off = p.tok_off + L * p.fstride, *loff_e = p.link_off_e + L * p.fstride, *loff_n = p.link_off_n + L * p.fstride;
  int *st_ntoks = p.st_ntoks + L * p.fstride;
  float *st_cur = p.st_cur + L * p.fstride, *st_ab = p.st_ab + L * p.fstride, *st_next = p.st_next + L * p.fstride, *st_co = p.st_co + L * p.fstride;
  const unsigned mask = (unsigned)p.hash_mask; const float kInf = __builtin_inff(); const int cap = p.frame_tokens_cap;
  const LitLane q(p, L);
  LaneCtx lc{tok_state, tok_cost, links, link_arc, tok_off, loff_e, loff_n, st_ntoks, st_cur, st_ab, st_next, st_co, lp.tcap, lp.lcap};
  // room behind what the lane holds for `need_l` more links and a whole frame of tokens; all threads call it where nothing of the frame being built is in the pools yet
  auto make_room = [&](long long n_tok, long long need_l) -> bool {
    const long long nl_ = sh.n_link;
    if (lp.tcap - n_tok >= p.frame_tokens_cap && lp.lcap - nl_ >= need_l) return true;
    if (!grow_lane_pools(p, L, lp, n_tok, nl_, p.frame_tokens_cap, need_l, &s_pool)) { if (tid == 0) sh.err = K3_ERR_OVERFLOW; __syncthreads(); return false; }
    tok_state = lp.tok_state; tok_cost = lp.tok_cost; links = lp.links; link_arc = lp.link_arc;
    lc.tok_state = tok_state; lc.tok_cost = tok_cost; lc.links = links; lc.link_arc = link_arc; lc.tcap = lp.tcap; lc.lcap = lp.lcap;
    return true;
  };
  long long cyc_fast = 0, cyc_general = 0, cyc_t0 = 0;
  // (thread 0: frames by path, reasons of the fast path's give-ups; k3_decoder_phase_cycles reads them)
  int n_fast = 0, n_gaveup = 0, n_general = 0;
  int why[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool v_valid = false, lds_dirty = false;      // the visit-order arrays of the fast path hold the current frame / the arena was used by a fast frame
  if (tid < 16) sh.prof[tid] = 0;
  if (tid < 12) fs.prof[tid] = 0;
  if (tid == 0) { sh.n_next = 0; sh.n_cand = 0; sh.err = 0; sh.n_link = 0; sh.min_tot = kEncMax; sh.flag = 0; sh.n_eps = 0; sh.n_emit = 0; sh.n_os = 0; }
  for (int i = tid; i < kHL; i += kBlock) { s_lkey[i] = kEmpty; s_lcost[i] = kEncMax; s_ltok[i] = -1; }
  for (int i = tid; i < 3 * (kHL / 32); i += kBlock) s_lmark[i] = 0;
  const Table tb{s_lkey, s_lcost, s_ltok, s_lmark, hash, mask};
  __syncthreads();
  long long t_last__ = 0; (void)t_last__;
  long long lt_last__ = (long long)__builtin_readcyclecounter(); (void)lt_last__;
  unsigned cnt_eps = 0, cnt_emit = 0, cnt_os = 0;
  long long cur_base = 0; int n_cur = 0, max_frame = 0, f0 = 0, status = kStOk, sel = 0; unsigned hash_size = 1000;      // :41 toks_.SetSize(1000)
  unsigned creg[kCurRegs]; int sreg[kCurRegs];
#pragma unroll
  for (int k = 0; k < kCurRegs; k++) { creg[k] = kEncMax; sreg[k] = 0; }
  const bool fresh = p.fresh[L] != 0;
  if (!fresh && T == 0) K3_LANE_DONE;
  if (fresh) {
    if (p.info[L].status < 0) {      // a failed utterance left the scratch in an unknown state: back to the idle patterns
      for (unsigned i = tid; i <= mask; i += kBlock) { Slot *s_ = &hash[i]; K3_AST(&s_->cost, kEncMax); K3_AST(&s_->stamp, 0); K3_AST(&s_->tok, -1); K3_AST(&s_->key, kEmpty); }
      for (int i = tid; i < cap; i += kBlock) K3_AST(&q.label[i], kLabelNone);
      for (int i = tid; i < p.seq_words_cap; i += kBlock) K3_AST(&q.bm[i], 0u);
      __threadfence(); __syncthreads();
    }
    if (tid == 0) {
      bool cl; const int slot = tb.claim(p.start, &cl);
      tb.cost_min(slot, enc(0.0f)); tb.set_tok(slot, 0); tok_slot[0] = slot; tok_state[0] = p.start; K3_AST(&tok_cost[0], kEncMax); sh.n_next = 1;
      sh.n_wl[0] = 0; sh.n_wl[2] = 0; sh.n_wl[1] = 1; for (int i = 0; i < 4; i++) sh.err_r[i] = 0;
      if (slot < kHL) s_lwl[1][0] = (unsigned short)slot; else { s_lwl[1][0] = 0xFFFF; wl[p.frame_tokens_cap] = slot; }
      tok_off[0] = 0; loff_n[0] = 0;
      K3_AST(&q.label[0], 0u); q.c0[0] = 0.0f;
    }
    __syncthreads();
  } else {
    const LaneInfo &li = p.info[L];
    if (li.status != kStOk) K3_LANE_DONE;
    f0 = li.num_frames; cur_base = li.cur_base; n_cur = li.n_cur; max_frame = li.max_frame_tokens; sel = li.order_sel; hash_size = (unsigned)li.hash_size;
    if (tid == 0) {
      sh.n_link = li.n_links;
      sh.n_eps = (unsigned long long)li.n_eps;
      sh.n_emit = (unsigned long long)li.n_cands;
      sh.n_os = (unsigned long long)li.n_order_sensitive;
    }
    __syncthreads();
  }

  for (int f = fresh ? -1 : f0; f < f0 + T; f++) {
    if (block_err(sh)) break;
    float accept = p.beam; long long nb = 0, eps_l0 = sh.n_link; unsigned m_e = 1; int n_e = 1;      // (f == -1: InitDecoding, no links yet)
#ifdef K3_LIT_PROF
    if (tid == 0) sh.prof_n = n_cur;
#endif
    const int *ord_cur = q.order[sel]; int *ord_nxt = q.order[sel ^ 1];
    if (f >= 0) {
      const float *ll = p.lane_rows ? p.lane_rows[L] + (long long)(f - f0 + skip) * p.ld : p.loglikes + (r0 + (f - f0)) * p.ld;
      if (n_cur == 0) { status = kStNoTokens; break; }
      // the pools hold a whole frame of tokens and the links of an LDS-resident frame (<= kFM emitting + kFE epsilon) behind what the lane has; the general path
      // asks again once it knows how many emitting arcs the frame examines (after pass A)
      if (!make_room(cur_base + n_cur, (long long)kFM + kFE)) break;
      const int *cst = tok_state + cur_base; const unsigned *ccs = tok_cost + cur_base;
      // ---- the LDS-resident frame (k3_decoder_fast.h) when the frame fits; a frame it gives up on is redone below, from the same inputs
      if (tid == 0) cyc_t0 = (long long)__builtin_readcyclecounter();
      if (p.fast_cap > 0 && n_cur <= p.fast_cap) {
        lds_dirty = true;
        if (v_valid || fast_import(p, arena, lc, ord_cur, cur_base, n_cur, fs)) {
          const long long link0 = sh.n_link;
          const int n_new = lit_frame_fast(p, sh, fs, arena, q, lc, f, ll, cur_base, n_cur, hash_size, ord_nxt, p.fast_cap, cnt_emit, cnt_os, cnt_eps);
          if (n_new >= 0) {
            cur_base += n_cur; n_cur = n_new; max_frame = n_cur > max_frame ? n_cur : max_frame; sel ^= 1; v_valid = true; n_fast++;
            if (tid == 0) cyc_fast += (long long)__builtin_readcyclecounter() - cyc_t0;
#ifdef K3_LIT_FRAMECYC      // profiling builds only (tools/prof_frames.py): FrameStats' adaptive_beam column = cycles of the frame, > 0: LDS-resident path
            if (tid == 0) st_ab[f] = (float)((long long)__builtin_readcyclecounter() - cyc_t0);
#endif
            continue;
          }
          __syncthreads();
          if (tid == 0) { sh.n_link = link0; sh.n_next = 0; n_gaveup++; const int r_ = fs.reason; if (r_ >= 1 && r_ <= 12) why[r_ - 1]++; }
#ifdef K3_LIT_DEBUG
          if (tid == 0) printf("lane %d frame %d: fast path gave up (reason %d), n_cur %d\n", L, f, fs.reason, n_cur);
#endif
          __syncthreads();
          if (block_err(sh)) break;
        }
      }
      v_valid = false; n_general++;
      if (lds_dirty) {      // the arena held a fast frame: the general path starts from an empty level-1 table
        for (int i = tid; i < kHL; i += kBlock) { s_lkey[i] = kEmpty; s_lcost[i] = kEncMax; s_ltok[i] = -1; }
        lds_dirty = false;
        __syncthreads();
      }
      // ---- GetCutoff (:653-720): the best token is the FIRST minimum-cost token of the list (strict <, :661-663)
      unsigned long long bm = ~0ull;
      for (int r = tid; r < n_cur; r += kBlock) { const unsigned long long v = ((unsigned long long)ccs[ord_cur[r]] << 32) | (unsigned)r; bm = v < bm ? v : bm; }
      bm = block_min_u64(bm, sh);
      const float best = dec((unsigned)(bm >> 32)); const int best_state = cst[ord_cur[(int)(unsigned)(bm & 0xFFFFFFFFull)]];
      auto for_keys = [&](auto fn) { for (int i0 = 0; i0 < n_cur; i0 += kBlock) { const int i = i0 + tid; fn(i < n_cur, i < n_cur ? ccs[i] : 0u); } };
      float cur_cutoff, ab;
      const float beam_cutoff = best + p.beam;
      if (p.max_active == 0x7FFFFFFF && p.min_active == 0) { ab = p.beam; cur_cutoff = beam_cutoff; }
      else {
        const unsigned ebc = enc(beam_cutoff);
        int c_lt = 0, c_le = 0;
        for (int i = tid; i < n_cur; i += kBlock) { const unsigned k = ccs[i]; c_lt += k < ebc; c_le += k <= ebc; }
        c_lt = block_sum_i32(c_lt, sh); c_le = block_sum_i32(c_le, sh);
        int kth = -1;
        if (n_cur > p.max_active && c_lt > p.max_active) kth = p.max_active;
        else if (n_cur > p.min_active && p.min_active == 0) { ab = p.beam; cur_cutoff = beam_cutoff; }
        else if (n_cur > p.min_active && c_le > p.min_active) { ab = p.beam; cur_cutoff = beam_cutoff; }
        else if (n_cur > p.min_active) kth = p.min_active;
        else { ab = kInf - best + p.beam_delta; cur_cutoff = kInf; }
        if (kth >= 0) { const float sel_ = dec(block_select_kth(for_keys, kth, sh)); ab = sel_ - best + p.beam_delta; cur_cutoff = sel_; }
      }
      K3_LT(0); K3_LQ(0);
      { const unsigned want = (unsigned)((float)n_cur * p.hash_ratio); if (want > hash_size) hash_size = want; }      // PossiblyResizeHash (:227-233)
      if (hash_size > (unsigned)p.hash_cap) { if (tid == 0) sh.err = K3_ERR_OVERFLOW; }
      const float co = -best;
      // ---- pre-pass over the best token's emitting arcs (:753-768)
      __syncthreads();
      unsigned n0 = kEncMax;
      {
        const int2 a = p.offs[best_state];
        for (int arc = a.x + tid; arc < a.y; arc += kBlock) {
          const ArcRec r = p.arcs[arc];
          const float nw_ = r.w + co - ll[r.pdf] + best;
          const unsigned e = enc(nw_ + ab);
          n0 = e < n0 ? e : n0;
        }
      }
      n0 = (unsigned)(block_min_u64((unsigned long long)n0, sh) & 0xFFFFFFFFull);
      const float next0 = n0 == kEncMax ? kInf : dec(n0);
      if (tid == 0) { sh.n_cand = 0; sh.n_next = 0; }
      if (tid < 3) sh.n_wl[tid] = 0;
      if (tid < 4) sh.err_r[tid] = 0;
      for (int i = tid; i < 3 * (kHL / 32); i += kBlock) s_lmark[i] = 0;
      __syncthreads();
      if (block_err(sh)) break;
      K3_LT(1); K3_LQ(1);
      // ---- pass A: per 64-token chunk of the visit order, the number of emitting arcs and min (tot + adaptive_beam)
      // (chunks of 2^csh tokens: 64 for the large frames, fewer when the frame would give the wavefronts less than two chunks each -- the first frame of an utterance is
      // a hundred tokens with tens of thousands of arcs, two 64-token chunks walked by two wavefronts --, within the chunk arrays' cap / 64 + 2 entries)
      int csh = 6;
      while (K3_LIT_CSH && csh > 3 && ((n_cur + (1 << csh) - 1) >> csh) < 2 * nw && ((n_cur + (1 << (csh - 1)) - 1) >> (csh - 1)) <= cap / 64) csh--;
      const int nchunks = (n_cur + (1 << csh) - 1) >> csh;
      auto chunk_tokens = [&](int c, int &i, float &cost, int &beg, int &deg) {
        const int r = (c << csh) + lane; const bool v = lane < (1 << csh) && r < n_cur;
        i = v ? ord_cur[r] : 0; const unsigned cb = v ? ccs[i] : kEncMax; const int st = v ? cst[i] : 0; cost = dec(cb); beg = 0; deg = 0;
        if (v && cost <= cur_cutoff) { const int2 a = p.offs[st]; beg = a.x; deg = a.y - a.x; }
      };
      for (int c = wave; c < nchunks; c += nw) {
        int i, beg, deg; float cost; chunk_tokens(c, i, cost, beg, deg);
        // pass B reads this instead of walking order -> token -> state -> offsets again
        if (lane < (1 << csh) && (c << csh) + lane < n_cur) q.vis[(c << csh) + lane] = make_int4(i, __float_as_int(cost), beg, deg);
        unsigned cm = kEncMax;
#if K3_LIT_PF_SEQ
        const int total = wave_expand_seq_pf(p.arcs, ll, beg, deg, [&](bool valid, int, int, int owner, const ArcRec &r, float llv) {
          const float oc = __shfl(cost, owner);
          if (valid) { const float ac = co - llv; const float tot = oc + ac + r.w; const unsigned e = enc(tot + ab); cm = e < cm ? e : cm; }
          cnt_emit += valid;
        });
#else
        const int total = wave_expand_seq(p.arcs, beg, deg, [&](bool valid, int, int, int owner, const ArcRec &r) {
          const float oc = __shfl(cost, owner);
          if (valid) { const float ac = co - ll[r.pdf]; const float tot = oc + ac + r.w; const unsigned e = enc(tot + ab); cm = e < cm ? e : cm; }
          cnt_emit += valid;
        });
#endif
        cm = wave_min_u32(cm);
        if (lane == 0) { q.cmin[c] = cm; q.ccnt[c] = total; }
      }
      __syncthreads();
      K3_LT(2); K3_LQ(2);
      // exclusive scans over the chunks: bound in force at a chunk's first arc, sequence number of its first arc
      if (wave == 0) {
        unsigned run = enc(next0); int base = 0;
        unsigned *cpre = q.cmin + (cap / 64 + 2); int *cbase = q.ccnt + (cap / 64 + 2);
        for (int c0 = 0; c0 < nchunks; c0 += 64) {
          const int c = c0 + lane; const unsigned m = c < nchunks ? q.cmin[c] : kEncMax; const int k = c < nchunks ? q.ccnt[c] : 0;
          unsigned em = m; int ik = k;
          em = wave_incl_min_u32(em); ik = wave_incl_sum_i32(ik);
          unsigned exm = wave_shr1_u32(em, kEncMax); exm = run < exm ? run : exm;
          if (c < nchunks) { cpre[c] = exm; cbase[c] = base + ik - k; }
          const unsigned wm = (unsigned)__builtin_amdgcn_readlane((int)em, 63); run = wm < run ? wm : run; base += __builtin_amdgcn_readlane(ik, 63);
        }
        if (lane == 0) { ls.final_cut = run; ls.m_e = base; }
      }
      if (tid == 0) loff_e[f] = sh.n_link;
      __syncthreads();
      accept = dec(ls.final_cut); m_e = (unsigned)ls.m_e;      // the frame's final next_cutoff; labels 0 .. m_e-1 belong to the emitting arcs
      if ((long long)m_e + cap > 32ll * p.seq_words_cap) { if (tid == 0) sh.err = K3_ERR_OVERFLOW; }
      if (block_err(sh)) break;
      nb = cur_base + n_cur;
      // every emitting arc examined makes at most one forward link, the closure at most eps_cap more (its own check): make room before anything of the frame is written
      if (!make_room(nb, (long long)m_e + p.eps_cap)) break;
      cst = tok_state + cur_base; ccs = tok_cost + cur_base;
      K3_LT(2); K3_LQ(3);
      // ---- pass B: accept against the bound in force at each arc; min cost / min sequence number per destination state; forward links
      {
        const unsigned *cpre = q.cmin + (cap / 64 + 2); const int *cbase = q.ccnt + (cap / 64 + 2);
        for (int c = wave; c < nchunks; c += nw) {
          int i = 0, beg = 0, deg = 0; float cost = 0.0f;
          if (lane < (1 << csh) && (c << csh) + lane < n_cur) { const int4 v = q.vis[(c << csh) + lane]; i = v.x; cost = __int_as_float(v.y); beg = v.z; deg = v.w; }
          unsigned run = cpre[c]; const int jbase = cbase[c];
#if K3_LIT_PF_SEQ
          wave_expand_seq_pf(p.arcs, ll, beg, deg, [&](bool valid, int j, int arc, int owner, const ArcRec &r, float llv) {
#else
          wave_expand_seq(p.arcs, beg, deg, [&](bool valid, int j, int arc, int owner, const ArcRec &r) {
            const float llv = valid ? ll[r.pdf] : 0.0f;
#endif
            const float oc = __shfl(cost, owner); const int oi = __shfl(i, owner);
            float ac = 0.0f, tot = 0.0f; unsigned e = kEncMax;
            if (valid) { ac = co - llv; tot = oc + ac + r.w; e = enc(tot + ab); }
            unsigned em = e;
            em = wave_incl_min_u32(em);
            unsigned exm = wave_shr1_u32(em, kEncMax); exm = run < exm ? run : exm;
            { const unsigned wm = (unsigned)__builtin_amdgcn_readlane((int)em, 63); run = wm < run ? wm : run; }
            const bool acc = valid && tot < dec(exm);
            cnt_os += acc && !(tot < accept);
            const int state = (int)((unsigned)r.next & ~kEpsFlag);
            bool claimed = false, mk = false; int slot = -1;
            if (acc) {
              slot = tb.claim(state, &claimed);
              if (slot < 0) { sh.err = K3_ERR_OVERFLOW; claimed = false; } else { tb.cost_min(slot, enc(tot)); mk = true; }
            }
            int idx = wave_append(claimed, &sh.n_next);
            if (claimed) {
              if (idx < p.frame_tokens_cap && nb + idx < lp.tcap) { tok_slot[idx] = slot; tok_state[nb + idx] = state; K3_AST(&tok_cost[nb + idx], kEncMax); }
              else { sh.err = K3_ERR_OVERFLOW; idx = 0; }
              tb.set_tok(slot, idx);
            }
            {
              const bool q1 = claimed && r.next < 0;
              const int pos1 = wave_append(q1, &sh.n_wl[1]);
              if (q1) {
                if (pos1 < kWlLds) s_lwl[1][pos1] = slot < kHL ? (unsigned short)slot : (unsigned short)0xFFFF;
                if (pos1 >= kWlLds || slot >= kHL) { if (pos1 < p.frame_tokens_cap) wl[p.frame_tokens_cap + pos1] = slot; else sh.err = K3_ERR_OVERFLOW; }
              }
            }
            if (mk && !claimed) { idx = tb.wait_tok(slot, &sh.err); if (idx < 0) idx = 0; }
            if (mk) k3a_min(&q.label[idx], (unsigned)(jbase + j));
            const long long pos = wave_append64(mk, &sh.n_link);
            if (mk) {
              if (pos < lp.lcap) { store_link(&links[pos], Link{(unsigned)(cur_base + oi), (unsigned)(nb + idx), tot, ac}); store_stream(&link_arc[pos], arc); }
              else sh.err = K3_ERR_OVERFLOW;
            }
          });
        }
      }
      if (block_err(sh)) break;
      K3_LT(4); K3_LQ(4);
      n_e = sh.n_next; eps_l0 = sh.n_link;      // the frame's eps links start here
      if (tid == 0) { loff_n[f + 1] = sh.n_link; st_ntoks[f] = n_cur; st_cur[f] = cur_cutoff; st_ab[f] = ab; st_next[f] = accept; st_co[f] = co; }
      for (int i = tid; i < n_e; i += kBlock) q.c0[i] = dec(tb.cost(tok_slot[i]));      // costs right after ProcessEmitting (the replay starts from them)
      __syncthreads();
    }   // f >= 0
    K3_LT(4); K3_LQ(5);
    // ---- ProcessNonemitting: order-free fixpoint (costs, new tokens, eps links)
    finish_frame<false>(p, sh, tb, accept, nb, tok_state, tok_cost, links, link_arc, lp.tcap, lp.lcap, tok_slot, wl, s_lwl, creg, sreg, t_last__, cnt_eps);
    if (block_err(sh)) break;
    const int n = sh.n_next;
    K3_LT(7); K3_LQ(6);
    // ---- closure sub-graph in "closure id" space.  Only tokens that take part in it get an id: sources (>= 1 eps arc that passes at the
    // token's FINAL cost -- an arc that fails there fails at every earlier, higher cost too) and their destinations.  Per id: the cost the
    // replay has seen so far (+inf = not created yet), meta = {first passing arc, number of passing arcs}; per passing arc, in FST order:
    // {destination id, weight}.
    if (tid == 0) { ls.n_created = 0; ls.n_csr = 0; }
    // step 1: final costs into the pool (the closure's "expanded at" marker is no longer needed).  The sub-graph is read off the eps links the
    // fixpoint wrote: a link carries the source cost it was made at, a token is expanded once at every cost it takes, so the links made at a
    // token's FINAL cost are exactly its arcs that pass there -- no second walk over offsets, arcs and the state table.
    // Frames of <= kCsrN tokens count in LDS (the level-1 table is dead once the costs are out; read-modify-write operations at the L2 are what the large frames of all lanes
    // queue for) and stream the counts out before the hash-order pass takes the arena.
    constexpr int kCsrN = 16384; static_assert((size_t)kCsrN * 4 + kCsrN / 8 <= kLitGeneralLds, "per-token counters of the closure sub-graph fit the arena");
    const bool csr_lds = n <= kCsrN;
    int *l_cnt = reinterpret_cast<int *>(arena); unsigned *l_flag = reinterpret_cast<unsigned *>(arena + (size_t)kCsrN * 4);
    for (int i = tid; i < n; i += kBlock) { tok_cost[nb + i] = tb.cost(tok_slot[i]); if (!csr_lds) { K3_AST(&q.rflag[i], 0); K3_AST(&q.rown[i], 0); } }
    __syncthreads();
    if (csr_lds) { for (int i = tid; i < n; i += kBlock) l_cnt[i] = 0; for (int i = tid; i < (n + 31) / 32; i += kBlock) l_flag[i] = 0u; __syncthreads(); }
    const long long eps_l1 = sh.n_link;
    for (long long l = eps_l0 + tid; l < eps_l1; l += kBlock) {
      const Link k = links[l];
      if (__float_as_uint(k.ac) == K3_ALD(&tok_cost[k.src])) {      // passing arcs per source; destinations flagged
        const int s_ = (int)(k.src - nb), d_ = (int)(k.dst - nb);
        if (csr_lds) { k3a_add(&l_cnt[s_], 1); k3a_or(&l_flag[d_ >> 5], 1u << (d_ & 31)); } else { k3a_add(&q.rown[s_], 1); k3a_or(&q.rflag[d_], 1); }
      }
    }
    __syncthreads();
    if (csr_lds) { for (int i = tid; i < n; i += kBlock) { q.rown[i] = l_cnt[i]; q.rflag[i] = (int)(l_flag[i >> 5] >> (i & 31) & 1u); } }
    if (block_err(sh)) break;
    for (int i = tid; i < n; i += kBlock) { const int slot = tok_slot[i]; if (slot >= kHL) tb.clear(slot); }      // last use of the table in this frame
    __syncthreads();
    K3_LT(8); K3_LQ(7);
    // ---- the list ProcessNonemitting fills its queue from (:845-850): HashList order of the tokens ProcessEmitting made (their labels and states are
    // untouched by the closure).  Computed here, where the level-1 table's LDS is free for the bucket table of the pass.
    // (it leaves the emitting tokens' DENSE creation ranks 0 .. n_e-1 in q.label; the closure numbers its tokens n_e .. n-1 behind them, so the frame's second pass has
    // every rank at hand)
    if (n_e <= kHoN && m_e <= (unsigned)kHoM) lit_hash_order_lds(q, sh, n_e, m_e, tok_state + nb, hash_size, ord_nxt, false, false, true, smem_raw, s_tab, lt_last__);
    else if (!lit_hash_order_mid(q, sh, arena, n_e, m_e, tok_state + nb, hash_size, ord_nxt, false, false, true) &&
        (p.lit_force_hbm_order || !lit_hash_order_big(q, sh, arena, n_e, m_e, tok_state + nb, hash_size, ord_nxt, false, false, true, lt_last__)))
      lit_hash_order(q, sh, n_e, m_e, tok_state + nb, hash_size, ord_nxt, false, true, lt_last__);
    K3_LT(6); K3_LQ(8);
    // closure ids of the involved tokens and first arc slots of the sources: one pass
    const int4 tot2 = block_excl_scan4([&](int i) { const int pc = K3_ALD(&q.rown[i]); return make_int4((pc > 0 || K3_ALD(&q.rflag[i]) != 0) ? 1 : 0, pc, 0, 0); },
                                       // (LDS: the source's fill cursor starts at its first slot)
                                       [&](int i, int4 ex) {
                                         q.grp[i] = ex.x;
                                         q.lead[i] = (unsigned)ex.y;
                                         if (csr_lds) l_cnt[i] = ex.y;
                                         else K3_AST(&q.rtmp[i], 0);
                                       },
                                       n, reinterpret_cast<int4 *>(sh.hist));
    const int n_cid = tot2.x, n_arc = tot2.y;
    if (n_arc > p.eps_cap) { if (tid == 0) sh.err = K3_ERR_OVERFLOW; }
    if (block_err(sh)) break;
    // the live links again: (arc, destination token) into the source's slots (any order; step 2 sorts a source's few entries into FST order)
    int *ent_arc = q.cdst, *ent_dst = reinterpret_cast<int *>(q.cw);
    for (long long l = eps_l0 + tid; l < eps_l1; l += kBlock) {
      const Link k = links[l];
      if (__float_as_uint(k.ac) == K3_ALD(&tok_cost[k.src])) {
        const int sl = (int)(k.src - nb);
        const int pos = csr_lds ? k3a_add(&l_cnt[sl], 1) : (int)q.lead[sl] + k3a_add(&q.rtmp[sl], 1);
        ent_arc[pos] = link_arc[l];
        ent_dst[pos] = (int)(k.dst - nb);
      }
    }
    __syncthreads();
    K3_LQ(9);
    // mode 0: costs, meta and arcs in LDS; mode 1: costs in LDS, meta / arcs (read-only during the replay) in HBM; mode 2: all in HBM
    // (mode 0: kRN x (16 B meta + 4 B cost + 4 B creation list) = the table's 12 B x kHL; mode 1: 3 kHL costs in the table, the list where mode 0 keeps its arcs)
#if K3_LIT_CAPTURE      // (the capture build keeps every record of the replay in the lane's HBM scratch, where the host finds it after the launch)
    const int rmode = 2;
#else
    const int rmode = (n_cid <= kRN && n_arc <= kRA) ? 0 : ((n_cid <= 3 * kHL && n - n_e <= kRA * 2) ? 1 : 2);
#endif
    // LDS arena of the replay: the level-1 table's memory (dead from here until the end of the frame) + the dynamic segment
    float *rcost = rmode == 2 ? q.rcost : reinterpret_cast<float *>(s_tab) + (rmode == 0 ? 2 * kHL : 0);
    int4 *meta = rmode == 0 ? reinterpret_cast<int4 *>(s_tab) : q.meta;
    int2 *AR = rmode == 0 ? reinterpret_cast<int2 *>(smem_raw) : q.arcs2;
    int *par = rmode == 0 ? s_aux : q.par;
    const unsigned *clist = rmode == 0 ? reinterpret_cast<unsigned *>(s_tab) + 2 * kHL + kRN : rmode == 1 ? reinterpret_cast<unsigned *>(smem_raw) :
        reinterpret_cast<unsigned *>(q.rflag);
    if (tid == 0) ls.use_lds = 0;      // (the component replay's "take the serial replay" flag)
    // step 2: per involved token its record (and its component-replay cells: own parent, zero counters), per source its passing arcs compacted in
    // FST order.  kSB tokens per thread in flight (the kernel has 128 registers per thread: more would spill).
    constexpr int kSB = 2;
    for (int ib = tid; ib < n; ib += kSB * kBlock) {
      int pc4[kSB], fl4[kSB], cid4[kSB], abeg4[kSB], a1[kSB], d1[kSB], g1[kSB]; float c04[kSB], w1[kSB]; bool inv[kSB];
#pragma unroll
      for (int u = 0; u < kSB; u++) {
        const int i = ib + u * kBlock;
        pc4[u] = 0;
        fl4[u] = 0;
        if (i < n) {
          pc4[u] = K3_ALD(&q.rown[i]);
          fl4[u] = K3_ALD(&q.rflag[i]);
        }
        inv[u] = pc4[u] > 0 || fl4[u] != 0;
      }
#pragma unroll
      for (int u = 0; u < kSB; u++) { const int i = ib + u * kBlock; if (inv[u]) { cid4[u] = q.grp[i]; abeg4[u] = (int)q.lead[i]; c04[u] = i < n_e ? q.c0[i] : kInf; } }
#pragma unroll
      // most sources have exactly one passing arc
      for (int u = 0; u < kSB; u++) {
        a1[u] = -1;
        if (inv[u] && pc4[u] == 1) {
          a1[u] = ent_arc[abeg4[u]];
          d1[u] = ent_dst[abeg4[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < kSB; u++) if (a1[u] >= 0) { g1[u] = q.grp[d1[u]]; w1[u] = p.arcs[a1[u]].w; }
#pragma unroll
      for (int u = 0; u < kSB; u++) {
        const int i = ib + u * kBlock;
        if (inv[u]) {
          const int cid = cid4[u], abeg = abeg4[u], pc = pc4[u];
          q.c2t[cid] = i; rcost[cid] = c04[u]; int d0 = 0, w0 = 0;
          if (p.literal & 1) {
            K3_AST(&par[cid], cid);
            int *ci = reinterpret_cast<int *>(&q.cinfo[cid]);
            K3_AST(&ci[0], 0);
            K3_AST(&ci[1], 0);
            K3_AST(&ci[2], 0);
            K3_AST(&ci[3], 0);
          }
          if (pc == 1) { d0 = g1[u]; w0 = __float_as_int(w1[u]); AR[abeg] = make_int2(d0, w0); }
          else if (pc > 1) {
            for (int a = abeg + 1; a < abeg + pc; a++) {      // FST order = ascending arc index (a source's arcs are contiguous in the graph)
              const int ka = ent_arc[a], kd = ent_dst[a]; int b_ = a - 1;
              while (b_ >= abeg && ent_arc[b_] > ka) { ent_arc[b_ + 1] = ent_arc[b_]; ent_dst[b_ + 1] = ent_dst[b_]; b_--; }
              ent_arc[b_ + 1] = ka; ent_dst[b_ + 1] = kd;
            }
            for (int k = abeg; k < abeg + pc; k++) {
              const int2 ar = make_int2(q.grp[ent_dst[k]], __float_as_int(p.arcs[ent_arc[k]].w));
              if (k == abeg) {
                d0 = ar.x;
                w0 = ar.y;
              }
              AR[k] = ar;
            }
          }
          meta[cid] = make_int4(abeg, pc, d0, w0);      // the first passing arc rides along
        }
      }
    }
    // the initial queue (:845-850): the tokens ProcessEmitting made, in HashList order, that can expand (a token without a passing arc at its
    // final cost pops as a no-op); it is consumed from its back
    K3_LQ(10);
    const int n_iq = block_excl_scan([&](int r) { return K3_ALD(&q.rown[ord_nxt[r]]) > 0 ? 1u : 0u; }, reinterpret_cast<unsigned *>(q.dense), n_e, sh.redi);
    for (int r = tid; r < n_e; r += kBlock) { const int i = ord_nxt[r]; if (K3_ALD(&q.rown[i]) > 0) q.iq[q.dense[r]] = q.grp[i]; }
    __syncthreads();
    K3_LT(8); K3_LQ(11);
#ifdef K3_LIT_PROF
    if (tid == 0 && K3_LP_ON) { if (K3_LIT_PROF == 1) { sh.prof[13] += n; sh.prof[14] += rmode == 0 ? 1 : 0; } sh.prof[15] += 1; }
    const long long rp_t0 = (long long)__builtin_readcyclecounter(); (void)rp_t0;
#endif
    // ---- replay of the LIFO queue (:851-896): only the ORDER in which the closure creates tokens comes out of it.  literal_order = 1: split by
    // connected components of the closure sub-graph, one thread per component; literal_order = 2 (and frames whose component stacks overflow):
    // the whole queue by one wavefront.
    // (separate instantiations so that mode 0 compiles to LDS instructions only: a flat access would wait for every outstanding global access)
    int created_total = -1;
    if (p.literal & 1) {
      if (rmode == 0) created_total = lit_replay_components(p, q, sh, &ls.use_lds, reinterpret_cast<float *>(s_tab) + 2 * kHL,
          reinterpret_cast<const int4 *>(s_tab), reinterpret_cast<const int2 *>(smem_raw),
                                                            reinterpret_cast<unsigned *>(s_tab) + 2 * kHL + kRN, s_aux, n_cid, n_arc, n_iq, (unsigned)n_e, accept, lt_last__);
      else if (rmode == 1) created_total = lit_replay_components(p, q, sh, &ls.use_lds, reinterpret_cast<float *>(s_tab), (const int4 *)q.meta,
          (const int2 *)q.arcs2, reinterpret_cast<unsigned *>(smem_raw), q.par, n_cid, n_arc, n_iq, (unsigned)n_e, accept, lt_last__);
      else created_total = lit_replay_components(p, q, sh, &ls.use_lds, q.rcost, (const int4 *)q.meta, (const int2 *)q.arcs2,
          reinterpret_cast<unsigned *>(q.rflag), q.par, n_cid, n_arc, n_iq, (unsigned)n_e, accept, lt_last__);
#if defined(K3_LIT_PROF) && K3_LIT_PROF == 1
      if (tid == 0 && created_total < 0) sh.prof[5] += 1;
#endif
      if (created_total < 0) {      // back to the state step 2 left
        for (int c = tid; c < n_cid; c += kBlock) { const int i = q.c2t[c]; rcost[c] = i < n_e ? q.c0[i] : kInf; }
        __syncthreads();
      }
    }
#if K3_LIT_CAPTURE      // (the frame's second hash-order pass uses q.rlist as scratch: the roots, grouped by component in queue order, are kept in q.coffs -- dead by now -- for the host)
    if (p.cap) { int2 *keep = reinterpret_cast<int2 *>(q.coffs); for (int k = tid; k < n_iq; k += kBlock) keep[k] = q.rlist[k]; __syncthreads(); }
#endif
    if (created_total < 0) {
      if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);      // one wavefront on a dependent chain: let it issue ahead of the other workgroup's parallel phases
        int created = 0, err = 0, pops = 0;
        if (rmode == 0) lit_replay<0>(reinterpret_cast<float *>(s_tab) + 2 * kHL, reinterpret_cast<const int4 *>(s_tab),
            reinterpret_cast<const int2 *>(smem_raw), reinterpret_cast<int *>(smem_raw + (size_t)kRA * 8), kRS,
                                      reinterpret_cast<unsigned *>(s_tab) + 2 * kHL + kRN, q.iq, n_iq, accept, s_own, &created, &err, &pops);
        else if (rmode == 1) lit_replay<1>(reinterpret_cast<float *>(s_tab), q.meta, q.arcs2, reinterpret_cast<int *>(smem_raw + (size_t)kRA * 8), kRS,
                                           reinterpret_cast<unsigned *>(smem_raw), q.iq, n_iq, accept, s_own, &created, &err, &pops);
        else lit_replay<2>(q.rcost, q.meta, q.arcs2, q.stack, p.stack_cap, reinterpret_cast<unsigned *>(q.rflag), q.iq, n_iq, accept, s_own, &created, &err, &pops);
        __builtin_amdgcn_s_setprio(K3_LIT_BASE_PRIO);
#if defined(K3_LIT_PROF) && K3_LIT_PROF == 1
        if (lane == 0) { sh.prof[12] += pops; if (rmode != 0) sh.prof[3] += (long long)__builtin_readcyclecounter() - rp_t0; }
#endif
#ifdef K3_LIT_DEBUG
        if (lane == 0 && err) printf("lane %d frame %d: replay err %d pops %d created %d\n", L, f, err, pops, created);
#endif
        if (lane == 0) { ls.n_created = created; if (err) sh.err = err == 2 ? K3_ERR_HIP : K3_ERR_OVERFLOW; }
      }
      __syncthreads();
      if (block_err(sh)) break;
      created_total = ls.n_created;
      for (int k = tid; k < created_total; k += kBlock) K3_AST(&q.label[q.c2t[clist[k]]], (unsigned)n_e + (unsigned)k);      // creation ranks of the closure's tokens
    }
    K3_LT(9); K3_LQ(12);
#ifdef K3_LIT_DEBUG
    if (n_e + created_total != n && tid == 0) printf("lane %d frame %d: n_e %d created %d n %d n_cid %d n_arc %d n_iq %d rmode %d\n", L, f, n_e, created_total,
        n, n_cid, n_arc, n_iq, rmode);
#endif
#ifdef K3_LIT_STATS      // size statistics of the frames a fast (LDS-resident) path could take: n_cur and n <= K3_LIT_STATS tokens
    if (tid == 0 && f >= 0 && n_cur <= K3_LIT_STATS && n <= K3_LIT_STATS) {
      const long long el = eps_l1 - eps_l0; long long *P = p.prof + L * 16;
      auto mx = [&](int k, long long v) { if (v > P[k]) P[k] = v; };
      mx(0, n_cid); mx(1, n_arc); mx(2, n_iq); mx(3, el); mx(4, (long long)m_e + created_total);
      P[5] += 1;
      P[6] += n_cid > 1024;
      P[7] += n_cid > 1536;
      P[8] += n_iq > 1024;
      P[9] += el > 2048;
      P[10] += (long long)m_e + created_total > 32768;
      P[11] += n_cid;
      P[12] += n_arc;
      P[13] += n_iq;
      P[14] += el;
      P[15] += n_arc > 2048;
    }
#endif
    if (n_e + created_total != n) { if (tid == 0) sh.err = K3_ERR_HIP; }      // every token of the fixpoint must have been created by the replay
    if (block_err(sh)) break;
    __syncthreads();
    // ---- the frame's final HashList order (next frame's visit order; creation order for the final-frame sweeps)
    if (n <= kHoN) lit_hash_order_lds(q, sh, n, (unsigned)n, tok_state + nb, hash_size, ord_nxt, true, true, false, smem_raw, s_tab, lt_last__);
    else if (!lit_hash_order_mid(q, sh, arena, n, (unsigned)n, tok_state + nb, hash_size, ord_nxt, true, true, false) &&
             (p.lit_force_hbm_order || !lit_hash_order_big(q, sh, arena, n, (unsigned)n, tok_state + nb, hash_size, ord_nxt, true, true, false, lt_last__)))
      lit_hash_order(q, sh, n, (unsigned)n, tok_state + nb, hash_size, ord_nxt, true, false, lt_last__);
    for (int i = tid; i < kHL; i += kBlock) { tb.lkey[i] = kEmpty; tb.lcost[i] = kEncMax; tb.ltok[i] = -1; }      // the arena becomes the (empty) table again
    K3_LT(10); K3_LQ(13);
    // ---- publish the frame: final costs into the pool, empty table, idle labels
#if K3_LIT_CAPTURE      // (the host reads the frame's creation ranks out of q.label, then restores the idle pattern itself)
    if (p.cap && tid == 0 && f >= 0) { p.cap[0] = n_e; p.cap[1] = n; p.cap[2] = (long long)m_e; p.cap[3] = n_cid; p.cap[4] = n_arc; p.cap[5] = n_iq; p.cap[7] = created_total; p.cap[8] = (long long)hash_size; }
#else
    for (int i = tid; i < n; i += kBlock) K3_AST(&q.label[i], kLabelNone);
#endif
    __syncthreads();
    K3_LT(11); K3_LQ(14);
    cur_base = nb; n_cur = n; max_frame = n_cur > max_frame ? n_cur : max_frame; sel ^= 1;
    if (tid == 0 && f >= 0) cyc_general += (long long)__builtin_readcyclecounter() - cyc_t0;
#ifdef K3_LIT_FRAMECYC      // (< 0: general path, a given-up attempt on the LDS path included; cur_cutoff column = 1 when there was one)
    if (tid == 0 && f >= 0) { st_ab[f] = -(float)((long long)__builtin_readcyclecounter() - cyc_t0); }
#endif
    if (tid == 0) { tok_off[f + 2] = cur_base + n_cur; loff_e[f + 1] = sh.n_link; }
  }
  {
    const unsigned long long a = wave_sum_u64(cnt_eps), b = wave_sum_u64(cnt_emit), c = wave_sum_u64(cnt_os);
    if (lane == 0) {
      k3a_add(&sh.n_eps, a);
      k3a_add(&sh.n_emit, b);
      k3a_add(&sh.n_os, c);
    }
  }
  __syncthreads();
#ifdef K3_LIT_PROF
  if (tid < 16) p.prof[L * 16 + tid] += sh.prof[tid];
#elif defined(K3_FAST_PROF)
  if (tid == 0) {
    long long *P = p.prof + L * 16;
    for (int k = 0; k < 12; k++) P[k] += fs.prof[k];
    P[12] += n_fast;
    P[13] += n_gaveup;
    P[14] += n_general;
    P[15] += cyc_fast;
  }
#elif !defined(K3_LIT_STATS)
  if (tid == 0) {
    long long *P = p.prof + L * 16;
    for (int k = 0; k < 11; k++) P[k] += why[k];
    P[12] += n_fast;
    P[13] += n_gaveup;
    P[14] += n_general;
    P[15] += cyc_fast;
    P[11] += cyc_general;
  }
#endif
  if (tid == 0) {
    LaneInfo &li = p.info[L];
    li.n_tokens = cur_base + n_cur; li.n_links = sh.n_link; li.n_cands = (long long)sh.n_emit; li.n_eps = (long long)sh.n_eps; li.max_frame_tokens = max_frame;
    li.status = sh.err ? sh.err : status; li.num_frames = f0 + T; li.reached_final = 0; li.out_states = 0; li.out_arcs = 0;
    li.cur_base = cur_base; li.n_cur = n_cur; li.n_order_sensitive = (long long)sh.n_os; li.hash_size = (int)hash_size; li.order_sel = sel;
  }
  }      // (next lane of the work-queue)
#undef K3_LANE_DONE
}

#endif  // K3_LIT_FORWARD

#ifdef K3_LIT_FINAL      // the literal branch of the pruning kernel's last-frame stage: compiled by k3_decoder.hip
// PruneForwardLinksFinal (:385-467) exactly: in-place sweeps over the last frame's tokens in list order (newest token first) until no extra
// cost moves by more than 1e-5 relative (ApproxEqual, base/kaldi-math.h:265-275); a link is excised when it is found above the lattice beam.
// n tokens tb.., eps links [l0, l1).  base / extra: per token; off: first link of a token (n + 1); ldst (bit 31 = keep), ldelta: per live link.
__device__ __forceinline__ bool lit_approx_equal(float a, float b, float tol) {
  if (a == b) return true;
  const float diff = fabsf(a - b);
  if (diff == __builtin_inff() || diff != diff) return false;
  return diff <= tol * (fabsf(a) + fabsf(b));
}

template <bool PACKED>
__device__ __forceinline__ void lit_final_sweeps(const float *base, float *ex, const unsigned *off, int *ldst, const float *ldelta, const int *by_ins, int n, float lb) {
  const float kInf = __builtin_inff(); const unsigned omask = PACKED ? 0xFFFFu : 0xFFFFFFFFu;
  bool changed = true;
  for (int sweep = 0; changed && sweep < 100000; sweep++) {
    changed = false;
    for (int d = n - 1; d >= 0; d--) {      // active_toks_[frame].toks: newest token first
      const int t = PACKED ? (int)(off[d] >> 16) : by_ins[d];
      float te_ = base[t];
      for (unsigned j = off[t] & omask, je = off[t + 1] & omask; j < je; j++) {
        const int dst = ldst[j] & 0x7FFFFFFF;
        float le = ex[dst] + ldelta[j];
        if (le > lb) ldst[j] = dst;
        else { ldst[j] = dst | (int)0x80000000; if (le < 0.0f) le = 0.0f; if (le < te_) te_ = le; }
      }
      if (te_ > lb) te_ = kInf;
      if (!lit_approx_equal(ex[t], te_, 1.0e-05f)) changed = true;
      ex[t] = te_;
    }
  }
}

// The same sweeps WITHOUT the serial walk, for frames too large for LDS (where every step of the walk is a global round trip).  A token's
// new extra cost depends on the new values of the link destinations visited before it in the sweep (created later) and on the old values of the
// others: level(t) = 1 + max level over its earlier-visited destinations; the tokens of one level are independent, a sweep is one parallel
// pass per level with old / new values double-buffered.  Same arithmetic on the same operands as the walk, hence the same bits.  Returns false
// (nothing changed) when the dependency chains are deeper than kLvMax: the caller then walks.
constexpr int kLvMax = 62;
__device__ __forceinline__ bool lit_final_sweeps_parallel(const float *base, float *ex, const unsigned *off, int *ldst, const float *ldelta, const int *by_ins, int n, float lb,
                                                          int *dn, int *level, float *exn, int *blist) {
  __shared__ int s_lv[kLvMax + 2], s_cur[kLvMax + 2], s_flag;
  const int tid = threadIdx.x; const float kInf = __builtin_inff();
  for (int d = tid; d < n; d += kPBlock) { dn[by_ins[d]] = d; level[d] = 0; }
  if (tid == 0) s_flag = 0;
  __syncthreads();
  for (int round = 0;; round++) {
    for (int t = tid; t < n; t += kPBlock) {
      const int my = dn[t]; int lv = 0;
      for (unsigned j = off[t]; j < off[t + 1]; j++) { const int dst = ldst[j] & 0x7FFFFFFF; if (dn[dst] > my) { const int l_ = level[dst] + 1; lv = l_ > lv ? l_ : lv; } }
      if (lv != level[t]) { level[t] = lv; s_flag = 1; }
    }
    __syncthreads();
    const int f = s_flag;
    __syncthreads();
    if (!f) break;
    if (round > kLvMax) return false;
    if (tid == 0) s_flag = 0;
    __syncthreads();
  }
  if (tid < kLvMax + 2) { s_lv[tid] = 0; s_cur[tid] = 0; }
  __syncthreads();
  for (int t = tid; t < n; t += kPBlock) { const int lv = level[t]; if (lv > kLvMax) s_flag = 1; else atomicAdd(&s_lv[lv + 1], 1); }
  __syncthreads();
  if (s_flag) return false;
  if (tid == 0) { for (int l = 0; l <= kLvMax; l++) s_lv[l + 1] += s_lv[l]; }      // s_lv[l] = first position of level l
  __syncthreads();
  for (int t = tid; t < n; t += kPBlock) { const int lv = level[t]; blist[s_lv[lv] + atomicAdd(&s_cur[lv], 1)] = t; }
  __syncthreads();
  for (int sweep = 0; sweep < 100000; sweep++) {
    for (int lv = 0; lv <= kLvMax; lv++) {
      const int b0 = s_lv[lv], b1 = s_lv[lv + 1];
      if (b0 == n) break;      // (uniform: no tokens at this level or above)
      for (int x = b0 + tid; x < b1; x += kPBlock) {
        const int t = blist[x], my = dn[t];
        float te_ = base[t];
        for (unsigned j = off[t]; j < off[t + 1]; j++) {
          const int dst = ldst[j] & 0x7FFFFFFF;
          float le = (dn[dst] > my ? exn[dst] : ex[dst]) + ldelta[j];
          if (le > lb) ldst[j] = dst;
          else { ldst[j] = dst | (int)0x80000000; if (le < 0.0f) le = 0.0f; if (le < te_) te_ = le; }
        }
        if (te_ > lb) te_ = kInf;
        if (!lit_approx_equal(ex[t], te_, 1.0e-05f)) s_flag = 1;
        exn[t] = te_;
      }
      __syncthreads();
    }
    const int f = s_flag;
    for (int t = tid; t < n; t += kPBlock) ex[t] = exn[t];
    __syncthreads();
    if (!f) break;
    if (tid == 0) s_flag = 0;
    __syncthreads();
  }
  return true;
}

// The literal branch of k3_decode_prune_kernel's last-frame stage.  LDS arrays (each kPCap entries) are used when the frame fits.
__device__ __forceinline__ void lit_final_frame(const DecParams &p, int L, long long tb, long long te, long long l0, long long l1, float final_best, bool final_empty,
                                               const int *tok_state, const unsigned *tok_cost, float *extra, const Link *links, int *err,
                                               float *l_base, float *l_extra, unsigned *l_off, int *l_ldst, float *l_ldelta, int *redi, int *s_m) {
  const int tid = threadIdx.x; const int n = (int)(te - tb); const float lb = p.lattice_beam;
  const long long fc = p.frame_cands_cap;
  unsigned *g_off = reinterpret_cast<unsigned *>(p.c_dst + L * fc); unsigned *cursor = reinterpret_cast<unsigned *>(p.c_src + L * fc); int *lid = p.c_arc + L * fc;
  float *g_base = p.c_tot + L * fc, *g_delta = p.c_ac + L * fc; int *g_ldst = p.wl + 2ll * L * p.frame_tokens_cap;
  const int *by_ins = reinterpret_cast<const int *>(reinterpret_cast<const char *>(p.lt_by_ins) + p.lt_lane_bytes * L);
  for (int t = tid; t < n; t += kPBlock) K3_AST(&cursor[t], 0u);
  __syncthreads();
  for (long long l = l0 + tid; l < l1; l += kPBlock) { const Link k = links[l]; if (eps_link_live(k, tok_cost[k.src])) k3a_add(&cursor[k.src - tb], 1u); }
  __syncthreads();
  const int m = block_excl_scan([&](int t) { return K3_ALD(&cursor[t]); }, g_off, n, redi);
  if (tid == 0) { g_off[n] = (unsigned)m; *s_m = m; }
  if (m > fc || m > 2ll * p.frame_tokens_cap || n + 1 > fc) { if (tid == 0) *err = K3_ERR_OVERFLOW; return; }
  for (int t = tid; t < n; t += kPBlock) K3_AST(&cursor[t], 0u);
  __syncthreads();
  for (long long l = l0 + tid; l < l1; l += kPBlock) {
    const Link k = links[l];
    if (!eps_link_live(k, tok_cost[k.src])) continue;
    const unsigned pos = g_off[k.src - tb] + k3a_add(&cursor[k.src - tb], 1u);
    lid[pos] = (int)(l - l0); g_ldst[pos] = (int)(k.dst - tb); g_delta[pos] = k.tot - dec(tok_cost[k.dst]);      // (tot + ac + graph) - next_tok->tot_cost
  }
  for (int t = tid; t < n; t += kPBlock) { const float fcost = final_empty ? 0.0f : p.final_cost[tok_state[tb + t]]; g_base[t] = dec(tok_cost[tb + t]) + fcost - final_best; }
  __syncthreads();
  const bool use_lds = n + 1 <= kPCap && m <= kPCap;
  float *base = use_lds ? l_base : g_base, *ex = use_lds ? l_extra : extra + tb, *ldelta = use_lds ? l_ldelta : g_delta;
  unsigned *off = use_lds ? l_off : g_off;
  int *ldst = use_lds ? l_ldst : g_ldst;
  if (use_lds) {      // (offsets < 65536 here: the creation list rides in the upper halves, off[d] >> 16 = the token with creation rank d)
    for (int t = tid; t <= n; t += kPBlock) { off[t] = g_off[t] | (t < n ? (unsigned)by_ins[t] << 16 : 0u); if (t < n) base[t] = g_base[t]; }
    for (int j = tid; j < m; j += kPBlock) { ldst[j] = g_ldst[j]; ldelta[j] = g_delta[j]; }
  }
  for (int t = tid; t < n; t += kPBlock) ex[t] = 0.0f;      // a new token's extra_cost (:271)
  __syncthreads();
  // (two call sites so that the LDS case compiles to ds_ instructions: through a pointer chosen at run time every access of the serial walk
  // would be a flat access with the latency of a global one)
  if (use_lds) { if (tid == 0) lit_final_sweeps<true>(l_base, l_extra, l_off, l_ldst, l_ldelta, by_ins, n, lb); }
  else {
    const LitLane q(p, L);      // (forward-pass scratch, idle here)
    if (!lit_final_sweeps_parallel(g_base, extra + tb, g_off, g_ldst, g_delta, by_ins, n, lb, p.tok_slot + (long long)L * p.frame_tokens_cap,
        reinterpret_cast<int *>(cursor), q.c0, q.dense)) {
      if (tid == 0) lit_final_sweeps<false>(g_base, extra + tb, g_off, g_ldst, g_delta, by_ins, n, lb);
    }
  }
  __syncthreads();
  if (use_lds) { for (int t = tid; t < n; t += kPBlock) extra[tb + t] = ex[t]; for (int j = tid; j < m; j += kPBlock) g_ldst[j] = ldst[j]; }
  __syncthreads();
}
#endif  // K3_LIT_FINAL
