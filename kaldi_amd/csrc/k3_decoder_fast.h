// k3_decoder_fast.h -- one frame of literal_order token passing with every per-frame structure resident in LDS (included by k3_decoder_lit.hip
// behind k3_decoder_literal.h, inside its anonymous namespace).
//
// The general path of k3_decoder_literal.h keeps ~35 per-lane scratch arrays in HBM and pays a global round trip at the end of most of its ~100
// barrier-separated phases: a frame costs ~190 us whatever its size.  The usual frame (<= kFT tokens before and after, 95 % of the frames of the
// benchmark) fits in the 160 KB of LDS of a CU: the visit-ordered token list, the state -> token table, the tokens being built (cost, creation label,
// arc ranges -- fetched when the token is claimed, so that neither ProcessEmitting nor an epsilon round starts with an offsets look-up), the epsilon
// links of the closure, the closure sub-graph, the component replay and both HashList-order passes.  What goes to HBM is what the later kernels read
// (token pool, link pool, per-frame offsets) plus the visit order (a frame that does not fit is redone / continued on the general path from there), all
// fire-and-forget stores; what comes from HBM is the graph (arcs of the tokens expanded) and the frame's log-likelihoods.
// Same algorithm, same arithmetic, same results as the general path (phase by phase the code below follows it; the tests run both and compare both
// with the reference decoder).  A frame that exceeds any capacity of this path -- tokens, table window, epsilon links, closure ids / arcs, queue,
// labels, a stack slice -- ABORTS before it has published anything the next frame reads and is redone on the general path.

constexpr int kFT = 1536;       // tokens of a frame (current and next)
constexpr int kFH = 4096;       // state -> token table slots
constexpr int kFProbe = 64;     // probe window of the table (a full window gives the frame up)
constexpr int kFE = 1024;       // epsilon links written by the closure of a frame
constexpr int kFC = 768;        // closure ids (tokens that take part in the closure sub-graph)
constexpr int kFA = 1024;       // passing arcs of the closure sub-graph
constexpr int kFQ = 512;        // initial queue of the replay
constexpr int kFM = 8192;       // creation labels (emitting arcs examined + tokens the closure creates)
constexpr int kFW = 768;        // work-list of an epsilon round
constexpr int kFB = 2048;       // bucket table of the HashList-order passes
static_assert(kFT % 64 == 0 && kFT / 64 <= 64 && (kFH & (kFH - 1)) == 0 && (kFB & (kFB - 1)) == 0 && kFB > kFT && kFC % 2 == 0 && kFT % 2 == 0, "fast-path geometry");
// ---- LDS map (byte offsets into the arena; lifetimes in the phase list of lit_frame_fast).  Two workgroups share a CU's 160 KB: the whole frame has to fit
// in the 77.5 KB the general path uses, so every region below is reused by the phases that follow its last reader.
// (1) the tokens being built, alive for the whole frame
constexpr int oN_cost = 0, oN_abeg = oN_cost + 4 * kFT, oN_ne = oN_abeg + 4 * kFT, oLab16 = oN_ne + 2 * kFT, oX = oLab16 + 2 * kFT /* labels (u32) -> expanded-at cost -> bucket16 | ord1 */, oN_nn = oX + 4 * kFT;
// (2) the current frame in visit order (read until pass B, written by the last phase); in between: c0, the 32-bit columns of the epsilon links, order / replay scratch
constexpr int oV_cost = oN_nn + 2 * kFT, oV_abeg = oV_cost + 4 * kFT, oV_ne = oV_abeg + 4 * kFT, oV_tok = oV_ne + 2 * kFT, oV_end = oV_tok + 2 * kFT;
constexpr int oC0 = oV_cost, oE_arc = oV_abeg, oE_stamp = oE_arc + 4 * kFE, oE_w = oE_stamp + 4 * kFE;
// (3) the state -> token table (pass B .. closure); afterwards the closure sub-graph
constexpr int oT_key = oV_end, oT_tix = oT_key + 4 * kFH, oT_end = oT_tix + 2 * kFH;
// (4) work-lists, mark bits, chunk records, the 16-bit columns of the epsilon links
constexpr int oWl = oT_end, oMarks = oWl + 2 * 2 * kFW, oChunk = oMarks + 3 * (kFT / 32) * 4, oE_src = oChunk + 512, oE_dst = oE_src + 2 * kFE, kFastArena = oE_dst + 2 * kFE;
// closure sub-graph (over the table and the work-lists)
constexpr int oRown = oT_key, oCid = oRown + 4 * kFT, oLead = oCid + 2 * kFT, oAR_arc = oLead + 2 * kFT, oAR_dst = oAR_arc + 4 * kFA,
    oAR_w = oAR_dst + 2 * kFA, oM_abeg = oAR_w + 4 * kFA, oM_pc = oM_abeg + 2 * kFC,
              oC2t = oM_pc + 2 * kFC, oRflag = oC2t + 2 * kFC, oSrcbit = oRflag + kFT / 8, oSub_end = oSrcbit + kFT / 8;
constexpr int oRcost = oE_arc;      // (over E_arc, dead after the second pass over the links)
// order pass 1 (emitting tokens, between the sub-graph and the replay): in the holes the links and the build scratch left
constexpr int oO1_btab = oE_stamp, oO1_bm = oRown, oO1_wpre = oO1_bm + kFM / 8, oO1_lead = oO1_wpre + kFM / 32 * 2, oO1_grp = oLead, oO1_curs = oAR_arc, oOrd1 = oX + 2 * kFT;
// order pass 2 (all tokens, last phase): everything but (1) is dead; the result goes straight to (2)
constexpr int oO2_btab = oT_key, oO2_bm = oO2_btab + 4 * kFB, oO2_wpre = oO2_bm + kFM / 8, oO2_lead = oO2_wpre + kFM / 32 * 2,
    oO2_grp = oO2_lead + 2 * kFT + 8, oO2_curs = oO2_grp + 2 * kFT + 8, oO2_end = oO2_curs + 2 * kFT + 8;
// replay (after order pass 1)
constexpr int oIq = oE_stamp, oPar = oIq + 2 * kFQ, oCroots = oPar + 4 * kFC, oCcreated = oCroots + 2 * kFC + 8, oCarcs = oCcreated + 2 * kFC + 8,
    oCcurs = oCarcs + 2 * kFC + 8, oOx = oCcurs + 2 * kFC + 8, oOy = oOx + 2 * kFC, oOz = oOy + 2 * kFC,
              oOw = oOz + 2 * kFC, oRlist = oOw + 2 * kFC, oStack = oRlist + 4 * kFQ, oRep_end = oStack + 2 * kFA;
constexpr int oRinfo = oOrd1 /* consumed by then */, oRtmp = oN_nn, oDense = oRtmp + 2 * kFQ, oWrec = oC0 /* consumed by then */, oClist = oE_src;
static_assert(kFastArena <= 79360, "two workgroups per CU: the frame lives in the 77.5 KB of the general path");
static_assert(oE_w + 4 * kFE <= oV_end && oSub_end <= oChunk && oRcost + 4 * kFC <= oE_stamp && oO1_btab + 4 * kFB <= oV_end + 0 * kFB &&
    oO1_lead + 2 * kFT + 8 <= oCid && oO1_grp + 2 * kFT <= oAR_arc &&
              oO1_curs + 2 * kFT + 8 <= oAR_dst && oOrd1 + 2 * kFT <= oN_nn && oRep_end <= oAR_dst && oRinfo + 4 * kFQ <= oN_nn && oDense + 2 * kFQ <= oV_cost
                  && oWrec + 10 * kFQ <= oRcost &&
              oClist + 2 * kFT <= kFastArena && oO2_end <= kFastArena && 2 * kFT <= 4 * kFT / 2 + 2 * kFT, "LDS map");

struct LaneCtx {      // this lane's slices of the pools and per-frame arrays
  int *tok_state; unsigned *tok_cost; Link *links; int *link_arc; long long *tok_off, *loff_e, *loff_n; int *st_ntoks; float *st_cur, *st_ab, *st_next, *st_co;
  // capacities of the lane's token / link pools (the caller has made room for a whole LDS-resident frame: the checks below cannot fire, they guard the memory)
  long long tcap, lcap;
};
// cnt: the frame's three running counts in one word -- tokens created [0, 16), first work-list entries [16, 32), forward links made [32, 64) -- so that a wavefront step
// reserves its token indices, work-list slots and link slots with ONE LDS atomic (three dependent round trips before)
struct FastShared { unsigned long long cnt; int abort, abort_r[4], n_wl[3], n_el; unsigned next0; int reason; long long prof[12]; };
__device__ __forceinline__ void wave_append3(bool a, bool b, bool c, unsigned long long *counter, int &ia, int &ib, long long &ic) {
  const int lane = threadIdx.x & 63;
  const unsigned long long ma = __ballot(a), mb = __ballot(b), mc = __ballot(c), any = ma | mb | mc;
  ia = 0; ib = 0; ic = 0;
  if (any == 0) return;
  const int leader = __ffsll((long long)any) - 1;
  unsigned long long base = 0;
  if (lane == leader) base = k3a_add(counter, (unsigned long long)__popcll(ma) | ((unsigned long long)__popcll(mb) << 16) | ((unsigned long long)__popcll(mc) << 32));
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)base, leader), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(base >> 32), leader);
  const unsigned long long lt = (1ull << lane) - 1ull;
  ia = (int)(lo & 0xFFFFu) + __popcll(ma & lt); ib = (int)(lo >> 16) + __popcll(mb & lt); ic = (long long)hi + __popcll(mc & lt);
}
#ifdef K3_FAST_PROF
#define K3_FP(i) do { if (threadIdx.x == 0) { const long long now__ = (long long)__builtin_readcyclecounter(); fs.prof[i] += now__ - fp_last__; fp_last__ = now__; } } while (0)
#elif defined(K3_FAST_MARK)      // -S builds: a comment in the ISA at every phase boundary (tools/count_fast_isa.py counts the instructions between them)
#define K3_FP(i) asm volatile("; K3MARK " #i)
#else
#define K3_FP(i) do { } while (0)
#endif
enum { kFaTokens = 1, kFaTable, kFaHash, kFaLabels, kFaWl, kFaLinks, kFaDegree, kFaClosure, kFaQueue, kFaStack, kFaMismatch, kFaPool };

__device__ __forceinline__ unsigned lds_ld(const unsigned *p_) { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_ld(const int *p_) { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned short lds_ld16(const unsigned short *p_) { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// 16-bit counters packed two to a word (LDS has no 16-bit atomics): fetch-add on one half
__device__ __forceinline__ unsigned add16(unsigned short *base, int i, unsigned v) {
  unsigned *w = reinterpret_cast<unsigned *>(base) + (i >> 1); const int sh_ = (i & 1) * 16;
  return (k3a_add(w, v << sh_) >> sh_) & 0xFFFFu;
}

// exclusive prefix sum of in(i), i < n, handed to out(i, sum); returns the total.  All threads call it.
template <typename In, typename Out>
__device__ __forceinline__ int block_excl_scan_f(In &&in, Out &&out, int n, int *redi) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (int)blockDim.x >> 6;
  const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x;      // every thread takes `per` consecutive items
  const int b = tid * per, e = b + per < n ? b + per : n; int sum = 0;
  for (int i = b; i < e; i++) sum += (int)in(i);
  const int incl = wave_incl_scan(sum);
  __syncthreads();
  if (lane == 63) redi[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
  for (int w = 0; w < nw; w++) { const int s = redi[w]; if (w < wave) woff += s; tot += s; }
  int run = woff + incl - sum;
  for (int i = b; i < e; i++) { const int x = (int)in(i); out(i, run); run += x; }      // (in(i) is read again before out(i) overwrites it: in-place scans are fine)
  __syncthreads();
  return tot;
}

// HashList order of n <= kFT tokens (label16[i] < M <= kFM unique creation labels, bkt16[i] = state % hash_size): emit(position, token, creation rank).
// The phase structure of lit_hash_order_lds; the bucket table packs {bucket, smallest creation rank} into one word (equal buckets -> equal upper
// halves, so the minimum over the word is the minimum over the ranks) and the members of a bucket are counted at its leader's rank.
// kDense: the labels already ARE dense creation ranks (0 .. n-1, the frame's second pass: the first pass left the emitting tokens' ranks in label16 and the closure numbers its
// tokens behind them), so the label bitmap, its prefix counts and their four barriers are skipped.
template <bool kDense, typename Emit>
__device__ __forceinline__ void fast_hash_order(Shared &sh, int n, unsigned M, const unsigned short *label16, const unsigned short *bkt16, unsigned *btab,
    unsigned *bm, unsigned short *wpre,
                                                unsigned short *lead, unsigned short *grp, unsigned short *curs, Emit &&emit) {
  const int tid = threadIdx.x; constexpr int kPer = (kFT + kBlock - 1) / kBlock;
  const int W = kDense ? 0 : (int)((M + 31u) >> 5);
  unsigned lab[kPer], bkt[kPer], lf[kPer], cnt[kPer]; int d[kPer], slot[kPer];
  for (int i = tid; i < kFB; i += kBlock) btab[i] = 0xFFFFFFFFu;
  for (int i = tid; i < W; i += kBlock) bm[i] = 0u;
  for (int i = tid; i < (n + 1) / 2 + 1; i += kBlock) { reinterpret_cast<unsigned *>(lead)[i] = 0u; reinterpret_cast<unsigned *>(curs)[i] = 0u; }
#pragma unroll
  for (int k = 0; k < kPer; k++) { const int i = tid + k * kBlock; lab[k] = 0; bkt[k] = 0; if (i < n) { lab[k] = label16[i]; bkt[k] = bkt16[i]; } }
  __syncthreads();
  if (!kDense) {
#pragma unroll
    for (int k = 0; k < kPer; k++) { const int i = tid + k * kBlock; if (i < n) k3a_or(&bm[lab[k] >> 5], 1u << (lab[k] & 31)); }
    __syncthreads();
    block_excl_scan_f([&](int w) { return __popc(bm[w]); }, [&](int w, int ex) { wpre[w] = (unsigned short)ex; }, W, sh.redi);
  }
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int i = tid + k * kBlock;
    if (i < n) {
      const unsigned l = lab[k]; d[k] = kDense ? (int)l : (int)wpre[l >> 5] + __popc(bm[l >> 5] & ((1u << (l & 31)) - 1u));
      const unsigned mine = (bkt[k] << 16) | (unsigned)d[k];
      unsigned h = (bkt[k] * 2654435761u) >> 21;      // 11 bits: kFB = 2048
      for (;;) {
        unsigned w = lds_ld(&btab[h]);
        if (w == 0xFFFFFFFFu) { const unsigned old = k3a_cas(&btab[h], 0xFFFFFFFFu, mine); if (old == 0xFFFFFFFFu) break; w = old; }
        if ((w >> 16) == bkt[k]) { k3a_min(&btab[h], mine); break; }
        h = (h + 1) & (kFB - 1);
      }
      slot[k] = (int)h;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPer; k++) { const int i = tid + k * kBlock; if (i < n) { lf[k] = btab[slot[k]] & 0xFFFFu; add16(lead, (int)lf[k], 1u); } }
  __syncthreads();
  bool multi = false;
#pragma unroll
  for (int k = 0; k < kPer; k++) { const int i = tid + k * kBlock; cnt[k] = 1; if (i < n) { cnt[k] = lead[lf[k]]; multi |= cnt[k] > 1u; } }
  multi = __syncthreads_or(multi);
  // (reads precede the writes: every thread scans its own consecutive ranks)
  block_excl_scan_f([&](int r) { return (int)lead[r]; }, [&](int r, int ex) { lead[r] = (unsigned short)ex; }, n, sh.redi);
  if (multi) {
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int i = tid + k * kBlock;
      if (i < n && cnt[k] > 1u) {
        const unsigned s_ = add16(curs, (int)lf[k], 1u);
        grp[lead[lf[k]] + s_] = (unsigned short)d[k];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int i = tid + k * kBlock;
    if (i < n) {
      const unsigned lp = lead[lf[k]]; unsigned rank = 0;
      if (cnt[k] > 1u) for (unsigned t = 0; t < cnt[k]; t++) rank += (int)grp[lp + t] < d[k];
      emit((int)(lp + rank), i, d[k]);
    }
  }
  __syncthreads();
}

// One component of the replay on one thread (lit_replay_component with 16-bit LDS records)
__device__ __forceinline__ bool fast_replay_component(float *rcost, const unsigned short *m_abeg, const unsigned short *m_pc, const unsigned short *ar_dst,
    const float *ar_w, unsigned short *clist,
                                                      const unsigned short *rlist, unsigned short *rinfo, unsigned short *stk, int scap, int r0, int rcnt, int cpos, float accept) {
  const float kInf = __builtin_inff();
  for (int j = 0; j < rcnt; j++) {
    const int k = rlist[2 * (r0 + j)]; int cur = rlist[2 * (r0 + j) + 1]; const int seg0 = cpos; int sp = 0;
    for (;;) {
      const float cc = rcost[cur]; const int abeg = m_abeg[cur], pc = m_pc[cur];
      int nxt = -1;
      if (cc < accept) {
        for (int a = 0; a < pc; a++) {
          const int d = ar_dst[abeg + a]; const float tot = cc + ar_w[abeg + a];
          if (tot < accept) {
            const float old = rcost[d]; const int dpc = m_pc[d];      // (both behind `d`, requested together: the chain is one LDS round trip shorter per arc)
            if (old > tot) {
              if (old == kInf) clist[cpos++] = (unsigned short)d;
              rcost[d] = tot;
              if (dpc > 0) { if (nxt >= 0) { if (sp >= scap) return false; stk[sp++] = (unsigned short)nxt; } nxt = d; }
            }
          }
        }
      }
      if (nxt >= 0) cur = nxt; else if (sp > 0) cur = stk[--sp]; else break;
    }
    rinfo[2 * k] = (unsigned short)seg0; rinfo[2 * k + 1] = (unsigned short)(cpos - seg0);
  }
  return true;
}

// The current frame's tokens into the visit-order arrays (after a frame of the general path, or at the start of a launch)
__device__ __forceinline__ bool fast_import(const DecParams &p, char *arena, const LaneCtx &c, const int *ord_cur, long long cur_base, int n_cur, FastShared &fs) {
  unsigned *V_cost = reinterpret_cast<unsigned *>(arena + oV_cost), *V_abeg = reinterpret_cast<unsigned *>(arena + oV_abeg);
  unsigned short *V_ne = reinterpret_cast<unsigned short *>(arena + oV_ne), *V_tok = reinterpret_cast<unsigned short *>(arena + oV_tok);
  if (threadIdx.x == 0) fs.abort = 0;
  __syncthreads();
  for (int r = threadIdx.x; r < n_cur; r += kBlock) {
    const int i = ord_cur[r]; const int st = c.tok_state[cur_base + i]; const int2 a = p.offs[st];
    V_cost[r] = c.tok_cost[cur_base + i]; V_abeg[r] = (unsigned)a.x; V_tok[r] = (unsigned short)i;
    const int ne = a.y - a.x; if (ne > 65535) fs.abort = 1;
    V_ne[r] = (unsigned short)ne;
  }
  __syncthreads();
  const bool ok = fs.abort == 0;
  __syncthreads();
  return ok;
}

#ifndef K3_FAST_INLINE
#define K3_FAST_INLINE __forceinline__
#endif
// One frame.  Returns the number of tokens of the new frame, or -1 when the frame has to be redone on the general path (nothing the next frame reads
// has been published; sh.n_link / sh.n_next are restored by the caller).
__device__ K3_FAST_INLINE int lit_frame_fast(const DecParams &p, Shared &sh, FastShared &fs, char *arena, const LitLane &q, const LaneCtx &c, int f,
    const float *ll, long long cur_base, int n_cur,
                                              unsigned &hash_size_io, int *ord_nxt, int cap_tokens, unsigned &cnt_emit_io, unsigned &cnt_os_io, unsigned &cnt_eps_io) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6; constexpr int nw = kBlock / 64; const float kInf = __builtin_inff();
  unsigned *N_cost = reinterpret_cast<unsigned *>(arena + oN_cost), *N_abeg = reinterpret_cast<unsigned *>(arena + oN_abeg), *X = reinterpret_cast<unsigned *>(arena + oX);
  unsigned short *N_ne = reinterpret_cast<unsigned short *>(arena + oN_ne), *N_nn = reinterpret_cast<unsigned short *>(arena + oN_nn),
      *lab16 = reinterpret_cast<unsigned short *>(arena + oLab16);
  unsigned short *B16 = reinterpret_cast<unsigned short *>(arena + oX);
  unsigned *V_cost = reinterpret_cast<unsigned *>(arena + oV_cost), *V_abeg = reinterpret_cast<unsigned *>(arena + oV_abeg);
  unsigned short *V_ne = reinterpret_cast<unsigned short *>(arena + oV_ne), *V_tok = reinterpret_cast<unsigned short *>(arena + oV_tok);
  int *T_key = reinterpret_cast<int *>(arena + oT_key); unsigned short *T_tix = reinterpret_cast<unsigned short *>(arena + oT_tix);
  unsigned short *wl = reinterpret_cast<unsigned short *>(arena + oWl); unsigned *marks = reinterpret_cast<unsigned *>(arena + oMarks);
  unsigned *cmin = reinterpret_cast<unsigned *>(arena + oChunk); int *ccnt = reinterpret_cast<int *>(arena + oChunk + 256);
  float *c0 = reinterpret_cast<float *>(arena + oC0);
  unsigned *E_arc = reinterpret_cast<unsigned *>(arena + oE_arc), *E_stamp = reinterpret_cast<unsigned *>(arena + oE_stamp); float *E_w = reinterpret_cast<float *>(arena + oE_w);
  unsigned short *E_src = reinterpret_cast<unsigned short *>(arena + oE_src), *E_dst = reinterpret_cast<unsigned short *>(arena + oE_dst);
  unsigned cnt_emit = 0, cnt_os = 0, cnt_eps = 0;
#if K3_LIT_PREFETCH_ARCS
  int arc_pf = 0;
#endif
  auto abort_now = [&](int why) { fs.abort = 1; fs.reason = why; };
  auto aborted = [&]() { __syncthreads(); const int a = fs.abort; __syncthreads(); return a != 0; };      // uniform snapshot between two barriers

  long long fp_last__ = (long long)__builtin_readcyclecounter(); (void)fp_last__;
#if K3_LIT_PREFETCH_ROW
  // The frame's log-likelihood row (24 KB for 6024 pdfs), requested now, coalesced: the ~2000 scattered 4-byte reads of passes A / B then find their lines in the L2 instead of
  // missing to HBM one by one (a miss is on the critical path of every 64-arc step).  The values are summed into a register that is looked at once, behind the cutoff phase
  // (which touches LDS only: nothing waits for these loads before that).
  float row_pf = 0.0f;
  for (int i = tid * 4; i + 3 < p.num_pdfs; i += kBlock * 4) { const float4 v = *reinterpret_cast<const float4 *>(ll + i); row_pf += v.x + v.y + v.z + v.w; }
#endif
  // ---- GetCutoff (:653-720) on the visit-ordered costs
  unsigned long long bm = ~0ull;
  for (int r = tid; r < n_cur; r += kBlock) { const unsigned long long v = ((unsigned long long)V_cost[r] << 32) | (unsigned)r; bm = v < bm ? v : bm; }
  bm = block_min_u64(bm, sh);
  const float best = dec((unsigned)(bm >> 32)); const int best_r = (int)(unsigned)(bm & 0xFFFFFFFFull);
  auto for_keys = [&](auto fn) { for (int i0 = 0; i0 < n_cur; i0 += kBlock) { const int i = i0 + tid; fn(i < n_cur, i < n_cur ? V_cost[i] : 0u); } };
  float cur_cutoff, ab;
  const float beam_cutoff = best + p.beam;
  int deg_beam = -1;      // emitting arcs of the tokens within beam_cutoff (what pass A will count when the cutoff is the beam's, the usual case): summed in the counting pass
  if (p.max_active == 0x7FFFFFFF && p.min_active == 0) { ab = p.beam; cur_cutoff = beam_cutoff; }
  else {
    const unsigned ebc = enc(beam_cutoff);
    int c_lt = 0, c_le = 0, dsum = 0;
    for (int i = tid; i < n_cur; i += kBlock) { const unsigned k = V_cost[i]; c_lt += k < ebc; c_le += k <= ebc; if (k <= ebc) dsum += (int)V_ne[i]; }
    {      // (n_cur <= kFT < 65536: the two counts share a word; one barrier pair for both sums)
      int both = wave_sum_i32((c_lt << 16) | c_le); dsum = wave_sum_i32(dsum);
      __syncthreads();
      if (lane == 0) { sh.redi[wave] = both; sh.hist[wave] = dsum; }
      __syncthreads();
      both = 0; dsum = 0;
      for (int w = 0; w < nw; w++) { both += sh.redi[w]; dsum += sh.hist[w]; }
      c_lt = both >> 16; c_le = both & 0xFFFF; deg_beam = dsum;
    }
    int kth = -1;
    if (n_cur > p.max_active && c_lt > p.max_active) kth = p.max_active;
    else if (n_cur > p.min_active && p.min_active == 0) { ab = p.beam; cur_cutoff = beam_cutoff; }
    else if (n_cur > p.min_active && c_le > p.min_active) { ab = p.beam; cur_cutoff = beam_cutoff; }
    else if (n_cur > p.min_active) kth = p.min_active;
    else { ab = kInf - best + p.beam_delta; cur_cutoff = kInf; }
    if (kth >= 0) { const float sel_ = dec(block_select_kth(for_keys, kth, sh)); ab = sel_ - best + p.beam_delta; cur_cutoff = sel_; }
  }
  unsigned hash_size = hash_size_io;
  { const unsigned want = (unsigned)((float)n_cur * p.hash_ratio); if (want > hash_size) hash_size = want; }      // PossiblyResizeHash (:227-233)
  if (hash_size > 65535u || hash_size > (unsigned)p.hash_cap) return -1;      // (uniform; nothing touched yet)
  const float co = -best;
  const long long nb = cur_base + n_cur; const long long link0 = sh.n_link;
  {      // the number of emitting arcs pass A would count (m_e below) from the per-token degrees: a frame whose labels cannot fit (the first frame of an utterance: a hundred
    // tokens with tens of thousands of arcs, walked by two wavefronts) leaves before that walk
    int deg_sum = deg_beam;
    if (deg_beam < 0 || cur_cutoff != beam_cutoff) {
      deg_sum = 0;
      for (int r = tid; r < n_cur; r += kBlock) if (dec(V_cost[r]) <= cur_cutoff) deg_sum += (int)V_ne[r];
      deg_sum = block_sum_i32(deg_sum, sh);
    }
    if ((unsigned)deg_sum + (unsigned)cap_tokens > (unsigned)kFM) return -1;      // (uniform; nothing touched yet)
  }
  K3_FP(0);
#if K3_LIT_PREFETCH_ROW
  if (row_pf == 1.2345678e33f) fs.reason = -7;      // (never true: keeps the row's loads alive)
#endif
  // ---- the frame's structures
  if (tid == 0) {
    fs.abort = 0;
    fs.reason = 0;
    fs.next0 = kEncMax;
    fs.n_wl[0] = 0;
    fs.n_wl[1] = 0;
    fs.n_wl[2] = 0;
    fs.abort_r[0] = fs.abort_r[1] = fs.abort_r[2] = fs.abort_r[3] = 0;
    fs.cnt = 0ull;
    c.loff_e[f] = link0;
  }
  for (int i = tid; i < cap_tokens; i += kBlock) { N_cost[i] = kEncMax; X[i] = kLabelNone; }
  for (int i = tid; i < kFH; i += kBlock) T_key[i] = kEmpty;
  for (int i = tid; i < kFH / 2; i += kBlock) reinterpret_cast<unsigned *>(T_tix)[i] = 0xFFFFFFFFu;
  for (int i = tid; i < 3 * (kFT / 32); i += kBlock) marks[i] = 0u;
  __syncthreads();
  // ---- pass A (:779-797 first half): per 64-token chunk of the visit order the number of emitting arcs and min (tot + adaptive_beam); the pre-pass
  // (:753-768) rides along: the best token's arcs are among them
  // A chunk = the 2^csh consecutive tokens of the visit order a wavefront expands together (their arcs one per lane).  64 tokens for the large frames; fewer for a frame
  // that would otherwise give the eight wavefronts less than two chunks each (600 tokens = 10 chunks of ~80 arcs: two wavefronts walk four 64-arc steps while six walk two;
  // 19 chunks of ~40 arcs: three steps at most), down to 8 tokens.  At most 64 chunks: the scan below is one chunk per lane.
  int csh = 6;
  while (K3_LIT_CSH && csh > 3 && ((n_cur + (1 << csh) - 1) >> csh) < 2 * nw && ((n_cur + (1 << (csh - 1)) - 1) >> (csh - 1)) <= 64) csh--;
  const int nchunks = (n_cur + (1 << csh) - 1) >> csh;
  auto chunk_tokens = [&](int ch, float &cost, int &beg, int &deg, int &vtok) {
    const int r = (ch << csh) + lane; const bool v = lane < (1 << csh) && r < n_cur;
    cost = v ? dec(V_cost[r]) : 0.0f; beg = 0; deg = 0; vtok = v ? (int)V_tok[r] : 0;
    if (v && cost <= cur_cutoff) { beg = (int)V_abeg[r]; deg = (int)V_ne[r]; }
  };
  for (int ch = wave; ch < nchunks; ch += nw) {
    float cost; int beg, deg, vtok; chunk_tokens(ch, cost, beg, deg, vtok);
    unsigned cm = kEncMax, pm = kEncMax;
    const int total = wave_expand_seq(p.arcs, beg, deg, [&](bool valid, int, int, int owner, const ArcRec &r) {
      const float oc = __shfl(cost, owner);
      if (valid) {
        const float llv = ll[r.pdf]; const float ac = co - llv; const float tot = oc + ac + r.w; const unsigned e = enc(tot + ab); cm = e < cm ? e : cm;
        if ((ch << csh) + owner == best_r) { const float nw_ = r.w + co - llv + best; const unsigned e0 = enc(nw_ + ab); pm = e0 < pm ? e0 : pm; }
      }
      cnt_emit += valid;
    });
    cm = wave_min_u32(cm); pm = wave_min_u32(pm);
    if (lane == 0) { cmin[ch] = cm; ccnt[ch] = total; if (pm != kEncMax) k3a_min(&fs.next0, pm); }
  }
  __syncthreads();
  K3_FP(1);
  // exclusive scans over the chunks (<= 48: one per lane; every wavefront does them for itself)
  float accept; unsigned m_e; unsigned cpre_l; int cbase_l;
  {
    const unsigned n0 = fs.next0; const float next0 = n0 == kEncMax ? kInf : dec(n0); const unsigned run0 = enc(next0);
    const unsigned m = lane < nchunks ? cmin[lane] : kEncMax; const int k = lane < nchunks ? ccnt[lane] : 0;
    unsigned em = m; int ik = k;
    em = wave_incl_min_u32(em); ik = wave_incl_sum_i32(ik);
    unsigned exm = wave_shr1_u32(em, kEncMax); exm = run0 < exm ? run0 : exm;
    cpre_l = exm; cbase_l = ik - k;
    const unsigned wm = (unsigned)__builtin_amdgcn_readlane((int)em, 63); accept = dec(wm < run0 ? wm : run0); m_e = (unsigned)__builtin_amdgcn_readlane(ik, 63);
  }
  if (m_e + (unsigned)cap_tokens > (unsigned)kFM) return -1;      // (uniform; only LDS touched so far)
  // ---- pass B: accept against the bound in force at each arc; tokens, costs, creation labels, forward links
  for (int ch = wave; ch < nchunks; ch += nw) {
    float cost; int beg, deg, vtok; chunk_tokens(ch, cost, beg, deg, vtok);
    unsigned run = __shfl(cpre_l, ch); const int jbase = __shfl(cbase_l, ch);
    wave_expand_seq(p.arcs, beg, deg, [&](bool valid, int j, int arc, int owner, const ArcRec &r) {
      const float oc = __shfl(cost, owner); const int oi = __shfl(vtok, owner);
      float ac = 0.0f, tot = 0.0f; unsigned e = kEncMax;
      if (valid) { ac = co - ll[r.pdf]; tot = oc + ac + r.w; e = enc(tot + ab); }
      unsigned em = e;
      em = wave_incl_min_u32(em);
      unsigned exm = wave_shr1_u32(em, kEncMax); exm = run < exm ? run : exm;
      { const unsigned wm = (unsigned)__builtin_amdgcn_readlane((int)em, 63); run = wm < run ? wm : run; }
      const bool acc = valid && tot < dec(exm);
      cnt_os += acc && !(tot < accept);
      const int state = (int)((unsigned)r.next & ~kEpsFlag);
      bool claimed = false, mk = false; int slot = -1;
      if (acc) {
        unsigned h = hash_state(state) & (kFH - 1);
        for (int probe = 0; probe < kFProbe; probe++) {
          int k = lds_ld(&T_key[h]); bool cl = false;
          if (k == kEmpty) { const int old = k3a_cas(&T_key[h], kEmpty, state); if (old == kEmpty) { cl = true; k = state; } else k = old; }
          if (k == state) { slot = (int)h; claimed = cl; break; }
          h = (h + 1) & (kFH - 1);
        }
        if (slot < 0) abort_now(kFaTable); else mk = true;
      }
      const bool q1 = claimed && r.next < 0;
      int idx, pos1; long long lpos;
      wave_append3(claimed, q1, mk, &fs.cnt, idx, pos1, lpos);      // token index / work-list slot / link slot of this arc, where it needs them
      int2 oa = make_int2(0, 0), ob = make_int2(0, 0);
      if (claimed) {
        if (idx >= cap_tokens || nb + idx >= c.tcap) { abort_now(kFaTokens); idx = 0; }
        else { c.tok_state[nb + idx] = state; oa = p.offs[state]; ob = p.offs[state + 1]; }
        __hip_atomic_store(&T_tix[slot], (unsigned short)idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (q1) { if (pos1 < kFW) wl[kFW + pos1] = (unsigned short)idx; else abort_now(kFaWl); }
      if (mk && !claimed) {
        for (int spin = 0;; spin++) {
          const unsigned short t = lds_ld16(&T_tix[slot]);
          if (t != 0xFFFFu) {
            idx = t;
            break;
          }
          if (spin > (1 << 22)) {
            sh.err = K3_ERR_HIP;
            idx = 0;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (mk) { k3a_min(&N_cost[idx], enc(tot)); k3a_min(&X[idx], (unsigned)(jbase + j)); }
      const long long pos = link0 + lpos;
      if (mk) {
        if (pos < c.lcap) { store_link(&c.links[pos], Link{(unsigned)(cur_base + oi), (unsigned)(nb + idx), tot, ac}); store_stream(&c.link_arc[pos], arc); }
        else abort_now(kFaPool);
      }
      if (claimed) {      // the new token's arc ranges (requested above, arrived by now): neither the next frame nor an epsilon round starts with an offsets look-up
        // (a per-arc destination record loaded beside every arc instead -- DecParams::dinfo, as the epsilon rounds do -- costs this pass 2 k cycles: a scattered 8-byte load per
        // arc examined against two per token created)
        const int ne = oa.y - oa.x, nn = ob.x - oa.y;
        if (ne > 65535 || nn > 65535) abort_now(kFaDegree);
        N_abeg[idx] = (unsigned)oa.x; N_ne[idx] = (unsigned short)ne; N_nn[idx] = (unsigned short)nn;
      }
    });
  }
  if (aborted()) return -1;
  const unsigned long long cnt_e = fs.cnt;      // (uniform: read between the barriers of aborted() and the next phase's; the rounds below add to it only after another barrier)
  const int n_e = (int)(cnt_e & 0xFFFFull); const long long eps_l0 = link0 + (long long)(cnt_e >> 32);
  K3_FP(2);
  if (tid == 0) { c.loff_n[f + 1] = eps_l0; c.st_ntoks[f] = n_cur; c.st_cur[f] = cur_cutoff; c.st_ab[f] = ab; c.st_next[f] = accept; c.st_co[f] = co; }
  // X: creation labels -> "expanded at" costs
  for (int i = tid; i < cap_tokens; i += kBlock) {
    if (i < n_e) {
      c0[i] = dec(N_cost[i]);
      lab16[i] = (unsigned short)X[i];
    }
    X[i] = kEncMax;
  }
  __syncthreads();
  K3_FP(3);
  // ---- ProcessNonemitting (:830-897): the order-free fixpoint of finish_frame on token indices; every epsilon link also stays in LDS
  const float cutoff = accept;
  {
    int n = (int)((cnt_e >> 16) & 0xFFFFull);
    for (int round = 1; n > 0; round++) {
      if (round > 100000) { sh.err = K3_ERR_HIP; break; }
      const int cur = round & 1; const unsigned short *wl_cur = wl + cur * kFW; unsigned short *wl_nxt = wl + (cur ^ 1) * kFW; int *n_nxt = &fs.n_wl[(round + 1) % 3];
      auto fail = [&](int why) { abort_now(why); fs.abort_r[round & 3] = 1; };
      for (int i0 = 0; i0 < n; i0 += kBlock) {
        const int i = i0 + tid; int ti = 0, beg = 0, deg = 0; unsigned cb = 0u;
        if (i < n) {
          ti = wl_cur[i]; cb = lds_ld(&N_cost[ti]);
          // a token is expanded once per cost value
          if (dec(cb) < cutoff) {
            const unsigned prev = k3a_exch(&X[ti], cb);
            if (prev != cb) {
              beg = (int)(N_abeg[ti] + N_ne[ti]);
              deg = (int)N_nn[ti];
            }
          }
        }
        wave_expand_d(p.arcs, p.dinfo, beg, deg, [&](bool valid, int arc, int owner, const ArcRec &r, const int2 &di) {
          const unsigned ocb = __shfl(cb, owner); const int oti = __shfl(ti, owner); const float oc = dec(ocb);
          bool claimed = false, push = false, mk = false; int slot = -1, nxt = 0; float tot = 0.0f;
          cnt_eps += valid;
          if (valid) {
            tot = oc + r.w; nxt = (int)((unsigned)r.next & ~kEpsFlag);
            if (tot < cutoff) {
              unsigned h = hash_state(nxt) & (kFH - 1);
              for (int probe = 0; probe < kFProbe; probe++) {
                int k = lds_ld(&T_key[h]); bool cl = false;
                if (k == kEmpty) { const int old = k3a_cas(&T_key[h], kEmpty, nxt); if (old == kEmpty) { cl = true; k = nxt; } else k = old; }
                if (k == nxt) { slot = (int)h; claimed = cl; break; }
                h = (h + 1) & (kFH - 1);
              }
              if (slot < 0) fail(kFaTable); else mk = true;
            }
          }
          int idx, unused_; long long lrel;
          wave_append3(claimed, false, mk, &fs.cnt, idx, unused_, lrel);
          if (claimed) {
            if (idx >= cap_tokens || nb + idx >= c.tcap) { fail(kFaTokens); idx = 0; }
            else c.tok_state[nb + idx] = nxt;
            __hip_atomic_store(&T_tix[slot], (unsigned short)idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if (mk && !claimed) {
            for (int spin = 0;; spin++) {
              const unsigned short t = lds_ld16(&T_tix[slot]);
              if (t != 0xFFFFu) {
                idx = t;
                break;
              }
              if (spin > (1 << 22)) {
                sh.err = K3_ERR_HIP;
                idx = 0;
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          if (mk) {
            const unsigned e = enc(tot); const unsigned old = k3a_min(&N_cost[idx], e);
            // only tokens whose state has eps arcs are queued, once per round
            if (e < old && r.next < 0) {
              const unsigned bit = 1u << (idx & 31);
              push = (k3a_or(&marks[((round + 1) % 3) * (kFT / 32) + (idx >> 5)], bit) & bit) == 0;
            }
          }
          const int pos = wave_append(push, n_nxt);
          if (push) { if (pos < kFW) wl_nxt[pos] = (unsigned short)idx; else fail(kFaWl); }
          const long long lp = link0 + lrel;
          if (mk) {
            const long long el = lp - eps_l0;
            if (lp < c.lcap && el < kFE) {
              store_link(&c.links[lp], Link{(unsigned)(nb + oti), (unsigned)(nb + idx), tot, __uint_as_float(ocb)}); store_stream(&c.link_arc[lp], arc);
              E_src[el] = (unsigned short)oti; E_dst[el] = (unsigned short)idx; E_arc[el] = (unsigned)arc; E_stamp[el] = ocb; E_w[el] = r.w;
            } else fail(lp < c.lcap ? kFaLinks : kFaPool);
          }
          if (claimed) {
            const unsigned ne = (unsigned)di.y & 0xFFFFu, nn = (unsigned)di.y >> 16;
            if (ne == 0xFFFFu || nn == 0xFFFFu) fail(kFaDegree);
            N_abeg[idx] = (unsigned)di.x; N_ne[idx] = (unsigned short)ne; N_nn[idx] = (unsigned short)nn;
          }
        });
      }
      if (tid == 0) { fs.n_wl[(round + 2) % 3] = 0; fs.abort_r[(round + 2) & 3] = 0; }
      for (int i = tid; i < kFT / 32; i += kBlock) marks[((round + 2) % 3) * (kFT / 32) + i] = 0u;
      __syncthreads();
      if (fs.abort_r[round & 3]) break;
      n = fs.n_wl[(round + 1) % 3];
    }
  }
  { __syncthreads(); const int bad = sh.err | fs.abort; __syncthreads(); if (bad) return -1; }      // (the lane's error flag and the give-up flag in one snapshot)
  const unsigned long long cnt_n = fs.cnt;      // (uniform: nothing adds to it any more)
  const int n = (int)(cnt_n & 0xFFFFull); const long long n_link_end = link0 + (long long)(cnt_n >> 32); const int n_el = (int)(n_link_end - eps_l0);
  K3_FP(4);
  // ---- the frame's final costs into the pool; buckets of the HashList (the table is dead afterwards)
  for (int i = tid; i < n; i += kBlock) c.tok_cost[nb + i] = N_cost[i];
  for (int s_ = tid; s_ < kFH; s_ += kBlock) { const int k = T_key[s_]; if (k != kEmpty) B16[T_tix[s_]] = (unsigned short)((unsigned)k % hash_size); }
  __syncthreads();
  // ---- closure sub-graph from the live links (made at their source's FINAL cost), in closure-id space
  unsigned *rown = reinterpret_cast<unsigned *>(arena + oRown), *rflag = reinterpret_cast<unsigned *>(arena + oRflag), *srcbit = reinterpret_cast<unsigned *>(arena + oSrcbit);
  unsigned short *cid = reinterpret_cast<unsigned short *>(arena + oCid), *lead = reinterpret_cast<unsigned short *>(arena + oLead);
  unsigned *AR_arc = reinterpret_cast<unsigned *>(arena + oAR_arc);
  unsigned short *AR_dst = reinterpret_cast<unsigned short *>(arena + oAR_dst);
  float *AR_w = reinterpret_cast<float *>(arena + oAR_w);
  unsigned short *m_abeg = reinterpret_cast<unsigned short *>(arena + oM_abeg), *m_pc = reinterpret_cast<unsigned short *>(arena + oM_pc),
      *c2t = reinterpret_cast<unsigned short *>(arena + oC2t);
  float *rcost = reinterpret_cast<float *>(arena + oRcost);
  for (int i = tid; i < n; i += kBlock) rown[i] = 0u;
  for (int i = tid; i < kFT / 32; i += kBlock) { rflag[i] = 0u; srcbit[i] = 0u; }
  __syncthreads();
  for (int l = tid; l < n_el; l += kBlock) {
    const int s_ = E_src[l];
    if (E_stamp[l] == N_cost[s_]) { const int d = E_dst[l]; k3a_add(&rown[s_], 1u); k3a_or(&rflag[d >> 5], 1u << (d & 31)); }
  }
  __syncthreads();
  int4 *red4 = reinterpret_cast<int4 *>(sh.hist);
  const int4 tot2 = block_excl_scan4([&](int i) { const int pc = (int)rown[i]; return make_int4((pc > 0 || (rflag[i >> 5] >> (i & 31) & 1u)) ? 1 : 0, pc, 0, 0); },
                                     [&](int i, int4 ex) {
                                       const int pc = (int)rown[i];
                                       cid[i] = (unsigned short)ex.x; lead[i] = (unsigned short)ex.y;
                                       if ((pc > 0 || (rflag[i >> 5] >> (i & 31) & 1u)) && ex.x < kFC) {
                                         m_abeg[ex.x] = (unsigned short)ex.y;
                                         m_pc[ex.x] = (unsigned short)pc;
                                         c2t[ex.x] = (unsigned short)i;
                                       }
                                       if (pc > 0) k3a_or(&srcbit[i >> 5], 1u << (i & 31));
                                     }, n, red4);
  const int n_cid = tot2.x, n_arc = tot2.y;
  if (n_cid > kFC || n_arc > kFA) { if (tid == 0) fs.reason = kFaClosure; return -1; }      // (uniform)
  for (int l = tid; l < n_el; l += kBlock) {
    const int s_ = E_src[l];
    if (E_stamp[l] == N_cost[s_]) {
      const int pos = (int)lead[s_] + (int)k3a_add(&rown[s_], 0xFFFFFFFFu) - 1;
      AR_arc[pos] = E_arc[l];
      AR_dst[pos] = cid[E_dst[l]];
      AR_w[pos] = E_w[l];
    }
  }
  __syncthreads();
  unsigned *par = reinterpret_cast<unsigned *>(arena + oPar);
  unsigned short *croots = reinterpret_cast<unsigned short *>(arena + oCroots), *ccreated = reinterpret_cast<unsigned short *>(arena + oCcreated),
      *carcs = reinterpret_cast<unsigned short *>(arena + oCarcs),
                 *ccurs = reinterpret_cast<unsigned short *>(arena + oCcurs);
  // a source's passing arcs in FST order (ascending arc index); the replay's starting costs (the costs right after ProcessEmitting; +inf: not created yet)
  for (int cc = tid; cc < n_cid; cc += kBlock) {
    const int abeg = m_abeg[cc], pc = m_pc[cc];
    for (int a = abeg + 1; a < abeg + pc; a++) {
      const unsigned ka = AR_arc[a]; const unsigned short kd = AR_dst[a]; const float kw = AR_w[a]; int b_ = a - 1;
      while (b_ >= abeg && AR_arc[b_] > ka) { AR_arc[b_ + 1] = AR_arc[b_]; AR_dst[b_ + 1] = AR_dst[b_]; AR_w[b_ + 1] = AR_w[b_]; b_--; }
      AR_arc[b_ + 1] = ka; AR_dst[b_ + 1] = kd; AR_w[b_ + 1] = kw;
    }
  }
  __syncthreads();      // (rcost lies over E_arc and the order scratch below over E_stamp / E_w: the links have been consumed)
  for (int cc = tid; cc < n_cid; cc += kBlock) { const int i = c2t[cc]; rcost[cc] = i < n_e ? c0[i] : kInf; }
  K3_FP(5);
  // ---- the list ProcessNonemitting fills its queue from (:845-850): HashList order of the tokens ProcessEmitting made
  unsigned short *ord1 = reinterpret_cast<unsigned short *>(arena + oOrd1);
  // (the emitting tokens' labels -- arc sequence numbers -- become their dense creation ranks 0 .. n_e-1: the pass has every label in registers by now, and
  // the frame's second pass then needs no ranking of its own)
  fast_hash_order<false>(sh, n_e, m_e, lab16, B16, reinterpret_cast<unsigned *>(arena + oO1_btab), reinterpret_cast<unsigned *>(arena + oO1_bm),
      reinterpret_cast<unsigned short *>(arena + oO1_wpre),
                  reinterpret_cast<unsigned short *>(arena + oO1_lead), reinterpret_cast<unsigned short *>(arena + oO1_grp), reinterpret_cast<unsigned short *>(arena + oO1_curs),
                  [&](int r, int i, int d) { ord1[r] = (unsigned short)i; lab16[i] = (unsigned short)d; });
  K3_FP(6);
  // the initial queue: the tokens of that list that can expand (closure ids), consumed from its back
  unsigned short *iq = reinterpret_cast<unsigned short *>(arena + oIq), *dense = reinterpret_cast<unsigned short *>(arena + oDense);
  // (the component replay's cells -- own parent, zero counters -- are set here: the scan's barriers put them in front of the union phase, no barrier of their own)
  for (int cc = tid; cc < n_cid; cc += kBlock) par[cc] = (unsigned)cc;
  for (int i = tid; i < n_cid / 2 + 1; i += kBlock) {
    reinterpret_cast<unsigned *>(croots)[i] = 0u;
    reinterpret_cast<unsigned *>(ccreated)[i] = 0u;
    reinterpret_cast<unsigned *>(carcs)[i] = 0u;
    reinterpret_cast<unsigned *>(ccurs)[i] = 0u;
  }
  const int n_iq = block_excl_scan_f([&](int r) { const int i = ord1[r]; return (int)(srcbit[i >> 5] >> (i & 31) & 1u); },
                                     [&](int r, int ex) { const int i = ord1[r]; if ((srcbit[i >> 5] >> (i & 31) & 1u) && ex < kFQ) iq[ex] = cid[i]; }, n_e, sh.redi);
  if (n_iq > kFQ) { if (tid == 0) fs.reason = kFaQueue; return -1; }
  // ---- replay of the LIFO queue by connected components (oracle mode 4; lit_replay_components with 16-bit LDS records)
  unsigned short *ox = reinterpret_cast<unsigned short *>(arena + oOx), *oy = reinterpret_cast<unsigned short *>(arena + oOy),
      *oz = reinterpret_cast<unsigned short *>(arena + oOz), *ow = reinterpret_cast<unsigned short *>(arena + oOw);
  unsigned short *rlist = reinterpret_cast<unsigned short *>(arena + oRlist), *rinfo = reinterpret_cast<unsigned short *>(arena + oRinfo),
      *rtmp = reinterpret_cast<unsigned short *>(arena + oRtmp);
  unsigned short *wrec = reinterpret_cast<unsigned short *>(arena + oWrec), *stack = reinterpret_cast<unsigned short *>(arena + oStack),
      *clist = reinterpret_cast<unsigned short *>(arena + oClist);
  auto find = [&](int x) { for (;;) { const int q_ = (int)lds_ld(&par[x]); if (q_ == x) return x; x = q_; } };
  for (int cc = tid; cc < n_cid; cc += kBlock) {
    const int abeg = m_abeg[cc], pc = m_pc[cc];
    for (int a = 0; a < pc; a++) {      // lock-free union: the larger root hooks under the smaller one
      int x = cc, y = AR_dst[abeg + a];
      for (;;) { x = find(x); y = find(y); if (x == y) break; if (x < y) { const int t = x; x = y; y = t; } if (k3a_cas(&par[x], (unsigned)x, (unsigned)y) == (unsigned)x) break; }
    }
  }
  __syncthreads();
  K3_FP(10);
  for (int cc = tid; cc < n_cid; cc += kBlock) {
    const int r = find(cc); if (r != cc) __hip_atomic_store(&par[cc], (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (rcost[cc] == kInf) add16(ccreated, r, 1u);
    const int pc = m_pc[cc]; if (pc > 0) add16(carcs, r, (unsigned)pc);
  }
  for (int k = tid; k < n_iq; k += kBlock) add16(croots, find(iq[k]), 1u);
  __syncthreads();
  const int4 tot = block_excl_scan4([&](int cc) { const int rr = croots[cc]; return make_int4(rr, (int)ccreated[cc], (int)carcs[cc], rr > 0 ? 1 : 0); },
                                    [&](int cc, int4 ex) {
                                      ox[cc] = (unsigned short)ex.x;
                                      oy[cc] = (unsigned short)ex.y;
                                      oz[cc] = (unsigned short)ex.z;
                                      ow[cc] = (unsigned short)ex.w;
                                    },
                                    n_cid, red4);
  const int n_workers = tot.w;
  bool multi = false;
  for (int k = tid; k < n_iq; k += kBlock) {      // a component's roots in queue order (the queue is consumed from its back: descending k)
    const int e = iq[k]; const int r = (int)par[e]; const int nr = croots[r];
    if (nr == 1) {
      rlist[2 * ox[r]] = (unsigned short)k;
      rlist[2 * ox[r] + 1] = (unsigned short)e;
      unsigned short *w = wrec + 5 * ow[r];
      w[0] = ox[r];
      w[1] = 1;
      w[2] = oy[r];
      w[3] = oz[r];
      w[4] = carcs[r];
    }
    else { multi = true; const unsigned pos = add16(ccurs, r, 1u); rtmp[ox[r] + pos] = (unsigned short)k; }
  }
  multi = __syncthreads_or(multi);
  if (multi) {
    for (int k = tid; k < n_iq; k += kBlock) {
      const int e = iq[k]; const int r = (int)par[e]; const int nr = croots[r];
      if (nr > 1) {
        int rank = 0;
        for (int t = 0; t < nr; t++) rank += (int)rtmp[ox[r] + t] > k;
        rlist[2 * (ox[r] + rank)] = (unsigned short)k; rlist[2 * (ox[r] + rank) + 1] = (unsigned short)e;
        if (rank == 0) { unsigned short *w = wrec + 5 * ow[r]; w[0] = ox[r]; w[1] = (unsigned short)nr; w[2] = oy[r]; w[3] = oz[r]; w[4] = carcs[r]; }
      }
    }
    __syncthreads();
  }
  K3_FP(11);
  for (int w_ = tid; w_ < n_workers; w_ += kBlock) {
    const unsigned short *w = wrec + 5 * w_;
    if (!fast_replay_component(rcost, m_abeg, m_pc, AR_dst, AR_w, clist, rlist, rinfo, stack + w[3], (int)w[4], (int)w[0], (int)w[1], (int)w[2], accept)) abort_now(kFaStack);
  }
  if (aborted()) return -1;
  K3_FP(7);
  // creation labels: roots in queue order (j-th root processed = position n_iq - 1 - j), tokens of a root in the order it created them
  const int created = block_excl_scan_f([&](int j) { return (int)rinfo[2 * (n_iq - 1 - j) + 1]; }, [&](int j, int ex) { dense[j] = (unsigned short)ex; }, n_iq, sh.redi);
  // every token of the fixpoint must have been created by the replay (the general path re-checks)
  if (n_e + created != n) {
    if (tid == 0) fs.reason = kFaMismatch;
    return -1;
  }
  for (int j = tid; j < n_iq; j += kBlock) {
    const int seg0 = rinfo[2 * (n_iq - 1 - j)], cnt = rinfo[2 * (n_iq - 1 - j) + 1]; const unsigned base = (unsigned)n_e + (unsigned)dense[j];      // (dense ranks: behind the emitting tokens')
    for (int t = 0; t < cnt; t++) lab16[c2t[clist[seg0 + t]]] = (unsigned short)(base + (unsigned)t);
  }
  __syncthreads();
  K3_FP(8);
  // ---- the frame's final HashList order = the next frame's visit order, written where the next frame reads it; creation order for the final-frame sweeps
  fast_hash_order<true>(sh, n, (unsigned)n, lab16, B16, reinterpret_cast<unsigned *>(arena + oO2_btab), reinterpret_cast<unsigned *>(arena + oO2_bm),
      reinterpret_cast<unsigned short *>(arena + oO2_wpre),
                  reinterpret_cast<unsigned short *>(arena + oO2_lead), reinterpret_cast<unsigned short *>(arena + oO2_grp), reinterpret_cast<unsigned short *>(arena + oO2_curs),
                  [&](int r, int i, int d) {
                    V_cost[r] = N_cost[i]; V_abeg[r] = N_abeg[i]; V_ne[r] = N_ne[i]; V_tok[r] = (unsigned short)i; ord_nxt[r] = i; q.by_ins[d] = i;
#if K3_LIT_PREFETCH_ARCS
                    if (N_ne[i] > 0) arc_pf += p.arcs[N_abeg[i]].pdf;      // (the next frame's pass A finds the token's first arcs in the L2)
#endif
                  });
#if K3_LIT_PREFETCH_ARCS
  if (arc_pf == 0x7FFFFFF1) fs.reason = -8;      // (never true: keeps the loads alive; looked at behind the phase's barrier)
#endif
  K3_FP(9);
  if (tid == 0) { c.tok_off[f + 2] = nb + n; c.loff_e[f + 1] = n_link_end; sh.n_link = n_link_end; sh.n_next = n; }
  hash_size_io = hash_size; cnt_emit_io += cnt_emit; cnt_os_io += cnt_os; cnt_eps_io += cnt_eps;
  return n;
}
