// k3_decoder.hip -- batched HCLG lattice decoder for gfx950 (MI355X): wavefront-scan + LDS-atomic token passing
// over a CSR graph resident in HBM, one persistent workgroup per lane (utterance).
//
// What it computes (paths relative to the reference's src/): LatticeFasterDecoder::Decode + GetRawLattice
// (decoder/lattice-faster-decoder.cc:63-197,588-649), i.e. per frame GetCutoff (:653-720), ProcessEmitting (:723-814),
// ProcessNonemitting (:830-897), then FinalizeDecoding = PruneForwardLinksFinal (:385-467) + per-frame
// PruneForwardLinks (:308-379) / PruneTokensForFrame (:488-507), with float32 costs formed in the reference's
// evaluation order ((cur + (cost_offset - loglike)) + graph; eps arcs cur + graph; link_extra = next.extra +
// ((tot + ac + graph) - next.tot)).  The beam is applied against the FINAL next_cutoff of the frame (the serial code
// tightens it while it walks its hash list, which makes its exact arc set depend on hash order; DESIGN.md
// "decoder parity") and ties for the best token go to the smaller state id.
//
// Design (not the CUDA reference's: that one launches ~22 kernels per frame over arc-level tokens, prunes with a
// histogram and builds the lattice on the host):
//  * one workgroup per lane keeps the whole 333-step frame recurrence inside ONE kernel launch: no host sync, no
//    inter-workgroup communication; 2 x 256 CUs' worth of lanes fill the chip, lanes of different length just retire
//    at different times.  Counters, cutoffs, radix-select histograms and the frame's log-likelihood row live in LDS.
//  * tokens are (frame, state) with the minimum cost kept by atomicMin on an order-preserving integer image of
//    the float in a per-lane open-addressing table {state, cost, token, stamp} (16 B slots, one cache line per probe);
//  * arcs are expanded wave-cooperatively: 64 tokens per wavefront, a wave scan of their out-degrees, then every
//    lane takes one arc per step (binary search over the scan through ds_bpermute), so a 4000-arc loop state and a
//    1-arc chain state cost the same per arc; queue appends are ballot/popcount aggregated (one LDS atomic per wave);
//  * max_active / min_active use an exact radix select (std::nth_element's k-th value), not a histogram estimate;
//  * every accepted arc leaves a 16 B forward link {src, dst, arc, acoustic}; lattice-beam pruning runs on the GPU
//    after the last frame (one workgroup per lane, frames in reverse, epsilon links iterated to the exact fixpoint)
//    and only the surviving lattice is compacted and copied to the host.
// HBM layout: graph = {int2 offsets[S+1] (first arc, first eps arc), 16 B arcs {next, weight, pdf, olabel} with the
// emitting arcs of a state before its eps arcs, finals[S], arc ilabels[A]} in ONE allocation (broadcastable);
// per lane: token pool (state, cost, extra) for all frames, link pool, per-frame offsets and cutoffs.

#include "k3_decoder_dev.h"

namespace {

#ifndef K3_DEC_WPE
#define K3_DEC_WPE 4
#endif
__global__ __launch_bounds__(kBlock, K3_DEC_WPE) void k3_decode_forward_kernel(DecParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ Shared sh;
  __shared__ int s_lkey[kHL]; __shared__ unsigned s_lcost[kHL]; __shared__ int s_ltok[kHL]; __shared__ unsigned s_lmark[3 * (kHL / 32)];
  __shared__ unsigned short s_lwl[2][kWlLds];
  float *s_ll = reinterpret_cast<float *>(smem_raw);
  const int L = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const long long r0 = p.row_off[L]; const int T = (int)(p.row_off[L + 1] - r0);
  __shared__ LanePool s_pool;
  LanePool lp = k3_uniform_pool(p.pools[L]);
  int *tok_state = lp.tok_state; unsigned *tok_cost = lp.tok_cost; Link *links = lp.links; int *link_arc = lp.link_arc;
  Slot *hash = p.hash + (long long)L * (p.hash_mask + 1);
  int *tok_slot = p.tok_slot + (long long)L * p.frame_tokens_cap, *wl = p.wl + 2ll * L * p.frame_tokens_cap;
  float *c_tot = p.c_tot + (long long)L * p.frame_cands_cap, *c_ac = p.c_ac + (long long)L * p.frame_cands_cap;
  int *c_dst = p.c_dst + (long long)L * p.frame_cands_cap, *c_arc = p.c_arc + (long long)L * p.frame_cands_cap, *c_src = p.c_src + (long long)L * p.frame_cands_cap;
  long long *tok_off = p.tok_off + L * p.fstride, *loff_e = p.link_off_e + L * p.fstride, *loff_n = p.link_off_n + L * p.fstride;
  int *st_ntoks = p.st_ntoks + L * p.fstride; float *st_cur = p.st_cur + L * p.fstride, *st_ab = p.st_ab + L * p.fstride,
        *st_next = p.st_next + L * p.fstride, *st_co = p.st_co + L * p.fstride;
  const unsigned mask = (unsigned)p.hash_mask;
  const float kInf = __builtin_inff();

  if (tid < 16) sh.prof[tid] = 0;
  if (tid == 0) { sh.n_next = 0; sh.n_cand = 0; sh.err = 0; sh.n_link = 0; sh.min_tot = kEncMax; sh.flag = 0; sh.n_eps = 0; sh.n_emit = 0; sh.n_os = 0; }
  for (int i = tid; i < kHL; i += kBlock) { s_lkey[i] = kEmpty; s_lcost[i] = kEncMax; s_ltok[i] = -1; }
  for (int i = tid; i < 3 * (kHL / 32); i += kBlock) s_lmark[i] = 0;
  const Table tb{s_lkey, s_lcost, s_ltok, s_lmark, hash, mask};
  __syncthreads();
  long long t_last__ = (long long)__builtin_readcyclecounter();
  unsigned cnt_eps = 0, cnt_emit = 0, cnt_os = 0;      // per-thread arc counters (reduced once at the end of the kernel)
  long long cur_base = 0; int n_cur = 0, max_frame = 0, f0 = 0, status = kStOk;
  unsigned creg[kCurRegs]; int sreg[kCurRegs]; bool in_regs = false;      // (cost, state) of current-frame token tid + k * kBlock
#pragma unroll
  for (int k = 0; k < kCurRegs; k++) { creg[k] = kEncMax; sreg[k] = 0; }
  const bool fresh = p.fresh[L] != 0;
  if (!fresh && T == 0) return;        // the lane idles in this call (it may already be finalised: its LaneInfo must stay as it is)
  if (fresh) {
    // The frame loop leaves through `break` on an error (capacity overflow, spin limit) without running the end of finish_frame, which is
    // what empties the level-2 (HBM) slots of the frame: a lane restarted after a failed utterance wipes its table first.
    if (p.info[L].status < 0) {
      for (unsigned i = tid; i <= mask; i += kBlock) { Slot *q = &hash[i]; K3_AST(&q->cost, kEncMax); K3_AST(&q->stamp, 0); K3_AST(&q->tok, -1); K3_AST(&q->key, kEmpty); }
      __threadfence(); __syncthreads();
    }
    // ---- InitDecoding (:63-81): start token, eps closure with cutoff = beam
    if (tid == 0) {
      bool cl; const int slot = tb.claim(p.start, &cl);
      // marker stores are agent-scope atomics: the expander's atomicExch (at L2) must not overtake them
      tb.cost_min(slot, enc(0.0f));
      tb.set_tok(slot, 0);
      tok_slot[0] = slot;
      tok_state[0] = p.start;
      K3_AST(&tok_cost[0], kEncMax);
      sh.n_next = 1;
      sh.n_wl[0] = 0; sh.n_wl[2] = 0; sh.n_wl[1] = 1; for (int i = 0; i < 4; i++) sh.err_r[i] = 0;      // round-1 list = the start token
      if (slot < kHL) s_lwl[1][0] = (unsigned short)slot; else { s_lwl[1][0] = 0xFFFF; wl[p.frame_tokens_cap] = slot; }
      tok_off[0] = 0; loff_n[0] = 0;
    }
    __syncthreads();
  } else {
    // ---- AdvanceDecoding on a later chunk: pick the lane up where the previous launch left it
    const LaneInfo &li = p.info[L];
    if (li.status != kStOk) return;
    f0 = li.num_frames; cur_base = li.cur_base; n_cur = li.n_cur; max_frame = li.max_frame_tokens;
    if (tid == 0) {
      sh.n_link = li.n_links;
      sh.n_eps = (unsigned long long)li.n_eps;
      sh.n_emit = (unsigned long long)li.n_cands;
      sh.n_os = (unsigned long long)li.n_order_sensitive;
    }
    __syncthreads();
  }

#ifdef K3_DEC_PROF
  long long t_frame__ = (long long)__builtin_readcyclecounter();
#endif
  // frame -1 (only on a fresh start) is InitDecoding's eps closure of the start token with cutoff = beam: it shares the
  // ProcessNonemitting code of the real frames (one copy of it in the instruction stream)
  for (int f = fresh ? -1 : f0; f < f0 + T; f++) {
    if (block_err(sh)) break;
#ifdef K3_DEC_PROF
    if (tid == 0) sh.prof_n = n_cur;
#endif
    float accept = p.beam; long long nb = 0;
    if (f >= 0) {
    // room for a whole frame behind what the lane holds (a frame makes at most frame_tokens_cap tokens, frame_cands_cap forward links and -- a bound the closure's
    // overflow check enforces -- as many epsilon links): a lane that has outgrown its reservation moves to bigger pools here, where nothing of the new frame exists yet
    { const long long nl_ = sh.n_link;
      if (lp.tcap - (cur_base + n_cur) < p.frame_tokens_cap || lp.lcap - nl_ < 2ll * p.frame_cands_cap) {
        if (!grow_lane_pools(p, L, lp, cur_base + n_cur, nl_, p.frame_tokens_cap, 2ll * p.frame_cands_cap, &s_pool)) {
          if (tid == 0) sh.err = K3_ERR_OVERFLOW;
          __syncthreads();
          break;
        }
        tok_state = lp.tok_state; tok_cost = lp.tok_cost; links = lp.links; link_arc = lp.link_arc;
      } }
    // stage the log-likelihood row of this frame in LDS (coalesced), overlapped with the cutoff passes
    K3_T(0);
    const float *row = p.lane_rows ? p.lane_rows[L] + (long long)(f - f0) * p.ld : p.loglikes + (r0 + (f - f0)) * p.ld;
    if (p.use_lds_row && f == f0) for (int i = tid; i < p.num_pdfs; i += kBlock) s_ll[i] = row[i];   // later rows are prefetched one frame ahead
    const float *ll = p.use_lds_row ? s_ll : row;
    const int *cst = tok_state + cur_base; const unsigned *ccs = tok_cost + cur_base;
    if (n_cur == 0) { status = kStNoTokens; break; }
    // ---- GetCutoff (:653-720)
    // fn(i, cost, state) over this thread's share of the current frame: from registers when the previous frame left them there
    auto for_cur = [&](auto fn) {
      if (in_regs) {
#pragma unroll
        for (int k = 0; k < kCurRegs; k++) { const int i = tid + k * kBlock; if (i < n_cur) fn(i, creg[k], sreg[k]); }
      } else {
        for (int i0 = tid; i0 < n_cur; i0 += 4 * kBlock) {
          unsigned c[4]; int st[4];
#pragma unroll
          for (int j = 0; j < 4; j++) { const int i = i0 + j * kBlock; c[j] = i < n_cur ? ccs[i] : 0u; st[j] = i < n_cur ? cst[i] : 0; }
#pragma unroll
          for (int j = 0; j < 4; j++) { const int i = i0 + j * kBlock; if (i < n_cur) fn(i, c[j], st[j]); }
        }
      }
    };
    auto for_keys = [&](auto fn) {       // wave-uniform trip count (block_select_kth)
      if (in_regs) {
#pragma unroll
        for (int k = 0; k < kCurRegs; k++) { if (k * kBlock < n_cur) fn(tid + k * kBlock < n_cur, creg[k]); }
      } else for (int i0 = 0; i0 < n_cur; i0 += kBlock) { const int i = i0 + tid; fn(i < n_cur, i < n_cur ? ccs[i] : 0u); }
    };
    unsigned long long bm = ~0ull;
    for_cur([&](int, unsigned c, int st) { const unsigned long long v = ((unsigned long long)c << 32) | (unsigned)st; bm = v < bm ? v : bm; });
    K3_T(1);
    bm = block_min_u64(bm, sh);
    K3_T(1);
    const float best = dec((unsigned)(bm >> 32)); const int best_state = (int)(unsigned)(bm & 0xFFFFFFFFull);
    float cur_cutoff, ab;
    const float beam_cutoff = best + p.beam;
    if (p.max_active == 0x7FFFFFFF && p.min_active == 0) { ab = p.beam; cur_cutoff = beam_cutoff; }
    else {
      const unsigned ebc = enc(beam_cutoff);
      int c_lt = 0, c_le = 0;
      for_cur([&](int, unsigned k, int) { c_lt += k < ebc; c_le += k <= ebc; });
      K3_T(1);
      c_lt = block_sum_i32(c_lt, sh); c_le = block_sum_i32(c_le, sh);
      K3_T(1);
      // tmp[max_active] < beam_cutoff  <=>  more than max_active elements are < beam_cutoff
      int kth = -1;
      if (n_cur > p.max_active && c_lt > p.max_active) kth = p.max_active;                                   // max_active_cutoff = tmp[max_active] < beam_cutoff
      else if (n_cur > p.min_active && p.min_active == 0) { ab = p.beam; cur_cutoff = beam_cutoff; }         // min_active_cutoff = best <= beam_cutoff
      else if (n_cur > p.min_active && c_le > p.min_active) { ab = p.beam; cur_cutoff = beam_cutoff; }      // tmp[min_active] <= beam_cutoff
      else if (n_cur > p.min_active) kth = p.min_active;                                                     // min_active_cutoff = tmp[min_active] > beam_cutoff
      else { ab = kInf - best + p.beam_delta; cur_cutoff = kInf; }                                           // min_active_cutoff = +inf
      if (kth >= 0) { const float sel = dec(block_select_kth(for_keys, kth, sh)); ab = sel - best + p.beam_delta; cur_cutoff = sel; }
    }
    K3_T(1);
    const float co = -best;
    // ---- pre-pass over the best token's emitting arcs (:753-768; note its own evaluation order)
    __syncthreads();
    unsigned n0 = kEncMax;
    {
      const int2 a = p.offs[best_state];
      for (int arc = a.x + tid; arc < a.y; arc += kBlock) {
        const ArcRec r = p.arcs[arc];
        const float nw = r.w + co - ll[r.pdf] + best;
        const unsigned e = enc(nw + ab); n0 = e < n0 ? e : n0;
      }
    }
    n0 = (unsigned)(block_min_u64((unsigned long long)n0, sh) & 0xFFFFFFFFull);
    const float next0 = n0 == kEncMax ? kInf : dec(n0);
    K3_T(2);
    if (tid == 0) { sh.n_cand = 0; sh.min_tot = kEncMax; sh.n_next = 0; }
    if (tid < 3) sh.n_wl[tid] = 0;
    if (tid < 4) sh.err_r[tid] = 0;
    for (int i = tid; i < 3 * (kHL / 32); i += kBlock) s_lmark[i] = 0;
    __syncthreads();
    // ---- ProcessEmitting pass 1 (:779-797): every emitting arc of every token <= cur_cutoff; keep tot < pre-pass bound
    auto expand_tok = [&](int t, float c, int beg, int deg) {
      wave_expand(p.arcs, beg, deg, [&](bool valid, int arc, int owner, const ArcRec &r) {
        const float oc = __shfl(c, owner); const int ot = __shfl(t, owner);
        bool pass = false; float tot = 0.0f, ac = 0.0f; int nxt = 0;
        cnt_emit += valid;
        if (valid) {
          ac = co - ll[r.pdf]; tot = oc + ac + r.w; nxt = r.next;
          pass = tot < next0;
        }
        const unsigned wm = wave_min_u32(pass ? enc(tot) : kEncMax);
        if (lane == 0 && wm != kEncMax) k3a_min(&sh.min_tot, wm);
        const int pos = wave_append(pass, &sh.n_cand);
        if (pass) {
          if (pos < p.frame_cands_cap) { c_tot[pos] = tot; c_ac[pos] = ac; c_dst[pos] = nxt; c_arc[pos] = arc; c_src[pos] = ot; }
          else sh.err = K3_ERR_OVERFLOW;
        }
      });
    };
    for (int t0 = 0, k = 0; t0 < n_cur; t0 += kBlock, k++) {
      const int t = t0 + tid; const bool v = t < n_cur;
      unsigned cb = kEncMax; int st = 0;
      if (in_regs) { cb = k == 0 ? creg[0] : k == 1 ? creg[1] : k == 2 ? creg[2] : creg[3]; st = k == 0 ? sreg[0] : k == 1 ? sreg[1] : k == 2 ? sreg[2] : sreg[3]; }
      else if (v) { cb = ccs[t]; st = cst[t]; }
      int beg = 0, deg = 0; const float c = dec(cb);
      if (v && c <= cur_cutoff) { const int2 a = p.offs[st]; beg = a.x; deg = a.y - a.x; }
      expand_tok(t, c, beg, deg);
    }
    if (tid == 0) loff_e[f] = sh.n_link;
    if (block_err(sh)) break;
    K3_T(3);
    // the LDS row is dead from here on: fetch the next frame's row into registers now (the loads fly during pass 2),
    // park it in LDS after pass 2
    float rowreg[kRowRegs];
    const bool prefetch = K3_DEC_LDSROW && p.use_lds_row && f + 1 < f0 + T;
    auto fetch_row = [&]() {
#pragma unroll
      for (int k = 0; k < kRowRegs; k++) { const int i = tid + k * kBlock; rowreg[k] = i < p.num_pdfs ? row[p.ld + i] : 0.0f; }
    };
    bool row_pending = prefetch;      // issued behind the first candidate loads: vector loads return in order, the candidates must not queue behind the row
    // ---- final bound of the frame, pass 2: tokens (min cost per state) for the accepted arcs
    accept = next0;
    { const unsigned mt = sh.min_tot; if (mt != kEncMax) { const float t = dec(mt) + ab; if (t < accept) accept = t; } }
    const int n_cand = sh.n_cand;
    nb = cur_base + n_cur;
    // candidate records are read one block ahead of their use
    float q_tot = 0.0f, q_c = 0.0f; int q_nxt = 0, q_a = 0, q_s = 0;
    if (tid < n_cand) { q_tot = c_tot[tid]; q_nxt = c_dst[tid]; q_a = c_arc[tid]; q_s = c_src[tid]; q_c = c_ac[tid]; }
    for (int j0 = 0; j0 < n_cand; j0 += kBlock) {
      const int j = j0 + tid; bool claimed = false, mk = false; int slot = -1;
      const float tot = q_tot, c_c = q_c; const int nxt = q_nxt, c_a = q_a, c_s = q_s;
      const int state = (int)((unsigned)nxt & ~kEpsFlag);
      { const int jn = j + kBlock; if (jn < n_cand) { q_tot = c_tot[jn]; q_nxt = c_dst[jn]; q_a = c_arc[jn]; q_s = c_src[jn]; q_c = c_ac[jn]; } }
      if (row_pending) { fetch_row(); row_pending = false; }
      if (j < n_cand) {
        cnt_os += !(tot < accept);
        if (tot < accept) {
          slot = tb.claim(state, &claimed);
          if (slot < 0) { sh.err = K3_ERR_OVERFLOW; claimed = false; } else { tb.cost_min(slot, enc(tot)); mk = true; }
        }
      }
      int idx = wave_append(claimed, &sh.n_next);
      if (claimed) {
        if (idx < p.frame_tokens_cap && nb + idx < lp.tcap) { tok_slot[idx] = slot; tok_state[nb + idx] = state; K3_AST(&tok_cost[nb + idx], kEncMax); }
        else { sh.err = K3_ERR_OVERFLOW; idx = 0; }
        tb.set_tok(slot, idx);
      }
      {   // round 1 of the eps closure works on the new tokens whose state has eps arcs
        const bool q1 = claimed && nxt < 0;
        const int pos1 = wave_append(q1, &sh.n_wl[1]);
        if (q1) {
          if (pos1 < kWlLds) s_lwl[1][pos1] = slot < kHL ? (unsigned short)slot : (unsigned short)0xFFFF;
          if (pos1 >= kWlLds || slot >= kHL) { if (pos1 < p.frame_tokens_cap) wl[p.frame_tokens_cap + pos1] = slot; else sh.err = K3_ERR_OVERFLOW; }
        }
      }
      // ---- forward link of the accepted arc (:803-806)
      if (mk && !claimed) { idx = tb.wait_tok(slot, &sh.err); if (idx < 0) idx = 0; }
      const long long pos = wave_append64(mk, &sh.n_link);
      if (mk) {
        if (pos < lp.lcap) { store_link(&links[pos], Link{(unsigned)(cur_base + c_s), (unsigned)(nb + idx), tot, c_c}); store_stream(&link_arc[pos], c_a); }
        else sh.err = K3_ERR_OVERFLOW;
      }
    }
    if (row_pending) fetch_row();
    if (block_err(sh)) break;
    K3_T(5);
    // park the prefetched row (nobody reads the LDS row between here and the next frame's pre-pass)
    if (prefetch) {
#pragma unroll
      for (int k = 0; k < kRowRegs; k++) { const int i = tid + k * kBlock; if (i < p.num_pdfs) s_ll[i] = rowreg[k]; }
    }
    if (tid == 0) { loff_n[f + 1] = sh.n_link; st_ntoks[f] = n_cur; st_cur[f] = cur_cutoff; st_ab[f] = ab; st_next[f] = accept; st_co[f] = co; }
    }   // f >= 0
    // ---- ProcessNonemitting(next_cutoff) + eps links + publish the frame
    finish_frame(p, sh, tb, accept, nb, tok_state, tok_cost, links, link_arc, lp.tcap, lp.lcap, tok_slot, wl, s_lwl, creg, sreg, t_last__, cnt_eps);
    if (block_err(sh)) break;
    cur_base = nb; n_cur = sh.n_next; max_frame = n_cur > max_frame ? n_cur : max_frame; in_regs = n_cur <= kCurRegs * kBlock;
    __syncthreads();
    if (tid == 0) { tok_off[f + 2] = cur_base + n_cur; loff_e[f + 1] = sh.n_link; }
#ifdef K3_DEC_PROF
    // profiling builds only: FrameStats' adaptive_beam column = cycles of the frame
    if (tid == 0 && f >= 0) {
      const long long now__ = (long long)__builtin_readcyclecounter();
      st_ab[f] = (float)(now__ - t_frame__);
      t_frame__ = now__;
    }
#endif
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
#ifdef K3_DEC_PROF
  if (tid < 16) p.prof[blockIdx.x * 16 + tid] += sh.prof[tid];
#endif
  if (tid == 0) {
    LaneInfo &li = p.info[L];
    li.n_tokens = cur_base + n_cur; li.n_links = sh.n_link; li.n_cands = (long long)sh.n_emit; li.n_eps = (long long)sh.n_eps; li.max_frame_tokens = max_frame;
    li.status = sh.err ? sh.err : status; li.num_frames = f0 + T; li.reached_final = 0; li.out_states = 0; li.out_arcs = 0;
    li.cur_base = cur_base; li.n_cur = n_cur; li.n_order_sensitive = (long long)sh.n_os;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// FinalizeDecoding (:634-649) on the GPU: one workgroup per lane, frames in reverse.  tok_extra holds extra_cost.
// Inside a frame the eps links make the extra costs depend on each other; the dependency graph is acyclic, so the
// fixpoint is unique and reached by iterating x <- min(base, min_links f(x)) until nothing changes.
__device__ __forceinline__ float link_extra_cost(float next_extra, float via_link_tot, float next_tot) {
  return next_extra + (via_link_tot - next_tot);      // next_tok->extra_cost + ((tok->tot_cost + ac + graph) - next_tok->tot_cost), :339-341
}

// eps links are written while the closure is still running; one whose source token was improved afterwards is stale:
// its stamp (Link::ac) is not the final cost of its source (k3_decode_forward_kernel, finish_frame)
__device__ __forceinline__ bool eps_link_live(const Link &k, unsigned src_cost_enc) { return __float_as_uint(k.ac) == src_cost_enc; }

#ifndef K3_PCAP
#define K3_PCAP 3072
#endif
constexpr int kPCap = K3_PCAP;      // frames with at most this many tokens are pruned entirely inside LDS

#define K3_LIT_FINAL
#include "k3_decoder_literal.h"

__global__ __launch_bounds__(kPBlock, 4) void k3_decode_prune_kernel(DecParams p) {
  __shared__ int s_changed, s_has_final, s_chg[4];
  __shared__ unsigned s_best, s_best_final;
  __shared__ float s_cost[2][kPCap], s_extra[2][kPCap];   // token costs / extra costs of frames f+1 (buffer nb) and f (buffer nb ^ 1)
  __shared__ unsigned s_xb[kPCap];
  const int L = p.lane_ids ? p.lane_ids[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
  __builtin_amdgcn_s_setprio(2);      // (a latency chain like the token-passing kernel: its few instructions go ahead of the GEMM workgroups it shares CUs with in a pipeline)
  LaneInfo &li = p.info[L];
  if (li.status != kStOk) return;
  const int T = li.num_frames;
  const LanePool lp = k3_uniform_pool(p.pools[L]);
  const int *tok_state = lp.tok_state; const unsigned *tok_cost = lp.tok_cost;
  float *extra = lp.tok_extra;
  const Link *links = lp.links;
  const long long *tok_off = p.tok_off + L * p.fstride, *loff_e = p.link_off_e + L * p.fstride, *loff_n = p.link_off_n + L * p.fstride;
  const float kInf = __builtin_inff(); const float lb = p.lattice_beam;

  // survivors (tokens with extra != inf, links with link_extra <= lattice_beam) are recorded frame by frame
  __shared__ int s_nt, s_nl;
  if (tid == 0) { s_nt = 0; s_nl = 0; }
#ifdef K3_PRUNE_PROF
  __shared__ long long s_pprof[16];
  if (tid < 16) s_pprof[tid] = 0;
  long long pt_last__ = (long long)__builtin_readcyclecounter();
#endif
  int *live_tok = p.live_tok + (long long)L * p.live_cap; long long *live_link = p.live_link + (long long)L * p.live_cap;
  int *newidx = lp.newidx;
  auto keep_tok = [&](long long t) { const int pos = k3a_add(&s_nt, 1); if (pos < p.live_cap) live_tok[pos] = (int)t; newidx[t] = pos; };
  auto keep_link = [&](long long l) { const int pos = k3a_add(&s_nl, 1); if (pos < p.live_cap) live_link[pos] = l; };
  // ---- last frame: ComputeFinalCosts (:545-586) + PruneForwardLinksFinal (:385-467), in HBM (one frame only)
  const long long tb = tok_off[T], te = tok_off[T + 1];
  if (tid == 0) { s_best = kEncMax; s_best_final = kEncMax; s_has_final = 0; }
  __syncthreads();
  for (long long t = tb + tid; t < te; t += kPBlock) {
    const float c = dec(tok_cost[t]), fc = p.final_cost[tok_state[t]];
    k3a_min(&s_best, enc(c)); k3a_min(&s_best_final, enc(c + fc));
    if (fc != kInf) s_has_final = 1;
  }
  __syncthreads();
  const float best = dec(s_best), best_final = dec(s_best_final);
  const float final_best = (best_final != kInf) ? best_final : best;
  const bool final_empty = !s_has_final;
  if (tid == 0) { li.reached_final = s_has_final; li.final_best_cost = final_best; li.final_empty = final_empty; }
  for (long long t = tb + tid; t < te; t += kPBlock) extra[t] = 0.0f;        // tokens on the last frame start with extra_cost 0
  __syncthreads();
  if (p.literal) {
    // literal_order: the reference's in-place sweeps in token-list order with its 1e-5 stop rule (k3_decoder_literal.h)
    const long long l0 = loff_n[T], l1 = loff_e[T];
    __shared__ int s_lit_err, s_lit_m, s_lit_redi[kPBlock / 64];
    if (tid == 0) s_lit_err = 0;
    __syncthreads();
    lit_final_frame(p, L, tb, te, l0, l1, final_best, final_empty, tok_state, tok_cost, extra, links, &s_lit_err,
                    s_cost[0], s_extra[0], s_xb, reinterpret_cast<int *>(s_cost[1]), s_extra[1], s_lit_redi, &s_lit_m);
    __syncthreads();
    if (s_lit_err) { if (tid == 0) li.status = s_lit_err; return; }
    const int m = s_lit_m; const long long fc = p.frame_cands_cap;
    const unsigned *g_off = reinterpret_cast<const unsigned *>(p.c_dst + L * fc); const int *lid = p.c_arc + L * fc; const int *g_ldst = p.wl + 2ll * L * p.frame_tokens_cap;
    for (long long t = tb + tid; t < te; t += kPBlock) if (extra[t] != kInf) keep_tok(t);
    for (long long t = tb + tid; t < te; t += kPBlock) {
      if (extra[t] == kInf) continue;
      for (unsigned j = g_off[t - tb]; j < g_off[t - tb + 1]; j++) if (g_ldst[j] < 0) keep_link(l0 + lid[j]);
    }
    (void)m;
  } else
  {
    const long long l0 = loff_n[T], l1 = loff_e[T];
    // Jacobi sweeps: xn <- base, atomicMin over the surviving eps links evaluated at the previous sweep's extras (xn lives in
    // this lane's candidate scratch), until no extra cost changes.
    unsigned *xn = reinterpret_cast<unsigned *>(p.c_tot + (long long)L * p.frame_cands_cap);   // frame_cands_cap >= frame_tokens_cap (checked on the host)
    for (int sweep = 0; sweep < 100000; sweep++) {
      __syncthreads();
      if (tid == 0) s_changed = 0;
      for (long long t = tb + tid; t < te; t += kPBlock) {
        const float fc = final_empty ? 0.0f : p.final_cost[tok_state[t]];
        xn[t - tb] = enc(dec(tok_cost[t]) + fc - final_best);
      }
      __syncthreads();
      for (long long l = l0 + tid; l < l1; l += kPBlock) {
        const Link k = links[l];
        if (!eps_link_live(k, tok_cost[k.src])) continue;
        float le = link_extra_cost(extra[k.dst], k.tot, dec(tok_cost[k.dst]));
        if (!(le > lb)) { if (le < 0.0f) le = 0.0f; k3a_min(&xn[k.src - tb], enc(le)); }
      }
      __syncthreads();
      for (long long t = tb + tid; t < te; t += kPBlock) {
        float v = dec(xn[t - tb]);
        if (v > lb) v = kInf;
        if (__float_as_uint(v) != __float_as_uint(extra[t])) s_changed = 1;
        extra[t] = v;
      }
      __syncthreads();
      if (!s_changed) break;
    }
    for (long long t = tb + tid; t < te; t += kPBlock) if (extra[t] != kInf) keep_tok(t);
    for (long long l = l0 + tid; l < l1; l += kPBlock) {
      const Link k = links[l];
      if (eps_link_live(k, tok_cost[k.src]) && extra[k.src] != kInf && !(link_extra_cost(extra[k.dst], k.tot, dec(tok_cost[k.dst])) > lb)) keep_link(l);
    }
  }
  __syncthreads();
  // ---- frames T-1 .. 0: PruneForwardLinks(f, delta = 0) then PruneTokensForFrame(f+1) (tokens with extra = inf vanish).
  // Fast path: both frames fit in LDS -> per frame one round trip to HBM (its links + its token costs), everything else in LDS.
  // The data a frame needs from HBM (its emitting links, its eps links, its token costs) does not depend on the sweep itself:
  // it is requested into registers while the frame above is still being processed, so that the LDS path never waits for HBM.
  // Inside that path the barriers order LDS traffic only (no vmcnt wait: the prefetch stays in flight, and nothing written to
  // HBM there is read back before a full __syncthreads()).
  struct FrameOff { long long b0, b1, b2, e0, e1, n0, n1; };
  auto uni = [](long long v) {       // block-uniform value -> scalar registers
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  auto load_off = [&](int f) {
    FrameOff o{0, 0, 0, 0, 0, 0, 0};
    if (f >= 0) {
      o.b0 = uni(tok_off[f]);
      o.b1 = uni(tok_off[f + 1]);
      o.b2 = uni(tok_off[f + 2]);
      o.e0 = uni(loff_e[f]);
      o.e1 = uni(loff_n[f + 1]);
      o.n0 = uni(loff_n[f]);
      o.n1 = o.e0;
    }
    return o;
  };
  // frame f - 1's offsets share all but three values with frame f's: those are fetched a frame early and made scalar late
  long long raw_b0 = 0, raw_e0 = 0, raw_n0 = 0;
  auto request_off = [&](int f) { if (f >= 0) { raw_b0 = tok_off[f]; raw_e0 = loff_e[f]; raw_n0 = loff_n[f]; } };
  auto finish_off = [&](int f, const FrameOff &above) {      // `above` = offsets of frame f + 1
    FrameOff o{0, 0, 0, 0, 0, 0, 0};
    if (f >= 0) { o.b0 = uni(raw_b0); o.b1 = above.b0; o.b2 = above.b1; o.e0 = uni(raw_e0); o.e1 = above.n0; o.n0 = uni(raw_n0); o.n1 = o.e0; }
    return o;
  };
  auto fits = [&](const FrameOff &o) { return o.b1 - o.b0 <= kPCap && o.b2 - o.b1 <= kPCap; };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  constexpr int kPre = 3, kEpsRegs = 3, kCostRegs = kPCap / kPBlock;      // what a thread holds of the next frame (the kernel has 128 registers per thread)
  Link pl[kPre], er[kEpsRegs]; unsigned pc[kCostRegs]; bool pre_valid = false;
  auto prefetch_main = [&](const FrameOff &o) {      // emitting links + token costs of a frame
#pragma unroll
    for (int k = 0; k < kPre; k++) { const long long l = o.e0 + tid + k * kPBlock; if (l < o.e1) pl[k] = links[l]; }
#pragma unroll
    for (int k = 0; k < kCostRegs; k++) { const long long i = tid + k * kPBlock; if (i < o.b1 - o.b0) pc[k] = tok_cost[o.b0 + i]; }
  };
  auto prefetch_eps = [&](const FrameOff &o) {
#pragma unroll
    for (int k = 0; k < kEpsRegs; k++) { const long long i = tid + k * kPBlock; if (i < o.n1 - o.n0) er[k] = links[o.n0 + i]; }
  };
  K3_PT(0);      // last frame
  int nbuf = 0; bool next_in_lds = false;
  FrameOff o_cur = load_off(T - 1), o_nxt = load_off(T - 2);
  if (T >= 1 && fits(o_cur)) { prefetch_main(o_cur); pre_valid = true; }
  for (int f = T - 1; f >= 0; f--) {
    const FrameOff o = o_cur; o_cur = o_nxt;
    request_off(f - 2);
    const long long b0 = o.b0, b1 = o.b1, b2 = o.b2, e0 = o.e0, e1 = o.e1, n0 = o.n0, n1 = o.n1;
    const int nf = (int)(b1 - b0), nn = (int)(b2 - b1);
    if (nf <= kPCap && nn <= kPCap) {
      float *ncost = s_cost[nbuf], *nextra = s_extra[nbuf], *ccost = s_cost[nbuf ^ 1], *cextra = s_extra[nbuf ^ 1];
      if (!pre_valid) prefetch_main(o);                                  // the frame above took the HBM path
      if (!next_in_lds) { for (int i = tid; i < nn; i += kPBlock) { ncost[i] = dec(tok_cost[b1 + i]); nextra[i] = extra[b1 + i]; } }
#pragma unroll
      for (int k = 0; k < kCostRegs; k++) { const int i = tid + k * kPBlock; if (i < nf) { ccost[i] = dec(pc[k]); s_xb[i] = kEncInf; cextra[i] = 0.0f; } }
      const int neps = (int)(n1 - n0);
      prefetch_eps(o);          // this frame's eps links: needed after the emitting links, their latency hides behind those
      lds_barrier();
      K3_PT(1);    // staging
      auto emit_link = [&](const Link &k, long long l) {
        float le = link_extra_cost(nextra[k.dst - b1], k.tot, ncost[k.dst - b1]);
        if (!(le > lb)) { keep_link(l); if (le < 0.0f) le = 0.0f; k3a_min(&s_xb[k.src - b0], enc(le)); }     // a surviving emitting link keeps its source alive
      };
#pragma unroll
      for (int k = 0; k < kPre; k++) { const long long l = e0 + tid + k * kPBlock; if (l < e1) emit_link(pl[k], l); }
      for (long long l = e0 + kPre * kPBlock + tid; l < e1; l += kPBlock) emit_link(links[l], l);                  // (rare) more than fit in registers
      // the frame below: its emitting links and costs are requested now
      const bool pre_next = f >= 1 && fits(o_cur);
      if (pre_next) prefetch_main(o_cur);
      unsigned elive = 0;
#pragma unroll
      for (int k = 0; k < kEpsRegs; k++) { const int i = tid + k * kPBlock; if (i < neps && eps_link_live(er[k], enc(ccost[er[k].src - b0]))) elive |= 1u << k; }
      lds_barrier();
      K3_PT(2);    // emitting links
      if (neps != 0) {
        // eps links relax s_xb downwards in place until nothing moves (acyclic dependencies: the unique fixpoint, reached link by
        // link).  The "changed" flags rotate over four slots so that one barrier per sweep is enough.
        if (tid < 4) s_chg[tid] = 0;
        lds_barrier();
        for (int sweep = 0; sweep < 100000; sweep++) {
          bool moved = false;
#pragma unroll
          for (int k = 0; k < kEpsRegs; k++) {
            if (elive >> k & 1) {
              float le = link_extra_cost(dec(K3_LLD(&s_xb[er[k].dst - b0])), er[k].tot, ccost[er[k].dst - b0]);
              if (!(le > lb)) { if (le < 0.0f) le = 0.0f; const unsigned e = enc(le); if (e < k3a_min(&s_xb[er[k].src - b0], e)) moved = true; }
            }
          }
          for (long long l = n0 + kEpsRegs * kPBlock + tid; l < n1; l += kPBlock) {       // (rare) more eps links than fit in registers
            const Link k = links[l];
            if (!eps_link_live(k, enc(ccost[k.src - b0]))) continue;
            float le = link_extra_cost(dec(K3_LLD(&s_xb[k.dst - b0])), k.tot, ccost[k.dst - b0]);
            if (!(le > lb)) { if (le < 0.0f) le = 0.0f; const unsigned e = enc(le); if (e < k3a_min(&s_xb[k.src - b0], e)) moved = true; }
          }
          if (moved) s_chg[sweep & 3] = 1;
          if (tid == 0) s_chg[(sweep + 2) & 3] = 0;      // read last after the barrier of sweep - 2: every wavefront is past it
          lds_barrier();
          if (!s_chg[sweep & 3]) break;
        }
      }
      for (int i = tid; i < nf; i += kPBlock) { const float v = dec(s_xb[i]); cextra[i] = v; extra[b0 + i] = v; if (v != kInf) keep_tok(b0 + i); }
      if (neps != 0) {
        lds_barrier();
#pragma unroll
        for (int k = 0; k < kEpsRegs; k++) {
          const int i = tid + k * kPBlock;
          if ((elive >> k & 1) && !(link_extra_cost(cextra[er[k].dst - b0], er[k].tot, ccost[er[k].dst - b0]) > lb)) keep_link(n0 + i);
        }
        for (long long l = n0 + kEpsRegs * kPBlock + tid; l < n1; l += kPBlock) {
          const Link k = links[l];
          if (eps_link_live(k, enc(ccost[k.src - b0])) && !(link_extra_cost(cextra[k.dst - b0], k.tot, ccost[k.dst - b0]) > lb)) keep_link(l);
        }
      }
      K3_PT(3);    // eps fixpoint + write-back
      pre_valid = pre_next;
      nbuf ^= 1; next_in_lds = true;
      o_nxt = finish_off(f - 2, o_cur);
      lds_barrier();
      K3_PT(4);    // offsets + end barrier
      continue;
    }
    __syncthreads();                   // leaving the LDS path: what it wrote to HBM (extra costs) is read below
    pre_valid = false;
    next_in_lds = false;
    // HBM path (a frame above kPCap tokens).  x[] = order-preserving image of the extra costs of the frame's tokens, in this
    // lane's candidate scratch.  Emitting links first, then the eps links relax x downwards in place (atomicMin) until nothing
    // moves: the eps dependency graph is acyclic, so this reaches the same unique fixpoint as the reference's sweeps, link by
    // link instead of token by token.  Loads are issued kB links at a time: a frame of 25 k tokens would otherwise pay one
    // memory round trip per 512 links.
    unsigned *x = reinterpret_cast<unsigned *>(p.c_tot + (long long)L * p.frame_cands_cap);
    constexpr int kB = 4, kE = 2;      // links in flight per thread: emitting / eps passes (register budget of the kernel: 128)
    for (long long t = b0 + tid; t < b1; t += kPBlock) x[t - b0] = kEncInf;
    __syncthreads();
    for (long long l0 = e0 + tid; l0 < e1; l0 += kB * kPBlock) {
      Link k[kB]; float xe[kB], xc[kB];
#pragma unroll
      for (int j = 0; j < kB; j++) { const long long l = l0 + j * kPBlock; if (l < e1) k[j] = links[l]; }
#pragma unroll
      for (int j = 0; j < kB; j++) { const long long l = l0 + j * kPBlock; if (l < e1) { xe[j] = extra[k[j].dst]; xc[j] = dec(tok_cost[k[j].dst]); } }
#pragma unroll
      for (int j = 0; j < kB; j++) {
        const long long l = l0 + j * kPBlock;
        if (l < e1) { float le = link_extra_cost(xe[j], k[j].tot, xc[j]); if (!(le > lb)) { keep_link(l); if (le < 0.0f) le = 0.0f; k3a_min(&x[k[j].src - b0], enc(le)); } }
      }
    }
    __syncthreads();
    if (n1 != n0) {
      for (int sweep = 0; sweep < 100000; sweep++) {
        __syncthreads();
        if (tid == 0) s_changed = 0;
        __syncthreads();
        for (long long l0 = n0 + tid; l0 < n1; l0 += kE * kPBlock) {
          Link k[kE]; unsigned sc[kE], xd[kE]; float dc[kE];
#pragma unroll
          for (int j = 0; j < kE; j++) { const long long l = l0 + j * kPBlock; if (l < n1) k[j] = links[l]; }
#pragma unroll
          for (int j = 0; j < kE; j++) {
            const long long l = l0 + j * kPBlock;
            if (l < n1) {
              sc[j] = tok_cost[k[j].src];
              xd[j] = K3_ALD(&x[k[j].dst - b0]);
              dc[j] = dec(tok_cost[k[j].dst]);
            }
          }
#pragma unroll
          for (int j = 0; j < kE; j++) {
            const long long l = l0 + j * kPBlock;
            if (l < n1 && eps_link_live(k[j], sc[j])) {
              float le = link_extra_cost(dec(xd[j]), k[j].tot, dc[j]);
              if (!(le > lb)) { if (le < 0.0f) le = 0.0f; const unsigned e = enc(le); if (e < k3a_min(&x[k[j].src - b0], e)) s_changed = 1; }
            }
          }
        }
        __syncthreads();
        if (!s_changed) break;
      }
    }
    for (long long t0 = b0 + tid; t0 < b1; t0 += kB * kPBlock) {
      unsigned v[kB];
#pragma unroll
      for (int j = 0; j < kB; j++) { const long long t = t0 + j * kPBlock; if (t < b1) v[j] = x[t - b0]; }
#pragma unroll
      for (int j = 0; j < kB; j++) { const long long t = t0 + j * kPBlock; if (t < b1) { const float e = dec(v[j]); extra[t] = e; if (e != kInf) keep_tok(t); } }
    }
    for (long long l0 = n0 + tid; l0 < n1; l0 += kE * kPBlock) {
      Link k[kE]; unsigned sc[kE], xd[kE], xs[kE]; float dc[kE];
#pragma unroll
      for (int j = 0; j < kE; j++) { const long long l = l0 + j * kPBlock; if (l < n1) k[j] = links[l]; }
#pragma unroll
      for (int j = 0; j < kE; j++) {
        const long long l = l0 + j * kPBlock;
        if (l < n1) {
          sc[j] = tok_cost[k[j].src];
          xd[j] = x[k[j].dst - b0];
          xs[j] = x[k[j].src - b0];
          dc[j] = dec(tok_cost[k[j].dst]);
        }
      }
#pragma unroll
      for (int j = 0; j < kE; j++) {
        const long long l = l0 + j * kPBlock;
        if (l < n1 && eps_link_live(k[j], sc[j]) && xs[j] != kEncInf && !(link_extra_cost(dec(xd[j]), k[j].tot, dc[j]) > lb)) keep_link(l);
      }
    }
    o_nxt = finish_off(f - 2, o_cur);
    __syncthreads();
    K3_PT(5);      // HBM-path frame
  }
  __syncthreads();
#ifdef K3_PRUNE_PROF
  if (tid < 16) p.prof[blockIdx.x * 16 + tid] = s_pprof[tid];
#endif
  if (tid == 0) { li.out_states = s_nt; li.out_arcs = s_nl; li.live_overflow = (s_nt > p.live_cap || s_nl > p.live_cap) ? 1 : 0; }
}

// ---------------------------------------------------------------------------------------------------------------
// GetRawLattice (:114-197): compact the surviving tokens / links of every lane into the output arrays.
struct OutParams {
  const long long *st_off, *arc_off;     // [U+1] prefix sums of out_states / out_arcs (device)
  int *st_frame, *st_state; float *st_cost, *st_final; int *arc_src, *arc_dst, *arc_il, *arc_ol; float *arc_g, *arc_ac;
};

__global__ __launch_bounds__(kPBlock) void k3_decode_output_kernel(DecParams p, OutParams o) {
  __shared__ int s_n;
  const int L = p.lane_ids ? p.lane_ids[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
  const LaneInfo &li = p.info[L];
  if (li.status != kStOk) return;
  const int T = li.num_frames;
  const LanePool lp = k3_uniform_pool(p.pools[L]);
  const int *tok_state = lp.tok_state; const unsigned *tok_cost = lp.tok_cost;
  const float *extra = lp.tok_extra;
  const Link *links = lp.links; const int *link_arc = lp.link_arc;
  const long long *tok_off = p.tok_off + L * p.fstride, *loff_e = p.link_off_e + L * p.fstride, *loff_n = p.link_off_n + L * p.fstride;
  const float *st_co = p.st_co + L * p.fstride;
  int *newidx = lp.newidx;
  const float kInf = __builtin_inff(); const float lb = p.lattice_beam;
  const long long so = o.st_off[blockIdx.x], ao = o.arc_off[blockIdx.x];      // offsets are per finalised lane, in launch order
  if (!li.live_overflow) {       // survivors were listed by the pruning pass: touch only them
    const int *live_tok = p.live_tok + (long long)L * p.live_cap; const long long *live_link = p.live_link + (long long)L * p.live_cap;
    for (int i = tid; i < li.out_states; i += kPBlock) {
      const long long t = live_tok[i];
      int lo = 0, hi = T;                                  // frame f with tok_off[f] <= t < tok_off[f+1]
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tok_off[mid] <= t) lo = mid; else hi = mid - 1; }
      o.st_frame[so + i] = lo; o.st_state[so + i] = tok_state[t]; o.st_cost[so + i] = dec(tok_cost[t]);
      float fin = kInf;
      if (lo == T) { if (li.final_empty) fin = 0.0f; else fin = p.final_cost[tok_state[t]]; }
      o.st_final[so + i] = fin;
    }
    for (int i = tid; i < li.out_arcs; i += kPBlock) {
      const long long l = live_link[i]; const Link k = links[l];
      int lo = 0, hi = T;                                  // frame f with loff_n[f] <= l < loff_n[f+1]
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (loff_n[mid] <= l) lo = mid; else hi = mid - 1; }
      const bool emitting = l >= loff_e[lo];
      const int arc = link_arc[l]; const ArcRec r = p.arcs[arc];
      o.arc_src[ao + i] = newidx[k.src]; o.arc_dst[ao + i] = newidx[k.dst]; o.arc_il[ao + i] = p.arc_ilabel[arc]; o.arc_ol[ao + i] = r.olabel;
      o.arc_g[ao + i] = r.w; o.arc_ac[ao + i] = emitting ? (k.ac - st_co[lo]) : 0.0f;
    }
    return;
  }
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int f = 0; f <= T; f++) {          // frame-major numbering (any per-frame order is a valid GetRawLattice numbering)
    const long long b0 = tok_off[f], b1 = tok_off[f + 1];
    for (long long t0 = b0; t0 < b1; t0 += kPBlock) {
      const long long t = t0 + tid; const bool v = t < b1 && extra[t] != kInf;
      const int pos = wave_append(v, &s_n);
      if (v) {
        newidx[t] = pos;
        o.st_frame[so + pos] = f; o.st_state[so + pos] = tok_state[t]; o.st_cost[so + pos] = dec(tok_cost[t]);
        float fin = kInf;
        if (f == T) { if (li.final_empty) fin = 0.0f; else fin = p.final_cost[tok_state[t]]; }
        o.st_final[so + pos] = fin;
      }
    }
  }
  __syncthreads();
  if (tid == 0) s_n = 0;
  __syncthreads();
  // links in pool order; frame of a link: emitting links of frame f lie in [loff_e[f], loff_n[f+1])
  for (int f = 0; f <= T; f++) {
    const long long n0 = loff_n[f], n1 = loff_e[f];
    const long long e0 = loff_e[f], e1 = (f < T) ? loff_n[f + 1] : loff_e[f];
    for (int part = 0; part < 2; part++) {
      const long long l0 = part ? e0 : n0, l1 = part ? e1 : n1;
      for (long long x0 = l0; x0 < l1; x0 += kPBlock) {
        const long long l = x0 + tid; bool v = l < l1; Link k{};
        if (v) {
          k = links[l];
          v = extra[k.src] != kInf && (part == 1 || eps_link_live(k, tok_cost[k.src]));
          if (v) { const float le = link_extra_cost(extra[k.dst], k.tot, dec(tok_cost[k.dst])); v = !(le > lb); }
        }
        const int pos = wave_append(v, &s_n);
        if (v) {
          const int arc = link_arc[l]; const ArcRec r = p.arcs[arc];
          o.arc_src[ao + pos] = newidx[k.src]; o.arc_dst[ao + pos] = newidx[k.dst]; o.arc_il[ao + pos] = p.arc_ilabel[arc]; o.arc_ol[ao + pos] = r.olabel;
          o.arc_g[ao + pos] = r.w; o.arc_ac[ao + pos] = part ? (k.ac - st_co[f]) : 0.0f;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Best-path traceback on the token / link pools of a lane, finalised or not (CudaDecoder::GetBestPath cuda-decoder.cc:1072-1179, and the
// traceback behind GetPartialHypothesis / EndpointDetected :1864-1960).  A token's cost is the minimum over its incoming links of the
// link's `tot`, so the best predecessor of a token is an incoming link whose `tot` equals the token's cost bit for bit (ties: the link
// created first).  One workgroup per requested lane walks back from the best token of the newest frame: per step one scan over the eps
// links of the frame, then over the emitting links into it.  Output per lane: the arcs of the path, last arc first.
struct BestPathParams { int *path_il, *path_ol; float *path_g, *path_ac; int *path_len; float *final_cost, *relative_cost; int *reached_final; int cap; int use_final; };

__global__ __launch_bounds__(kPBlock) void k3_decode_best_path_kernel(DecParams p, BestPathParams o) {
  __shared__ unsigned long long s_best; __shared__ unsigned s_min, s_minf; __shared__ long long s_link;
  const int L = p.lane_ids[blockIdx.x], tid = threadIdx.x;
  const LaneInfo &li = p.info[L];
  const long long po = (long long)blockIdx.x * o.cap;
  if (tid == 0) { o.path_len[blockIdx.x] = 0; o.final_cost[blockIdx.x] = 0.0f; o.relative_cost[blockIdx.x] = __builtin_inff(); o.reached_final[blockIdx.x] = 0; }
  if (li.status != kStOk) return;
  const int T = li.num_frames;
  const LanePool lp = k3_uniform_pool(p.pools[L]);
  const int *tok_state = lp.tok_state; const unsigned *tok_cost = lp.tok_cost;
  const Link *links = lp.links; const int *link_arc = lp.link_arc;
  const long long *tok_off = p.tok_off + L * p.fstride, *loff_e = p.link_off_e + L * p.fstride, *loff_n = p.link_off_n + L * p.fstride;
  const float *st_co = p.st_co + L * p.fstride; const float kInf = __builtin_inff();
  // best token of the newest frame: min (cost + final) if a final state was reached and final-probs are asked for, else min cost
  const long long tb = tok_off[T], te = tok_off[T + 1];
  if (tid == 0) { s_min = kEncMax; s_minf = kEncMax; s_best = ~0ull; }
  __syncthreads();
  for (long long t = tb + tid; t < te; t += kPBlock) { const float c = dec(tok_cost[t]); k3a_min(&s_min, enc(c)); k3a_min(&s_minf, enc(c + p.final_cost[tok_state[t]])); }
  __syncthreads();
  const bool any_final = s_minf != kEncMax && dec(s_minf) != kInf; const bool with_final = o.use_final && any_final;
  for (long long t = tb + tid; t < te; t += kPBlock) {
    const float c = dec(tok_cost[t]); const float v = with_final ? c + p.final_cost[tok_state[t]] : c;
    k3a_min(&s_best, ((unsigned long long)enc(v) << 32) | (unsigned)(t - tb));
  }
  __syncthreads();
  if (te == tb) return;
  long long cur = tb + (long long)(s_best & 0xFFFFFFFFull);
  if (tid == 0) {
    o.reached_final[blockIdx.x] = any_final;
    o.final_cost[blockIdx.x] = with_final ? p.final_cost[tok_state[cur]] : 0.0f;
    o.relative_cost[blockIdx.x] = any_final ? dec(s_minf) - dec(s_min) : kInf;
  }
  int len = 0, f = T;
  for (int guard = 0; guard < 4 * (T + 1) + 64; guard++) {
    const unsigned cc = tok_cost[cur];
    __syncthreads();
    if (tid == 0) s_link = 0x7FFFFFFFFFFFFFFFll;
    __syncthreads();
    // an eps link of frame f into cur (source in the same frame), stamped with its source's final cost (live), whose tot is cur's cost
    for (long long l = loff_n[f] + tid; l < loff_e[f]; l += kPBlock) {
      const Link k = links[l];
      if (k.dst == (unsigned)cur && __float_as_uint(k.tot) == __float_as_uint(dec(cc)) &&
          eps_link_live(k, tok_cost[k.src])) k3a_min((unsigned long long *)&s_link, (unsigned long long)l);
    }
    __syncthreads();
    long long best = s_link; bool emitting = false;
    if (best == 0x7FFFFFFFFFFFFFFFll && f > 0) {
      __syncthreads();
      for (long long l = loff_e[f - 1] + tid; l < loff_n[f]; l += kPBlock) {
        const Link k = links[l];
        if (k.dst == (unsigned)cur && __float_as_uint(k.tot) == __float_as_uint(dec(cc))) k3a_min((unsigned long long *)&s_link, (unsigned long long)l);
      }
      __syncthreads();
      best = s_link; emitting = true;
    }
    if (best == 0x7FFFFFFFFFFFFFFFll) break;      // the start token (or a token nothing points to: cannot happen for a token with finite cost)
    const Link k = links[best];
    if (len >= o.cap) { if (tid == 0) o.path_len[blockIdx.x] = -1; return; }
    if (tid == 0) {
      const int a = link_arc[best];
      const ArcRec r = p.arcs[a];
      o.path_il[po + len] = p.arc_ilabel[a];
      o.path_ol[po + len] = r.olabel;
      o.path_g[po + len] = r.w;
      o.path_ac[po + len] = emitting ? k.ac - st_co[f - 1] : 0.0f;
    }
    len++; cur = k.src; if (emitting) f--;
  }
  if (tid == 0) o.path_len[blockIdx.x] = len;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

// k3_decoder_lit.hip: the literal_order token-passing kernel (compiled with its own block size); `params` is this file's DecParams
extern "C" int k3_lit_forward_prepare();
extern "C" int k3_lit_fast_tokens();
extern "C" int k3_lit_capture_launch(const void *params, size_t params_bytes, int nworkgroups, hipStream_t stream);      // k3_decoder_lit_cap.hip
extern "C" int k3_lit_has_queue();      // 0: this build's kernel decodes one lane per workgroup only (resident_lanes is then ignored)
extern "C" void k3_lit_forward_launch(const void *params, size_t params_bytes, int nworkgroups, int flags, hipStream_t stream);      // flags: 1 exclusive LDS, 2 no template kernels

// ------------------------------------------------------------------------------------------------ graph ----
struct k3_fst {
  // max_pdf: largest column of the log-likelihood matrix an arc reads (-1: unknown, image imported)
  int32_t num_states = 0, start = 0;
  int64_t num_arcs = 0;
  int32_t max_pdf = -1;
  void *image = nullptr; size_t bytes = 0;
  int2 *offs = nullptr; ArcRec *arcs = nullptr; float *final_cost = nullptr; int *arc_ilabel = nullptr;
  // derived from the image when a decoder first asks for it (not part of the image: a graph that arrived through k3_fst_bcast / k3_fst_import_image derives its own;
  // dropped when the image is handed out for writing): per arc {first arc of the destination state, its emitting arcs | eps arcs << 16}
  mutable int2 *dinfo = nullptr;
  void drop_derived() const { if (dinfo) { (void)hipFree(dinfo); dinfo = nullptr; } }
  ~k3_fst() { drop_derived(); if (image) (void)hipFree(image); }
};

namespace {
__global__ __launch_bounds__(256) void k3_fst_dinfo_kernel(const int2 *offs, const ArcRec *arcs, long long num_arcs, int2 *dinfo) {
  const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
  if (a >= num_arcs) return;
  const int nxt = (int)((unsigned)arcs[a].next & ~kEpsFlag);
  const int2 o = offs[nxt]; const int e1 = offs[nxt + 1].x;
  const int ne = o.y - o.x, nn = e1 - o.y;
  dinfo[a] = make_int2(o.x, (ne < 65535 ? ne : 65535) | ((nn < 65535 ? nn : 65535) << 16));
}
}  // namespace
static int fst_dinfo(const k3_fst *f, const int2 **out) {
  if (!f->dinfo && f->num_arcs > 0) {
    K3_HIP_CHECK(hipMalloc((void **)&f->dinfo, sizeof(int2) * (size_t)f->num_arcs));
    hipLaunchKernelGGL(k3_fst_dinfo_kernel, dim3((unsigned)((f->num_arcs + 255) / 256)), dim3(256), 0, nullptr, f->offs, f->arcs, (long long)f->num_arcs, f->dinfo);
    K3_HIP_CHECK(hipGetLastError());
    K3_HIP_CHECK(hipDeviceSynchronize());
  }
  *out = f->dinfo;
  return K3_OK;
}

static int fst_alloc(k3_fst *f) {
  const size_t o0 = 0, o1 = align_up(o0 + sizeof(int2) * (size_t)(f->num_states + 1), 256), o2 = align_up(o1 + sizeof(ArcRec) * (size_t)f->num_arcs, 256),
               o3 = align_up(o2 + sizeof(float) * (size_t)f->num_states, 256), o4 = align_up(o3 + sizeof(int) * (size_t)f->num_arcs, 256);
  f->bytes = o4;
  K3_HIP_CHECK(hipMalloc(&f->image, f->bytes));
  char *b = (char *)f->image;
  f->offs = (int2 *)(b + o0); f->arcs = (ArcRec *)(b + o1); f->final_cost = (float *)(b + o2); f->arc_ilabel = (int *)(b + o3);
  return K3_OK;
}

extern "C" int k3_fst_create(int32_t num_states, int32_t start, const int32_t *h_off, const int32_t *h_il, const int32_t *h_ol, const float *h_w,
                             const int32_t *h_next, const float *h_final, const int32_t *h_tid2pdf, int32_t num_tids, k3_fst **out) {
  K3_REQUIRE(num_states > 0 && start >= 0 && start < num_states && h_off && h_il && h_ol && h_w && h_next && h_final && h_tid2pdf && out, "k3_fst_create: bad argument");
  std::unique_ptr<k3_fst> f(new k3_fst());
  f->num_states = num_states; f->start = start; f->num_arcs = h_off[num_states];
  { const int rc = fst_alloc(f.get()); if (rc) return rc; }
  std::vector<char> host(f->bytes, 0);
  int2 *offs = (int2 *)(host.data() + ((char *)f->offs - (char *)f->image));
  ArcRec *arcs = (ArcRec *)(host.data() + ((char *)f->arcs - (char *)f->image));
  float *fin = (float *)(host.data() + ((char *)f->final_cost - (char *)f->image));
  int *ail = (int *)(host.data() + ((char *)f->arc_ilabel - (char *)f->image));
  // bit 31 of ArcRec::next = "the destination has epsilon arcs": the eps closure then never queues tokens that cannot expand
  std::vector<char> has_eps(num_states, 0);
  for (int32_t s = 0; s < num_states; s++) for (int32_t a = h_off[s]; a < h_off[s + 1]; a++) if (h_il[a] == 0) { has_eps[s] = 1; break; }
  int64_t pos = 0;
  for (int32_t s = 0; s < num_states; s++) {       // emitting arcs of a state first, then its eps arcs, FST order kept inside each group
    offs[s].x = (int)pos;
    for (int pass = 0; pass < 2; pass++) {
      if (pass == 1) offs[s].y = (int)pos;
      for (int32_t a = h_off[s]; a < h_off[s + 1]; a++) {
        const bool emit = h_il[a] != 0;
        if (emit != (pass == 0)) continue;
        if (h_next[a] < 0 || h_next[a] >= num_states) { k3::set_error("k3_fst_create: arc %d has next state %d out of range", a, h_next[a]); return K3_ERR_ARG; }
        int pdf = 0;
        if (emit) {
          if (h_il[a] < 0 || h_il[a] >= num_tids) { k3::set_error("k3_fst_create: ilabel %d outside the transition-id map [1, %d)", h_il[a], num_tids); return K3_ERR_ARG; }
          pdf = h_tid2pdf[h_il[a]];
          if (pdf < 0) { k3::set_error("k3_fst_create: transition-id %d maps to pdf %d", h_il[a], pdf); return K3_ERR_ARG; }
          f->max_pdf = std::max(f->max_pdf, pdf);
        }
        arcs[pos] = ArcRec{(int)((unsigned)h_next[a] | (has_eps[h_next[a]] ? kEpsFlag : 0u)), h_w[a], pdf, h_ol[a]}; ail[pos] = h_il[a]; pos++;
      }
    }
    fin[s] = h_final[s];
  }
  offs[num_states].x = (int)pos; offs[num_states].y = (int)pos;
  K3_HIP_CHECK(hipMemcpy(f->image, host.data(), f->bytes, hipMemcpyHostToDevice));
  *out = f.release();
  return K3_OK;
}

extern "C" int k3_fst_create_empty(int32_t num_states, int64_t num_arcs, int32_t start, k3_fst **out) {
  K3_REQUIRE(num_states > 0 && num_arcs >= 0 && out, "k3_fst_create_empty: bad argument");
  std::unique_ptr<k3_fst> f(new k3_fst());
  f->num_states = num_states; f->start = start; f->num_arcs = num_arcs;
  { const int rc = fst_alloc(f.get()); if (rc) return rc; }
  *out = f.release();
  return K3_OK;
}
extern "C" int k3_fst_export_image(const k3_fst *f, void *d_dst) {
  K3_REQUIRE(f && d_dst, "k3_fst_export_image: null argument");
  K3_HIP_CHECK(hipMemcpy(d_dst, f->image, f->bytes, hipMemcpyDeviceToDevice));
  return K3_OK;
}
extern "C" int k3_fst_import_image(k3_fst *f, const void *d_src) {
  K3_REQUIRE(f && d_src, "k3_fst_import_image: null argument");
  f->drop_derived();
  K3_HIP_CHECK(hipMemcpy(f->image, d_src, f->bytes, hipMemcpyDeviceToDevice));
  return K3_OK;
}
extern "C" int k3_fst_shape_and_image(const k3_fst *f, int64_t *shape, void **d_image) {      // (k3_comm.hip)
  K3_REQUIRE(f && shape && d_image, "k3_fst_shape_and_image: null argument");
  f->drop_derived();      // (the caller may write the image: k3_fst_bcast on a receiving rank)
  shape[0] = f->num_states; shape[1] = f->num_arcs; shape[2] = f->start; shape[3] = (int64_t)f->bytes; shape[4] = f->max_pdf; *d_image = f->image; return K3_OK;
}
extern "C" int k3_fst_create_shaped(const int64_t *shape, k3_fst **out) {
  k3_fst *f = nullptr; const int rc = k3_fst_create_empty((int32_t)shape[0], shape[1], (int32_t)shape[2], &f); if (rc) return rc;
  if ((int64_t)f->bytes != shape[3]) { delete f; k3::set_error("k3_fst_bcast: image size mismatch between ranks (different library builds?)"); return K3_ERR_ARG; }
  f->max_pdf = (int32_t)shape[4]; *out = f; return K3_OK;
}
extern "C" void k3_fst_destroy(k3_fst *f) { delete f; }
extern "C" int64_t k3_fst_num_arcs(const k3_fst *f) { return f ? f->num_arcs : -1; }
extern "C" int32_t k3_fst_num_states(const k3_fst *f) { return f ? f->num_states : -1; }
extern "C" int32_t k3_fst_start(const k3_fst *f) { return f ? f->start : -1; }
extern "C" int k3_fst_image(const k3_fst *f, void **d_image, int64_t *bytes) {
  K3_REQUIRE(f && d_image && bytes, "k3_fst_image: null argument");
  f->drop_derived();
  *d_image = f->image; *bytes = (int64_t)f->bytes; return K3_OK;
}

// ------------------------------------------------------------------------------------------------ decoder ----
struct k3_decoder {
  const k3_fst *fst = nullptr; k3_decoder_config cfg{}; int nlanes = 0, num_pdfs = 0;
  DecParams p{};
  std::vector<void *> allocs;
  long long fstride = 0; std::vector<void *> frame_allocs;
  long long *d_row_off = nullptr; int *d_fresh = nullptr, *d_lane_ids = nullptr; const float **d_lane_rows = nullptr;
  int last_utts = 0; std::vector<int> last_frames, fresh, lane_final;   // per lane: frames consumed, InitDecoding pending, FinalizeDecoding done
  std::vector<int> sel;                                                   // lanes of the latest finalize call (what the lattice getters return)
  hipStream_t last_stream = nullptr;
  std::vector<int> lane_ids_uploaded;      // what d_lane_ids holds
  std::vector<LaneInfo> h_info; bool info_valid = false;
  void *out_buf = nullptr; size_t out_bytes = 0;
  bool started = false, finalized = false;
  bool profiling = false; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  // recorded behind every token-passing launch: the last reader of the caller's log-likelihoods (k3_decoder_stream_wait_token_passing)
  hipEvent_t ev_tp = nullptr;
  bool ev_tp_recorded = false;
  // Per-call arguments of AdvanceDecoding (row offsets, per-lane frame pointers, "fresh" flags): a ring of slots, each a page-locked host block + its device
  // copy + an event recorded
  // behind the launch that reads it.  The call fills a slot, copies it asynchronously on the caller's stream and returns: no hipStreamSynchronize + synchronous hipMemcpy per chunk
  // (round 4: a streaming round of 17 frames paid a stream drain and three blocking copies).  A slot is reused kArgSlots calls later, after its event.
  static constexpr int kArgSlots = 4;
  struct ArgSlot { char *h = nullptr, *d = nullptr; hipEvent_t ev = nullptr; bool used = false; };
  ArgSlot arg[kArgSlots]; unsigned arg_seq = 0; size_t arg_off_rows = 0, arg_off_fresh = 0, arg_off_queue = 0, arg_bytes = 0;
  // literal_order launch shape: > 0 = that many workgroups take the call's lanes from a work-queue (longest first) instead of one workgroup per lane; exclusive: a workgroup
  // asks for more than half of a CU's LDS, so that it shares its CU with other kernels' workgroups (the next batch's front end) instead of a second lane
  int lit_resident = 0; bool lit_exclusive = false;
  bool row_skip_dirty = false;      // DecParams::row_skip may hold a 1 from an earlier call (launch_literal)
  bool capture_launch = false;      // (k3_decoder_create only: the next token-passing launch is the capture build's, see build_frame0_template)

  ~k3_decoder() {
    for (void *q : allocs) (void)hipFree(q);
    for (void *q : frame_allocs) (void)hipFree(q);
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    if (ev_tp) (void)hipEventDestroy(ev_tp);
    if (out_buf) (void)hipFree(out_buf);
    for (ArgSlot &a : arg) {
      if (a.h) (void)hipHostFree(a.h);
      if (a.d) (void)hipFree(a.d);
      if (a.ev) (void)hipEventDestroy(a.ev);
    }
  }
};

extern "C" int k3_decoder_init_decoding(k3_decoder *d, int32_t num_utts, int32_t max_total_frames, void *stream);
extern "C" int k3_decoder_advance_decoding(k3_decoder *d, int32_t num_utts, const float *d_loglikes, int64_t ld, const int64_t *h_row_off, void *stream);

extern "C" void k3_decoder_config_default(k3_decoder_config *c) {
  if (!c) return;
  c->beam = 16.0f; c->max_active = std::numeric_limits<int32_t>::max(); c->min_active = 200; c->lattice_beam = 10.0f; c->beam_delta = 0.5f;
  c->frame_tokens_cap = 32768; c->frame_cands_cap = 65536; c->lane_tokens_cap = 2000000; c->lane_links_cap = 4000000;
  c->literal_order = 0; c->hash_ratio = 2.0f; c->fast_frame_tokens = -1; c->spare_pool_bytes = -1; c->resident_lanes = 0; c->resident_exclusive = 0;
}

template <typename T> static int dmalloc(std::vector<void *> *allocs, T **ptr, size_t n) {
  *ptr = nullptr;
  K3_HIP_CHECK(hipMalloc((void **)ptr, std::max<size_t>(n, 1) * sizeof(T)));
  allocs->push_back(*ptr);
  return K3_OK;
}

// InitDecoding's result is the same for every utterance of a decoder: decode "no frames" on lane 0 once (the literal_order kernel's own f == -1 pass), keep what it left as
// the template DecParams::tpl_* describes, and give the lane back.  A start closure beyond the template's bounds (or one that fails) simply leaves the decoder without one.
static int build_init_template(k3_decoder *d) {
  DecParams &p = d->p; int rc;
  if ((rc = k3_decoder_init_decoding(d, 1, 1, nullptr))) return rc;
  const int64_t ro[2] = {0, 0};
  if ((rc = k3_decoder_advance_decoding(d, 1, reinterpret_cast<const float *>(p.prof), d->num_pdfs, ro, nullptr))) return rc;
  K3_HIP_CHECK(hipDeviceSynchronize());
  LaneInfo li; LanePool pool0;
  K3_HIP_CHECK(hipMemcpy(&li, p.info, sizeof(li), hipMemcpyDeviceToHost));
  K3_HIP_CHECK(hipMemcpy(&pool0, p.pools, sizeof(pool0), hipMemcpyDeviceToHost));
  const long long n = li.n_tokens, nl = li.n_links;
  if (li.status == kStOk && li.cur_base == 0 && li.n_cur == n && n > 0 && n <= 8192 && nl <= 32768) {
    int *t_state = nullptr, *t_arc = nullptr, *t_order = nullptr, *t_by_ins = nullptr; unsigned *t_cost = nullptr; Link *t_links = nullptr;
    if ((rc = dmalloc(&d->allocs, &t_state, (size_t)n)) || (rc = dmalloc(&d->allocs, &t_cost, (size_t)n)) || (rc = dmalloc(&d->allocs, &t_order, (size_t)n)) ||
        (rc = dmalloc(&d->allocs, &t_by_ins, (size_t)n)) || (rc = dmalloc(&d->allocs, &t_links, (size_t)std::max<long long>(nl, 1))) ||
        (rc = dmalloc(&d->allocs, &t_arc, (size_t)std::max<long long>(nl, 1)))) return rc;
    K3_HIP_CHECK(hipMemcpy(t_state, pool0.tok_state, sizeof(int) * n, hipMemcpyDeviceToDevice));
    K3_HIP_CHECK(hipMemcpy(t_cost, pool0.tok_cost, sizeof(unsigned) * n, hipMemcpyDeviceToDevice));
    K3_HIP_CHECK(hipMemcpy(t_order, p.lt_order + (size_t)li.order_sel * p.frame_tokens_cap, sizeof(int) * n, hipMemcpyDeviceToDevice));
    K3_HIP_CHECK(hipMemcpy(t_by_ins, p.lt_by_ins, sizeof(int) * n, hipMemcpyDeviceToDevice));
    if (nl > 0) {
      K3_HIP_CHECK(hipMemcpy(t_links, pool0.links, sizeof(Link) * nl, hipMemcpyDeviceToDevice));
      K3_HIP_CHECK(hipMemcpy(t_arc, pool0.link_arc, sizeof(int) * nl, hipMemcpyDeviceToDevice));
    }
    p.tpl_state = t_state; p.tpl_cost = t_cost; p.tpl_links = t_links; p.tpl_arc = t_arc; p.tpl_order = t_order; p.tpl_by_ins = t_by_ins;
    p.tpl_n = (int)n; p.tpl_nl = (int)nl; p.tpl_eps = li.n_eps;
  }
  if (getenv("K3_DEBUG_QUEUE")) fprintf(stderr, "k3 InitDecoding template: status %d, %lld tokens, %lld links, cur_base %lld n_cur %d -> %s\n", li.status, n, nl, li.cur_base, li.n_cur,
      p.tpl_n > 0 ? "kept" : "none");
  // the lane as it was: no utterance yet
  K3_HIP_CHECK(hipMemset(p.info, 0, sizeof(LaneInfo)));
  d->last_utts = 0; d->last_frames.clear(); d->fresh.clear(); d->lane_final.clear(); d->sel.clear(); d->started = false; d->finalized = false; d->info_valid = false;
  return K3_OK;
}

// The FIRST FRAME's structure (DecParams::t0_*): lane 0 decodes one frame of all-zero log-likelihoods from the InitDecoding template with the capture build of the
// token-passing kernel, which leaves the replay's records in the lane's scratch; they, the frame's tokens and its links (as source / destination / arc) become the template.
// Only when frame 0 is structurally the same for every utterance: the tokens after InitDecoding do not exceed min_active (adaptive beam +inf).
static int build_frame0_template(k3_decoder *d) {
  DecParams &p = d->p; int rc;
  if (p.tpl_n <= 0 || p.literal != 1 || p.tpl_n > p.min_active || p.tpl_n > p.max_active) return K3_OK;
  long long *cap = nullptr; float *zero_row = nullptr;
  if ((rc = dmalloc(&d->allocs, &cap, 16)) || (rc = dmalloc(&d->allocs, &zero_row, (size_t)d->num_pdfs))) return rc;
  K3_HIP_CHECK(hipMemset(cap, 0, 16 * sizeof(long long))); K3_HIP_CHECK(hipMemset(zero_row, 0, sizeof(float) * d->num_pdfs));
  p.cap = cap; d->capture_launch = true;
  const int64_t ro[2] = {0, 1};
  rc = k3_decoder_init_decoding(d, 1, 2, nullptr);
  if (!rc) rc = k3_decoder_advance_decoding(d, 1, zero_row, d->num_pdfs, ro, nullptr);
  d->capture_launch = false; p.cap = nullptr;
  if (rc) return rc;
  K3_HIP_CHECK(hipDeviceSynchronize());
  long long hc[16]; LaneInfo li; LanePool pool0;
  K3_HIP_CHECK(hipMemcpy(hc, cap, sizeof hc, hipMemcpyDeviceToHost));
  K3_HIP_CHECK(hipMemcpy(&li, p.info, sizeof(li), hipMemcpyDeviceToHost));
  K3_HIP_CHECK(hipMemcpy(&pool0, p.pools, sizeof(pool0), hipMemcpyDeviceToHost));
  const long long n_e = hc[0], n = hc[1], m_e = hc[2], ncid = hc[3], narc = hc[4], niq = hc[5], nwork = hc[6], created = hc[7], nb = p.tpl_n, l0 = p.tpl_nl;
  const size_t cap_tok = (size_t)p.frame_tokens_cap;
  bool ok = li.status == kStOk && li.num_frames == 1 && li.cur_base == nb && li.n_cur == n && n > 0 && n <= 65535 && p.tpl_n <= 8191 && n_e + created == n && ncid > 0 && nwork > 0 && niq > 0;
  std::vector<long long> off(3, 0), le(2, 0), ln(2, 0);
  if (ok) {
    K3_HIP_CHECK(hipMemcpy(off.data(), p.tok_off, sizeof(long long) * 3, hipMemcpyDeviceToHost));
    K3_HIP_CHECK(hipMemcpy(le.data(), p.link_off_e, sizeof(long long) * 2, hipMemcpyDeviceToHost));
    K3_HIP_CHECK(hipMemcpy(ln.data(), p.link_off_n, sizeof(long long) * 2, hipMemcpyDeviceToHost));
    ok = off[1] == nb && off[2] == nb + n && le[0] == l0 && ln[1] >= l0 && le[1] == li.n_links && le[1] >= ln[1];
  }
  if (ok) {
    const long long nle = ln[1] - l0, nx_all = le[1] - ln[1];
    std::vector<Link> hl((size_t)(nle + nx_all)); std::vector<int> ha((size_t)(nle + nx_all)); std::vector<unsigned> hcst((size_t)(nb + n)); std::vector<int> hst((size_t)n), hrank((size_t)n);
    K3_HIP_CHECK(hipMemcpy(hl.data(), pool0.links + l0, sizeof(Link) * hl.size(), hipMemcpyDeviceToHost));
    K3_HIP_CHECK(hipMemcpy(ha.data(), pool0.link_arc + l0, sizeof(int) * ha.size(), hipMemcpyDeviceToHost));
    K3_HIP_CHECK(hipMemcpy(hcst.data(), pool0.tok_cost, sizeof(unsigned) * hcst.size(), hipMemcpyDeviceToHost));
    K3_HIP_CHECK(hipMemcpy(hst.data(), pool0.tok_state + nb, sizeof(int) * n, hipMemcpyDeviceToHost));
    K3_HIP_CHECK(hipMemcpy(hrank.data(), p.lt_label, sizeof(int) * n, hipMemcpyDeviceToHost));      // (left in place by the capture build: dense creation ranks)
    std::vector<ArcRec> harcs((size_t)d->fst->num_arcs);
    K3_HIP_CHECK(hipMemcpy(harcs.data(), d->fst->arcs, sizeof(ArcRec) * harcs.size(), hipMemcpyDeviceToHost));
    std::vector<int4> el, xl;
    for (long long l = 0; l < nle && ok; l++) {
      const Link &k = hl[(size_t)l]; const int a = ha[(size_t)l];
      if (k.src >= (unsigned)nb || k.dst < (unsigned)nb || k.dst >= (unsigned)(nb + n) || a < 0 || a >= d->fst->num_arcs) { ok = false; break; }
      el.push_back(make_int4((int)((k.src << 16) | (k.dst - (unsigned)nb)), a, harcs[(size_t)a].pdf, __builtin_bit_cast(int, harcs[(size_t)a].w)));
    }
    for (long long l = nle; l < nle + nx_all && ok; l++) {      // the closure's links that are live at their source's final cost
      const Link &k = hl[(size_t)l]; const int a = ha[(size_t)l];
      if (k.src < (unsigned)nb || k.src >= (unsigned)(nb + n) || k.dst < (unsigned)nb || k.dst >= (unsigned)(nb + n) || a < 0 || a >= d->fst->num_arcs) { ok = false; break; }
      if (__builtin_bit_cast(unsigned, k.ac) != hcst[k.src]) continue;
      xl.push_back(make_int4((int)(((k.src - (unsigned)nb) << 16) | (k.dst - (unsigned)nb)), a, 0, __builtin_bit_cast(int, harcs[(size_t)a].w)));
    }
    ok = ok && (long long)el.size() == m_e && (long long)xl.size() == narc;      // every emitting arc examined is accepted; the live closure links ARE the sub-graph's arcs
    float best = std::numeric_limits<float>::infinity();
    for (long long i = 0; i < nb; i++) { const unsigned c = hcst[(size_t)i]; const float v = __builtin_bit_cast(float, (c & 0x80000000u) ? (c ^ 0x80000000u) : ~c); best = v < best ? v : best; }
    if (ok) {
      int4 *t_el = nullptr, *t_xl = nullptr, *t_meta = nullptr, *t_wrec = nullptr; int2 *t_ar = nullptr, *t_rlist = nullptr; int *t_state = nullptr, *t_rank = nullptr, *t_c2t = nullptr;
      if ((rc = dmalloc(&d->allocs, &t_el, el.size())) || (rc = dmalloc(&d->allocs, &t_xl, std::max<size_t>(xl.size(), 1))) || (rc = dmalloc(&d->allocs, &t_meta, (size_t)ncid)) ||
          (rc = dmalloc(&d->allocs, &t_wrec, (size_t)(2 * nwork))) || (rc = dmalloc(&d->allocs, &t_ar, (size_t)std::max<long long>(narc, 1))) || (rc = dmalloc(&d->allocs, &t_rlist, (size_t)niq)) ||
          (rc = dmalloc(&d->allocs, &t_state, (size_t)n)) || (rc = dmalloc(&d->allocs, &t_rank, (size_t)n)) || (rc = dmalloc(&d->allocs, &t_c2t, (size_t)ncid))) return rc;
      K3_HIP_CHECK(hipMemcpy(t_el, el.data(), sizeof(int4) * el.size(), hipMemcpyHostToDevice));
      if (!xl.empty()) K3_HIP_CHECK(hipMemcpy(t_xl, xl.data(), sizeof(int4) * xl.size(), hipMemcpyHostToDevice));
      K3_HIP_CHECK(hipMemcpy(t_state, hst.data(), sizeof(int) * n, hipMemcpyHostToDevice));
      K3_HIP_CHECK(hipMemcpy(t_rank, hrank.data(), sizeof(int) * n, hipMemcpyHostToDevice));
      K3_HIP_CHECK(hipMemcpy(t_c2t, p.lt_c2t, sizeof(int) * ncid, hipMemcpyDeviceToDevice));
      K3_HIP_CHECK(hipMemcpy(t_meta, p.lt_meta, sizeof(int4) * ncid, hipMemcpyDeviceToDevice));
      if (narc > 0) K3_HIP_CHECK(hipMemcpy(t_ar, p.lt_arcs2, sizeof(int2) * narc, hipMemcpyDeviceToDevice));
      K3_HIP_CHECK(hipMemcpy(t_rlist, p.lt_coffs, sizeof(int2) * niq, hipMemcpyDeviceToDevice));      // (the capture build's copy of q.rlist: the frame's last pass reuses q.rlist)
      K3_HIP_CHECK(hipMemcpy(t_wrec, p.lt_wrec, sizeof(int4) * 2 * nwork, hipMemcpyDeviceToDevice));
      p.t0_elinks = t_el; p.t0_xlinks = t_xl; p.t0_state = t_state; p.t0_rank1 = t_rank; p.t0_c2t = t_c2t; p.t0_meta = t_meta; p.t0_ar = t_ar; p.t0_rlist = t_rlist; p.t0_wrec = t_wrec;
      p.t0_n_e = (int)n_e; p.t0_m_e = (int)m_e; p.t0_nle = (int)el.size(); p.t0_nlx = (int)xl.size(); p.t0_ncid = (int)ncid; p.t0_narc = (int)narc; p.t0_niq = (int)niq;
      p.t0_nworkers = (int)nwork; p.t0_hash_size = (unsigned)hc[8]; p.t0_best = best; p.t0_eps = li.n_eps - p.tpl_eps;
      if (const char *e = getenv("K3_T0_STAGE")) p.t0_pad = atoi(e);      // (developer: the first-frame kernel stops behind stage N; results are then garbage)
      p.t0_n = (int)n;      // (last: the template is in use from here on)
    }
    (void)cap_tok;
  }
  if (getenv("K3_DEBUG_QUEUE") && p.t0_n > 0) {
    std::vector<int2> a(8), b(8); std::vector<int4> w(8);
    (void)hipMemcpy(a.data(), p.lt_coffs, sizeof(int2) * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), p.lt_rlist, sizeof(int2) * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(w.data(), p.lt_wrec, sizeof(int4) * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; i++) fprintf(stderr, "  rlist(coffs)[%d] = (%d, %d)   q.rlist[%d] = (%d, %d)   wrec[%d] = (%d %d %d %d)\n", i, a[i].x, a[i].y, i, b[i].x, b[i].y, i, w[i].x, w[i].y, w[i].z, w[i].w);
  }
  if (getenv("K3_DEBUG_QUEUE")) fprintf(stderr, "k3 first-frame template: %s (%lld emitting + %lld closure tokens, %lld emitting arcs, closure %lld ids / %lld arcs / %lld roots / %lld components)\n",
      p.t0_n > 0 ? "kept" : "none", n_e, n - n_e, m_e, ncid, narc, niq, nwork);
  // the lane as it was: idle creation labels (the capture build left the frame's in place), no utterance
  K3_HIP_CHECK(hipMemset(p.lt_label, 0xFF, cap_tok * sizeof(unsigned)));
  K3_HIP_CHECK(hipMemset(p.info, 0, sizeof(LaneInfo)));
  d->last_utts = 0; d->last_frames.clear(); d->fresh.clear(); d->lane_final.clear(); d->sel.clear(); d->started = false; d->finalized = false; d->info_valid = false;
  return K3_OK;
}

extern "C" int k3_decoder_create(const k3_fst *fst, const k3_decoder_config *cfg, int32_t nlanes, int32_t num_pdfs, k3_decoder **out) {
  K3_REQUIRE(fst && cfg && out && nlanes > 0 && num_pdfs > 0, "k3_decoder_create: bad argument");
  if (fst->max_pdf >= num_pdfs) {
    k3::set_error("k3_decoder_create: the graph reads pdf %d but the log-likelihood matrix has %d columns (graph / model mismatch)", fst->max_pdf, num_pdfs);
    return K3_ERR_ARG;
  }
  K3_REQUIRE(cfg->beam > 0 && cfg->lattice_beam > 0 && cfg->max_active > 1 && cfg->min_active >= 0 && cfg->min_active < cfg->max_active,
      "k3_decoder_create: bad beam / active limits");
  K3_REQUIRE(cfg->frame_tokens_cap >= 64 && cfg->frame_cands_cap >= cfg->frame_tokens_cap && cfg->lane_tokens_cap >= cfg->frame_tokens_cap && cfg->lane_links_cap > 0 &&
             cfg->lane_tokens_cap < (1ll << 31) && cfg->lane_links_cap < (1ll << 40),
                 "k3_decoder_create: bad capacities (need frame_cands_cap >= frame_tokens_cap, lane_tokens_cap < 2^31)");
  std::unique_ptr<k3_decoder> d(new k3_decoder());
  d->fst = fst; d->cfg = *cfg; d->nlanes = nlanes; d->num_pdfs = num_pdfs;
  DecParams &p = d->p;
  p.offs = fst->offs; p.arcs = fst->arcs; p.final_cost = fst->final_cost; p.arc_ilabel = fst->arc_ilabel; p.start = fst->start;
  p.dinfo = nullptr;
  if (cfg->literal_order) { const int rc_ = fst_dinfo(fst, &p.dinfo); if (rc_) return rc_; }
  p.beam = cfg->beam; p.lattice_beam = cfg->lattice_beam; p.beam_delta = cfg->beam_delta; p.max_active = cfg->max_active; p.min_active = cfg->min_active;
  p.frame_tokens_cap = cfg->frame_tokens_cap; p.frame_cands_cap = cfg->frame_cands_cap;
  int hs = 1; while (hs < 2 * cfg->frame_tokens_cap) hs <<= 1;
  p.hash_mask = hs - 1; p.num_pdfs = num_pdfs;
  p.use_lds_row = (K3_DEC_LDSROW && (size_t)num_pdfs * sizeof(float) <= 25 * 1024) ? 1 : 0;   // keeps a lane under 80 KB of LDS: two lanes per CU
  const size_t nl = (size_t)nlanes;
  int rc;
  {      // the lanes' token / link pools: lane l starts on slice l of one allocation per array (the reservation); pools[l] says where it lives now
    int *tok_state, *link_arc, *newidx; unsigned *tok_cost; float *tok_extra; Link *links;
    if ((rc = dmalloc(&d->allocs, &tok_state, nl * cfg->lane_tokens_cap))) return rc;
    if ((rc = dmalloc(&d->allocs, &tok_cost, nl * cfg->lane_tokens_cap))) return rc;
    if ((rc = dmalloc(&d->allocs, &tok_extra, nl * cfg->lane_tokens_cap))) return rc;
    if ((rc = dmalloc(&d->allocs, &newidx, nl * cfg->lane_tokens_cap))) return rc;
    if ((rc = dmalloc(&d->allocs, &links, nl * cfg->lane_links_cap))) return rc;
    if ((rc = dmalloc(&d->allocs, &link_arc, nl * cfg->lane_links_cap))) return rc;
    std::vector<LanePool> pools(nl);
    for (size_t l = 0; l < nl; l++) pools[l] = LanePool{
      tok_state + l * cfg->lane_tokens_cap, tok_cost + l * cfg->lane_tokens_cap, tok_extra + l * cfg->lane_tokens_cap, newidx + l * cfg->lane_tokens_cap,
                                                        links + l * cfg->lane_links_cap, link_arc + l * cfg->lane_links_cap, (long long)cfg->lane_tokens_cap,
                                                            (long long)cfg->lane_links_cap
                                                        };
    if ((rc = dmalloc(&d->allocs, &p.pools, nl))) return rc;
    K3_HIP_CHECK(hipMemcpy(p.pools, pools.data(), nl * sizeof(LanePool), hipMemcpyHostToDevice));
    // spare arena for the lanes that outgrow the reservation (grow_lane_pools): -1 = a quarter of the reservation, but at least 1 GiB and 14 times a lane's reservation
    // (one lane growing 2x, 4x, 8x: an utterance eight times as long as the reservation was sized for -- the reference's host vectors grow without a limit; HBM is 288 GB)
    const long long lane_bytes = 16ll * cfg->lane_tokens_cap + 20ll * cfg->lane_links_cap + 6 * 256;
    p.spare_bytes = cfg->spare_pool_bytes >= 0 ? cfg->spare_pool_bytes : std::max<long long>({(long long)nl * lane_bytes / 4, 14 * lane_bytes, 1ll << 30});
    p.spare = nullptr;
    // (a DEFAULT-sized arena that does not fit beside the reservation is halved until it does -- on a smaller part, or with several decoder objects, the decoder still
    // comes up and only utterances that outgrow their reservation by more than what is left see K3_ERR_OVERFLOW; an arena the caller asked for explicitly must fit)
    while (p.spare_bytes > 0 && (rc = dmalloc(&d->allocs, &p.spare, (size_t)p.spare_bytes))) {
      if (cfg->spare_pool_bytes >= 0) return rc;
      (void)hipGetLastError(); p.spare = nullptr; p.spare_bytes = p.spare_bytes >= (64ll << 20) ? p.spare_bytes / 2 : 0;
    }
    if ((rc = dmalloc(&d->allocs, &p.spare_used, 1))) return rc;
    K3_HIP_CHECK(hipMemset(p.spare_used, 0, sizeof(unsigned long long)));
  }
  if ((rc = dmalloc(&d->allocs, &p.hash, nl * hs))) return rc;
  if ((rc = dmalloc(&d->allocs, &p.tok_slot, nl * cfg->frame_tokens_cap))) return rc;
  if ((rc = dmalloc(&d->allocs, &p.wl, 2 * nl * cfg->frame_tokens_cap))) return rc;
  if ((rc = dmalloc(&d->allocs, &p.c_tot, nl * cfg->frame_cands_cap))) return rc;
  if ((rc = dmalloc(&d->allocs, &p.c_ac, nl * cfg->frame_cands_cap))) return rc;
  if ((rc = dmalloc(&d->allocs, &p.c_dst, nl * cfg->frame_cands_cap))) return rc;
  if ((rc = dmalloc(&d->allocs, &p.c_arc, nl * cfg->frame_cands_cap))) return rc;
  if ((rc = dmalloc(&d->allocs, &p.c_src, nl * cfg->frame_cands_cap))) return rc;
  if ((rc = dmalloc(&d->allocs, &p.info, nl))) return rc;
  K3_HIP_CHECK(hipMemset(p.info, 0, nl * sizeof(LaneInfo)));      // status 0: the kernel's fresh path reads the lane's previous status
  if ((rc = dmalloc(&d->allocs, &p.prof, nl * 16))) return rc;
  K3_HIP_CHECK(hipMemset(p.prof, 0, nl * 16 * sizeof(long long)));
  if ((rc = dmalloc(&d->allocs, &d->d_row_off, nl + 1))) return rc;
  if ((rc = dmalloc(&d->allocs, &d->d_fresh, nl))) return rc;
  d->arg_off_rows = align_up(sizeof(long long) * (nl + 1), 16);
  d->arg_off_fresh = d->arg_off_rows + align_up(sizeof(float *) * nl, 16);
  d->arg_off_queue = d->arg_off_fresh + align_up(sizeof(int) * nl, 16);      // {head counter, pad x 3, lanes[nl]}
  d->arg_bytes = d->arg_off_queue + 16 + align_up(sizeof(int) * nl, 16);
  d->lit_resident = cfg->resident_lanes > 0 ? cfg->resident_lanes : 0; d->lit_exclusive = cfg->resident_exclusive != 0;
  if (const char *e = getenv("K3_LIT_RESIDENT")) d->lit_resident = atoi(e);      // (developer override for A/B runs)
  if (const char *e = getenv("K3_LIT_EXCLUSIVE")) d->lit_exclusive = atoi(e) != 0;
  for (k3_decoder::ArgSlot &a : d->arg) {
    K3_HIP_CHECK(hipHostMalloc((void **)&a.h, d->arg_bytes, hipHostMallocDefault)); K3_HIP_CHECK(hipMalloc((void **)&a.d, d->arg_bytes));
    K3_HIP_CHECK(hipEventCreateWithFlags(&a.ev, hipEventDisableTiming));
  }
  if ((rc = dmalloc(&d->allocs, &d->d_lane_ids, nl))) return rc;
  p.live_cap = (int)std::min<long long>(cfg->lane_tokens_cap, 1 << 18);
  if ((rc = dmalloc(&d->allocs, &p.live_tok, nl * p.live_cap))) return rc;
  if ((rc = dmalloc(&d->allocs, &p.live_link, nl * p.live_cap))) return rc;
  // 2: the closure's creation order by the one-wavefront replay (the fall-back of 1); 3: 1 with zero-length component stacks (exercises the fall-back)
  p.literal = (cfg->literal_order == 2 || cfg->literal_order == 3) ? cfg->literal_order : (cfg->literal_order ? 1 : 0);
  p.lit_force_hbm_order = cfg->literal_order == 4;
  p.hash_ratio = cfg->hash_ratio;
  p.fast_cap = p.literal == 1 ? (cfg->fast_frame_tokens < 0 ? k3_lit_fast_tokens() : std::min(cfg->fast_frame_tokens, k3_lit_fast_tokens())) : 0;
  if (p.literal) {
    K3_REQUIRE(cfg->hash_ratio > 0.0f && cfg->hash_ratio <= 64.0f, "k3_decoder_create: literal_order needs 0 < hash_ratio <= 64 (LatticeFasterDecoderConfig::hash_ratio)");
    K3_REQUIRE(cfg->frame_tokens_cap <= 65536 && cfg->frame_cands_cap > cfg->frame_tokens_cap,
        "k3_decoder_create: literal_order needs frame_tokens_cap <= 65536 < frame_cands_cap");
    const size_t cap = (size_t)cfg->frame_tokens_cap, nch = cap / 64 + 2;
    p.hash_cap = (int)std::max<double>(1000.0, std::ceil((double)cap * cfg->hash_ratio) + 1.0);
    p.seq_words_cap = (int)((8 * (size_t)cfg->frame_cands_cap + cap) / 32 + 64);      // labels: emitting arcs expanded on a frame + tokens its closure creates
    p.eps_cap = cfg->frame_cands_cap; p.stack_cap = 4 * cfg->frame_tokens_cap;
    // one arena, LANE-major: a lane's scratch arrays are neighbours in memory (a workgroup touches the first few KB of every one of them on
    // every frame; as separate allocations that was ~45 distant pages per lane, and the address translation dominated the memory latency)
    size_t off = 0;
    auto place = [&](auto **ptr, size_t count) {
      using T = std::remove_pointer_t<std::remove_pointer_t<decltype(ptr)>>;
      *ptr = reinterpret_cast<T *>(off);
      off += (count * sizeof(T) + 255) & ~(size_t)255;
    };
    place(&p.lt_order, 2 * cap); place(&p.lt_label, cap); place(&p.lt_c0, cap); place(&p.lt_rflag, cap); place(&p.lt_rown, cap); place(&p.lt_grp, cap); place(&p.lt_lead, cap + 1);
    place(&p.lt_crng, cap); place(&p.lt_c2t, cap); place(&p.lt_iq, cap); place(&p.lt_dense, cap); place(&p.lt_by_ins, cap); place(&p.lt_meta, cap); place(&p.lt_rcost, cap);
    place(&p.lt_par, cap);
    place(&p.lt_rtmp, cap);
    place(&p.lt_rlist, cap);
    place(&p.lt_wrec, 2 * cap);
    place(&p.lt_cinfo, cap);
    place(&p.lt_coffs, cap);
    place(&p.lt_rinfo, cap);
    place(&p.lt_vis, cap);
    {
      size_t ts = 1024;
      while (ts < 2 * cap) ts <<= 1;
      place(&p.lt_btab, ts);
    }
    place(&p.lt_cmin, 2 * nch); place(&p.lt_ccnt, 2 * nch); place(&p.lt_cdst, (size_t)p.eps_cap); place(&p.lt_cw, (size_t)p.eps_cap); place(&p.lt_arcs2, (size_t)p.eps_cap);
    place(&p.lt_stack, (size_t)p.stack_cap); place(&p.lt_bm, (size_t)p.seq_words_cap); place(&p.lt_wpre, (size_t)p.seq_words_cap);
    p.lt_lane_bytes = (long long)((off + 4095) & ~(size_t)4095);
    char *arena = nullptr;
    if ((rc = dmalloc(&d->allocs, &arena, nl * (size_t)p.lt_lane_bytes))) return rc;
    auto rebase = [&](auto **ptr) { using T = std::remove_pointer_t<std::remove_pointer_t<decltype(ptr)>>; *ptr = reinterpret_cast<T *>(arena + reinterpret_cast<size_t>(*ptr)); };
    rebase(&p.lt_order);
    rebase(&p.lt_label);
    rebase(&p.lt_c0);
    rebase(&p.lt_rflag);
    rebase(&p.lt_rown);
    rebase(&p.lt_grp);
    rebase(&p.lt_lead);
    rebase(&p.lt_crng);
    rebase(&p.lt_c2t);
    rebase(&p.lt_iq);
    rebase(&p.lt_dense);
    rebase(&p.lt_by_ins);
    rebase(&p.lt_meta);
    rebase(&p.lt_rcost);
    rebase(&p.lt_par);
    rebase(&p.lt_rtmp);
    rebase(&p.lt_rlist);
    rebase(&p.lt_wrec);
    rebase(&p.lt_cinfo);
    rebase(&p.lt_coffs);
    rebase(&p.lt_rinfo);
    rebase(&p.lt_vis);
    rebase(&p.lt_btab);
    rebase(&p.lt_cmin);
    rebase(&p.lt_ccnt);
    rebase(&p.lt_cdst);
    rebase(&p.lt_cw);
    rebase(&p.lt_arcs2);
    rebase(&p.lt_stack);
    rebase(&p.lt_bm);
    rebase(&p.lt_wpre);
    // idle patterns of the scratch: labels / bucket firsts all ones, bitmap / bucket counters zero
    K3_HIP_CHECK(hipMemset2D(p.lt_label, (size_t)p.lt_lane_bytes, 0xFF, cap * sizeof(unsigned), nl));
    K3_HIP_CHECK(hipMemset2D(p.lt_bm, (size_t)p.lt_lane_bytes, 0, (size_t)p.seq_words_cap * sizeof(unsigned), nl));
    K3_REQUIRE(k3_lit_forward_prepare() == 0, "k3_decoder_create: the literal_order kernel could not be configured");
    if ((rc = dmalloc(&d->allocs, &p.row_skip, nl))) return rc;
    K3_HIP_CHECK(hipMemset(p.row_skip, 0, nl * sizeof(int)));
  }
  // empty table: key = -1, cost = max, tok = -1, stamp = 0
  std::vector<Slot> init((size_t)hs, Slot{kEmpty, kEncMax, -1, 0});
  for (int l = 0; l < nlanes; l++) K3_HIP_CHECK(hipMemcpy(p.hash + (size_t)l * hs, init.data(), sizeof(Slot) * hs, hipMemcpyHostToDevice));
  K3_HIP_CHECK(hipFuncSetAttribute((const void *)k3_decode_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  if (p.literal && !getenv("K3_LIT_NO_INIT_TEMPLATE")) { const int rc_ = build_init_template(d.get()); if (rc_) return rc_; }
  if (p.literal && !getenv("K3_LIT_NO_INIT_TEMPLATE") && !getenv("K3_LIT_NO_FRAME0_TEMPLATE")) { const int rc_ = build_frame0_template(d.get()); if (rc_) return rc_; }
  *out = d.release();
  return K3_OK;
}

extern "C" void k3_decoder_destroy(k3_decoder *d) { delete d; }

extern "C" int k3_decoder_set_profiling(k3_decoder *d, int32_t on) {
  K3_REQUIRE(d, "k3_decoder_set_profiling: null argument");
  if (on) for (hipEvent_t &e : d->ev) if (!e) K3_HIP_CHECK(hipEventCreate(&e));
  d->profiling = on != 0;
  return K3_OK;
}
extern "C" int k3_decoder_kernel_times(k3_decoder *d, float *h_ms) {
  K3_REQUIRE(d && h_ms && d->profiling && d->last_utts > 0, "k3_decoder_kernel_times: profiling is off or nothing was decoded");
  K3_HIP_CHECK(hipEventSynchronize(d->ev[2]));
  K3_HIP_CHECK(hipEventElapsedTime(&h_ms[0], d->ev[0], d->ev[1]));
  K3_HIP_CHECK(hipEventElapsedTime(&h_ms[1], d->ev[1], d->ev[2]));
  return K3_OK;
}

static int ensure_frame_capacity(k3_decoder *d, int max_frames, hipStream_t st) {
  DecParams &p = d->p;
  if (max_frames + 2 <= d->fstride) return K3_OK;
  K3_HIP_CHECK(hipStreamSynchronize(st));
  for (void *q : d->frame_allocs) (void)hipFree(q);
  d->frame_allocs.clear();
  d->fstride = max_frames + 2; const size_t n = (size_t)d->nlanes * d->fstride; int rc;
  if ((rc = dmalloc(&d->frame_allocs, &p.tok_off, n))) return rc;
  if ((rc = dmalloc(&d->frame_allocs, &p.link_off_e, n))) return rc;
  if ((rc = dmalloc(&d->frame_allocs, &p.link_off_n, n))) return rc;
  if ((rc = dmalloc(&d->frame_allocs, &p.st_ntoks, n))) return rc;
  if ((rc = dmalloc(&d->frame_allocs, &p.st_cur, n))) return rc;
  if ((rc = dmalloc(&d->frame_allocs, &p.st_ab, n))) return rc;
  if ((rc = dmalloc(&d->frame_allocs, &p.st_next, n))) return rc;
  if ((rc = dmalloc(&d->frame_allocs, &p.st_co, n))) return rc;
  p.fstride = d->fstride;
  return K3_OK;
}

// InitDecoding for `num_utts` lanes (cuda-decoder.h:248 InitDecoding(channels)): the next k3_decoder_advance_decoding starts new utterances.
// max_total_frames bounds the frames any lane will be advanced by before k3_decoder_finalize_decoding.
extern "C" int k3_decoder_init_decoding(k3_decoder *d, int32_t num_utts, int32_t max_total_frames, void *stream) {
  K3_REQUIRE(d && num_utts > 0 && num_utts <= d->nlanes && max_total_frames > 0, "k3_decoder_init_decoding: bad argument");
  { const int rc = ensure_frame_capacity(d, max_total_frames, (hipStream_t)stream); if (rc) return rc; }
  d->last_utts = num_utts; d->last_frames.assign(num_utts, 0); d->fresh.assign(num_utts, 1); d->lane_final.assign(num_utts, 0); d->sel.clear();
  d->started = false; d->finalized = false; d->info_valid = false; d->last_stream = (hipStream_t)stream;
  return K3_OK;
}

// CudaDecoder::InitDecoding(channels) (cuda-decoder.h:248): restart the listed lanes of the current group; the other lanes keep decoding.
extern "C" int k3_decoder_init_channels(k3_decoder *d, const int32_t *channels, int32_t n, void *stream) {
  K3_REQUIRE(d && channels && n >= 0 && d->last_utts > 0, "k3_decoder_init_channels: call k3_decoder_init_decoding for the lane group first");
  for (int i = 0; i < n; i++) K3_REQUIRE(channels[i] >= 0 && channels[i] < d->last_utts, "k3_decoder_init_channels: channel out of range");
  for (int i = 0; i < n; i++) { d->last_frames[channels[i]] = 0; d->fresh[channels[i]] = 1; d->lane_final[channels[i]] = 0; }
  d->finalized = false; d->info_valid = false; (void)stream;
  return K3_OK;
}

// The literal_order kernel's lane list for this call (slot.h already holds the call's row offsets): the lanes with frames (or a fresh start), longest first.  Workgroup b of the
// launch decodes entry b: a CU's two workgroups (b, b + 256 of a 512-lane launch) then hold a long and a short utterance, and a batch of unequal lengths no longer pairs two long
// ones on a CU while another CU idles (the reference reschedules lanes per chunk for the same reason: cuda-online-pipeline-dynamic-batcher.cc).  With a work-queue build
// (k3_lit_has_queue) and resident_lanes < lanes, fewer workgroups take the entries through the head counter.  Returns the number of workgroups to launch.
// The literal_order launch of a call (slot.h holds the call's row offsets and pending-InitDecoding flags).  The two template kernels in front of the token-passing kernel have
// something to do only for a lane that starts in this call or that holds just its InitDecoding tokens and gets its first frame now; in every other call -- all but one of the ~20
// chunk calls of a streamed utterance -- they are not launched: each is one more launch that has to find 79 KB of LDS per workgroup on CUs other streams are using, and the two of
// them cost the online program 8 - 10 % (24.2 -> 26.2 k x RT at 51-frame chunks).  row_skip, which the first-frame kernel resets per call, is then cleared by a fill if it may hold
// a 1 from an earlier call.
static int launch_literal(k3_decoder *d, const k3_decoder::ArgSlot &slot, int lit_grid, hipStream_t st) {
  DecParams &p = d->p; const int U = d->last_utts;
  const long long *ro = reinterpret_cast<const long long *>(slot.h); const int *fresh = reinterpret_cast<const int *>(slot.h + d->arg_off_fresh);
  bool need = false;
  for (int u = 0; u < U && !need; u++) need = fresh[u] != 0 || (ro[u + 1] > ro[u] && d->last_frames[u] == 0);
  if (need) d->row_skip_dirty = true;
  else if (d->row_skip_dirty && p.row_skip) { K3_HIP_CHECK(hipMemsetAsync(p.row_skip, 0, sizeof(int) * (size_t)d->nlanes, st)); d->row_skip_dirty = false; }
  k3_lit_forward_launch(&p, sizeof(p), lit_grid, (d->lit_exclusive ? 1 : 0) | (need ? 0 : 2), st);
  return K3_OK;
}

static int fill_lane_queue(k3_decoder *d, k3_decoder::ArgSlot &slot, int U) {
  DecParams &p = d->p;
  p.q_head = nullptr; p.q_lanes = nullptr; p.q_n = 0;
  if (!p.literal) return U;
  const long long *ro = reinterpret_cast<const long long *>(slot.h); const int *fresh = reinterpret_cast<const int *>(slot.h + d->arg_off_fresh);
  int *head = reinterpret_cast<int *>(slot.h + d->arg_off_queue), *lanes = head + 4; int n = 0;
  bool sorted = true;
  for (int u = 0; u < U; u++) if (ro[u + 1] > ro[u] || fresh[u]) { if (n && ro[u + 1] - ro[u] > ro[lanes[n - 1] + 1] - ro[lanes[n - 1]]) sorted = false; lanes[n++] = u; }
  if (n == 0) return U;      // (nothing to do: every workgroup of the plain launch returns at once)
  if (!sorted) std::stable_sort(lanes, lanes + n, [ro](int a, int b) { return ro[a + 1] - ro[a] > ro[b + 1] - ro[b]; });
  head[0] = 0;
  if (getenv("K3_DEBUG_QUEUE")) fprintf(stderr, "k3 lane list: %d of %d lanes, sorted on entry %d, first %d (%lld frames) last %d (%lld frames)\n", n, U, (int)sorted, lanes[0],
      (long long)(ro[lanes[0] + 1] - ro[lanes[0]]), lanes[n - 1], (long long)(ro[lanes[n - 1] + 1] - ro[lanes[n - 1]]));
  p.q_lanes = reinterpret_cast<int *>(slot.d + d->arg_off_queue) + 4; p.q_n = n;
  if (k3_lit_has_queue() && d->lit_resident > 0 && d->lit_resident < n) { p.q_head = reinterpret_cast<int *>(slot.d + d->arg_off_queue); return d->lit_resident; }
  return n;
}

// AdvanceDecoding (cuda-decoder.h:262: AdvanceDecoding(lanes, loglikes)): lane u consumes rows h_row_offsets[u] .. [u+1] of d_loglikes as
// its NEXT frames (zero rows = the lane idles in this call).  Chunked calls give bit-identical results to one call with all frames.
extern "C" int k3_decoder_advance_decoding(k3_decoder *d, int32_t num_utts, const float *d_loglikes, int64_t ld, const int64_t *h_row_off, void *stream) {
  K3_REQUIRE(d && d_loglikes && h_row_off && num_utts == d->last_utts && ld >= d->num_pdfs,
      "k3_decoder_advance_decoding: bad argument (call k3_decoder_init_decoding for this many lanes first)");
  hipStream_t st = (hipStream_t)stream; DecParams &p = d->p;
  for (int u = 0; u < num_utts; u++) {
    const long long T = h_row_off[u + 1] - h_row_off[u];
    K3_REQUIRE(T >= 0 && d->last_frames[u] + T + 2 <= d->fstride, "k3_decoder_advance_decoding: more frames than max_total_frames of k3_decoder_init_decoding");
    K3_REQUIRE(T == 0 || !d->lane_final[u], "k3_decoder_advance_decoding: frames for a finalised lane (k3_decoder_init_channels restarts it)");
  }
  // (the decoder's host state -- frames consumed, pending InitDecoding flags, the argument ring's cursor -- changes only once the launch is queued: a HIP failure on the way
  // leaves it as it was, so the call can be repeated)
  k3_decoder::ArgSlot &slot = d->arg[d->arg_seq % k3_decoder::kArgSlots];
  if (slot.used) K3_HIP_CHECK(hipEventSynchronize(slot.ev));      // (the launch of kArgSlots calls ago: long finished)
  memcpy(slot.h, h_row_off, sizeof(long long) * (num_utts + 1)); memcpy(slot.h + d->arg_off_fresh, d->fresh.data(), sizeof(int) * num_utts);
  const int lit_grid = fill_lane_queue(d, slot, num_utts);
  K3_HIP_CHECK(hipMemcpyAsync(slot.d, slot.h, d->arg_bytes, hipMemcpyHostToDevice, st));
  p.loglikes = d_loglikes;
  p.ld = ld;
  p.row_off = reinterpret_cast<long long *>(slot.d);
  p.fresh = reinterpret_cast<int *>(slot.d + d->arg_off_fresh);
  p.lane_ids = nullptr;
  p.lane_rows = nullptr;
  const size_t lds = p.use_lds_row ? align_up((size_t)d->num_pdfs * sizeof(float), 16) : 16;
  if (d->profiling) K3_HIP_CHECK(hipEventRecord(d->ev[0], st));
  if (p.literal && d->capture_launch) { K3_REQUIRE(k3_lit_capture_launch(&p, sizeof(p), lit_grid, st) == 0, "k3_decoder: the capture launch failed"); }
  else if (p.literal) { const int rc_ = launch_literal(d, slot, lit_grid, st); if (rc_) return rc_; }
  else hipLaunchKernelGGL(k3_decode_forward_kernel, dim3(num_utts), dim3(kBlock), lds, st, p);
  K3_HIP_CHECK(hipGetLastError());
  if (d->profiling) K3_HIP_CHECK(hipEventRecord(d->ev[1], st));
  if (!d->ev_tp) K3_HIP_CHECK(hipEventCreateWithFlags(&d->ev_tp, hipEventDisableTiming));
  K3_HIP_CHECK(hipEventRecord(d->ev_tp, st)); d->ev_tp_recorded = true;
  K3_HIP_CHECK(hipEventRecord(slot.ev, st)); slot.used = true;
  for (int u = 0; u < num_utts; u++) d->last_frames[u] += (int)(h_row_off[u + 1] - h_row_off[u]);
  std::fill(d->fresh.begin(), d->fresh.end(), 0); d->arg_seq++;
  d->started = true; d->last_stream = st; d->info_valid = false;
  return K3_OK;
}

// AdvanceDecoding(lanes_assignements) of the reference (cuda-decoder.h:262): every listed channel gets a device pointer to the log-likelihoods of
// its next frame(s) -- num_frames rows, ld floats apart -- wherever they live; the other channels of the group idle.
extern "C" int k3_decoder_advance_decoding_lanes(k3_decoder *d, int32_t n, const int32_t *channels, const float *const *h_lane_frames, int32_t num_frames,
    int64_t ld, void *stream) {
  K3_REQUIRE(d && channels && h_lane_frames && n >= 0 && num_frames > 0 && ld >= d->num_pdfs && d->last_utts > 0,
      "k3_decoder_advance_decoding_lanes: bad argument (call k3_decoder_init_decoding first)");
  hipStream_t st = (hipStream_t)stream; DecParams &p = d->p; const int U = d->last_utts;
  std::vector<long long> ro(U + 1, 0); std::vector<const float *> rows(U, nullptr); std::vector<int> T(U, 0);
  for (int i = 0; i < n; i++) {
    const int c = channels[i];
    K3_REQUIRE(c >= 0 && c < U && h_lane_frames[i] && T[c] == 0, "k3_decoder_advance_decoding_lanes: channel out of range / listed twice / null frame pointer");
    K3_REQUIRE(d->last_frames[c] + num_frames + 2 <= d->fstride && !d->lane_final[c],
        "k3_decoder_advance_decoding_lanes: more frames than max_total_frames, or a finalised channel");
    T[c] = num_frames; rows[c] = h_lane_frames[i];
  }
  for (int u = 0; u < U; u++) ro[u + 1] = ro[u] + T[u];
  k3_decoder::ArgSlot &slot = d->arg[d->arg_seq % k3_decoder::kArgSlots];
  if (slot.used) K3_HIP_CHECK(hipEventSynchronize(slot.ev));
  memcpy(slot.h, ro.data(), sizeof(long long) * (U + 1));
  memcpy(slot.h + d->arg_off_rows, rows.data(), sizeof(float *) * U);
  memcpy(slot.h + d->arg_off_fresh, d->fresh.data(), sizeof(int) * U);
  const int lit_grid = fill_lane_queue(d, slot, U);
  K3_HIP_CHECK(hipMemcpyAsync(slot.d, slot.h, d->arg_bytes, hipMemcpyHostToDevice, st));
  p.loglikes = nullptr; p.ld = ld; p.row_off = reinterpret_cast<long long *>(slot.d); p.fresh = reinterpret_cast<int *>(slot.d + d->arg_off_fresh); p.lane_ids = nullptr;
  p.lane_rows = reinterpret_cast<const float **>(slot.d + d->arg_off_rows);
  const size_t lds = p.use_lds_row ? align_up((size_t)d->num_pdfs * sizeof(float), 16) : 16;
  if (d->profiling) K3_HIP_CHECK(hipEventRecord(d->ev[0], st));
  if (p.literal && d->capture_launch) { K3_REQUIRE(k3_lit_capture_launch(&p, sizeof(p), lit_grid, st) == 0, "k3_decoder: the capture launch failed"); }
  else if (p.literal) { const int rc_ = launch_literal(d, slot, lit_grid, st); if (rc_) return rc_; }
  else hipLaunchKernelGGL(k3_decode_forward_kernel, dim3(U), dim3(kBlock), lds, st, p);
  K3_HIP_CHECK(hipGetLastError());
  if (d->profiling) K3_HIP_CHECK(hipEventRecord(d->ev[1], st));
  if (!d->ev_tp) K3_HIP_CHECK(hipEventCreateWithFlags(&d->ev_tp, hipEventDisableTiming));
  K3_HIP_CHECK(hipEventRecord(d->ev_tp, st)); d->ev_tp_recorded = true;
  K3_HIP_CHECK(hipEventRecord(slot.ev, st)); slot.used = true;
  for (int u = 0; u < U; u++) d->last_frames[u] += T[u];
  std::fill(d->fresh.begin(), d->fresh.end(), 0); d->arg_seq++;
  d->started = true; d->last_stream = st; d->info_valid = false;
  return K3_OK;
}

// The same with a frame count per channel and the rows of a channel `ld` floats apart, wherever they lie: what the stateful network engine produces (k3_nnet_stream_forward:
// channel c's output row k at row k * num_channels + c, i.e. ld = num_channels * row length) is decoded where it is -- no gather of a pass's log-likelihoods into one block
// (round 5: that copy was 210 MB per pass of 512 channels, a sixth of the GPU work of a streaming round).  h_lane_first[u] = channel u's first row (null or h_num_frames[u] = 0:
// the channel idles in this call).
extern "C" int k3_decoder_advance_decoding_strided(k3_decoder *d, int32_t num_utts, const float *const *h_lane_first, const int32_t *h_num_frames, int64_t ld, void *stream) {
  K3_REQUIRE(d && h_lane_first && h_num_frames && num_utts == d->last_utts && ld >= d->num_pdfs,
      "k3_decoder_advance_decoding_strided: bad argument (call k3_decoder_init_decoding for this many lanes first)");
  hipStream_t st = (hipStream_t)stream; DecParams &p = d->p; const int U = d->last_utts;
  for (int u = 0; u < U; u++) {
    const int T = h_lane_first[u] ? h_num_frames[u] : 0;
    K3_REQUIRE(T >= 0 && d->last_frames[u] + T + 2 <= d->fstride, "k3_decoder_advance_decoding_strided: more frames than max_total_frames of k3_decoder_init_decoding");
    K3_REQUIRE(T == 0 || !d->lane_final[u], "k3_decoder_advance_decoding_strided: frames for a finalised lane (k3_decoder_init_channels restarts it)");
  }
  k3_decoder::ArgSlot &slot = d->arg[d->arg_seq % k3_decoder::kArgSlots];
  if (slot.used) K3_HIP_CHECK(hipEventSynchronize(slot.ev));
  long long *ro = reinterpret_cast<long long *>(slot.h); const float **rows = reinterpret_cast<const float **>(slot.h + d->arg_off_rows);
  ro[0] = 0;
  for (int u = 0; u < U; u++) { const int T = h_lane_first[u] ? h_num_frames[u] : 0; ro[u + 1] = ro[u] + T; rows[u] = h_lane_first[u]; }
  memcpy(slot.h + d->arg_off_fresh, d->fresh.data(), sizeof(int) * U);
  const int lit_grid = fill_lane_queue(d, slot, U);
  K3_HIP_CHECK(hipMemcpyAsync(slot.d, slot.h, d->arg_bytes, hipMemcpyHostToDevice, st));
  p.loglikes = nullptr; p.ld = ld; p.row_off = reinterpret_cast<long long *>(slot.d); p.fresh = reinterpret_cast<int *>(slot.d + d->arg_off_fresh); p.lane_ids = nullptr;
  p.lane_rows = reinterpret_cast<const float **>(slot.d + d->arg_off_rows);
  const size_t lds = p.use_lds_row ? align_up((size_t)d->num_pdfs * sizeof(float), 16) : 16;
  if (d->profiling) K3_HIP_CHECK(hipEventRecord(d->ev[0], st));
  if (p.literal && d->capture_launch) { K3_REQUIRE(k3_lit_capture_launch(&p, sizeof(p), lit_grid, st) == 0, "k3_decoder: the capture launch failed"); }
  else if (p.literal) { const int rc_ = launch_literal(d, slot, lit_grid, st); if (rc_) return rc_; }
  else hipLaunchKernelGGL(k3_decode_forward_kernel, dim3(U), dim3(kBlock), lds, st, p);
  K3_HIP_CHECK(hipGetLastError());
  if (d->profiling) K3_HIP_CHECK(hipEventRecord(d->ev[1], st));
  if (!d->ev_tp) K3_HIP_CHECK(hipEventCreateWithFlags(&d->ev_tp, hipEventDisableTiming));
  K3_HIP_CHECK(hipEventRecord(d->ev_tp, st)); d->ev_tp_recorded = true;
  K3_HIP_CHECK(hipEventRecord(slot.ev, st)); slot.used = true;
  for (int u = 0; u < U; u++) d->last_frames[u] += (int)(ro[u + 1] - ro[u]);
  std::fill(d->fresh.begin(), d->fresh.end(), 0); d->arg_seq++;
  d->started = true; d->last_stream = st; d->info_valid = false;
  return K3_OK;
}

// FinalizeDecoding (lattice-faster-decoder.cc:634-649) on the GPU: lattice-beam pruning with final-probs; lattices can be fetched afterwards.
static int finalize_lanes(k3_decoder *d, const std::vector<int> &lanes, hipStream_t st) {
  K3_REQUIRE(d->started, "k3_decoder_finalize_decoding: nothing to finalize");
  for (int l : lanes) K3_REQUIRE(l >= 0 && l < d->last_utts && !d->fresh[l], "k3_decoder_finalize: lane out of range or never advanced");
  if (lanes.empty()) { d->sel.clear(); return K3_OK; }
  if (lanes != d->lane_ids_uploaded) {      // (a whole-batch decode finalises lanes 0 .. n-1 every time: the list on the device is reused and the call stays asynchronous --
                                            //  the caller can queue the next batch's front end behind the token-passing kernel instead of waiting for it here)
    K3_HIP_CHECK(hipStreamSynchronize(st));            // d_lane_ids may still be read by an earlier finalize
    K3_HIP_CHECK(hipMemcpy(d->d_lane_ids, lanes.data(), sizeof(int) * lanes.size(), hipMemcpyHostToDevice));
    d->lane_ids_uploaded = lanes;
  }
  d->p.lane_ids = d->d_lane_ids;
  hipLaunchKernelGGL(k3_decode_prune_kernel, dim3((unsigned)lanes.size()), dim3(kPBlock), 0, st, d->p);
  K3_HIP_CHECK(hipGetLastError());
  if (d->profiling) K3_HIP_CHECK(hipEventRecord(d->ev[2], st));
  for (int l : lanes) d->lane_final[l] = 1;
  d->sel = lanes; d->last_stream = st; d->info_valid = false;
  return K3_OK;
}

extern "C" int k3_decoder_finalize_decoding(k3_decoder *d, void *stream) {
  K3_REQUIRE(d && d->started && !d->finalized, "k3_decoder_finalize_decoding: nothing to finalize");
  std::vector<int> all(d->last_utts); for (int u = 0; u < d->last_utts; u++) all[u] = u;
  const int rc = finalize_lanes(d, all, (hipStream_t)stream);
  if (rc == K3_OK) d->finalized = true;
  return rc;
}

// FinalizeDecoding of some lanes only (an utterance ended on those channels); k3_decoder_lattice_info / k3_decoder_get_raw_lattices
// then return these lanes, in the order given.  The other lanes of the group go on decoding.
extern "C" int k3_decoder_finalize_channels(k3_decoder *d, const int32_t *channels, int32_t n, void *stream) {
  K3_REQUIRE(d && channels && n >= 0, "k3_decoder_finalize_channels: bad argument");
  return finalize_lanes(d, std::vector<int>(channels, channels + n), (hipStream_t)stream);
}

extern "C" int32_t k3_decoder_num_frames_decoded(const k3_decoder *d, int32_t utt) { return (d && utt >= 0 && utt < d->last_utts) ? d->last_frames[utt] : -1; }

extern "C" int k3_decoder_decode_batch(k3_decoder *d, int32_t num_utts, const float *d_loglikes, int64_t ld, const int64_t *h_row_off, void *stream) {
  K3_REQUIRE(d && d_loglikes && h_row_off && num_utts > 0 && num_utts <= d->nlanes && ld >= d->num_pdfs, "k3_decoder_decode_batch: bad argument");
  int maxT = 0;
  for (int u = 0; u < num_utts; u++) {
    const long long T = h_row_off[u + 1] - h_row_off[u];
    K3_REQUIRE(T > 0 && T < (1 << 30), "k3_decoder_decode_batch: utterance with no frames");
    maxT = std::max(maxT, (int)T);
  }
  int rc;
  if ((rc = k3_decoder_init_decoding(d, num_utts, maxT, stream))) return rc;
  if ((rc = k3_decoder_advance_decoding(d, num_utts, d_loglikes, ld, h_row_off, stream))) return rc;
  return k3_decoder_finalize_decoding(d, stream);
}

// Make `stream` wait for the decoder's latest token-passing launch -- the last kernel that reads the caller's log-likelihood buffer (the pruning and output
// kernels behind it work on
// the lane's own pools).  A pipeline that refills that buffer for a later batch waits for THIS, not for the whole of k3_decoder_decode_batch: the pruning kernel cannot run beside a
// resident token-passing launch of another decoder object (LDS), so behind it the next front end would start tens of milliseconds later than it has to.
extern "C" int k3_decoder_stream_wait_token_passing(k3_decoder *d, void *stream) {
  K3_REQUIRE(d, "k3_decoder_stream_wait_token_passing: null decoder");
  if (d->ev_tp_recorded) K3_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, d->ev_tp, 0));
  return K3_OK;
}

static int fetch_info(k3_decoder *d) {
  if (d->info_valid) return K3_OK;
  K3_REQUIRE(d->last_utts > 0, "k3_decoder: no batch has been decoded");
  K3_HIP_CHECK(hipStreamSynchronize(d->last_stream));
  d->h_info.resize(d->last_utts);
  K3_HIP_CHECK(hipMemcpy(d->h_info.data(), d->p.info, sizeof(LaneInfo) * d->last_utts, hipMemcpyDeviceToHost));
  d->info_valid = true;
  return K3_OK;
}

extern "C" int k3_decoder_lattice_info(k3_decoder *d, int64_t *h_info) {
  K3_REQUIRE(d && h_info, "k3_decoder_lattice_info: null argument");
  { const int rc = fetch_info(d); if (rc) return rc; }
  int worst = K3_OK;
  for (size_t k = 0; k < d->sel.size(); k++) {
    const int u = d->sel[k];
    const LaneInfo &li = d->h_info[u]; int64_t *o = h_info + 10 * k;
    o[0] = li.status == kStOk ? li.out_states : 0; o[1] = li.status == kStOk ? li.out_arcs : 0; o[2] = li.status; o[3] = li.reached_final;
    o[4] = li.n_tokens; o[5] = li.n_links; o[6] = li.max_frame_tokens; o[7] = li.n_cands; o[8] = li.n_eps; o[9] = li.num_frames;
    if (li.status < 0) {
      worst = li.status;
      k3::set_error("k3_decoder: utterance %d failed with status %d (%s); tokens %lld links %lld max tokens/frame %d -- raise the k3_decoder_config capacities",
                                                          u, li.status, li.status == K3_ERR_OVERFLOW ? "capacity overflow" : "internal error", li.n_tokens,
                                                              li.n_links, li.max_frame_tokens); }
  }
  return worst;
}

extern "C" int k3_decoder_pool_growths(k3_decoder *d, int32_t *h_growths) {
  K3_REQUIRE(d && h_growths, "k3_decoder_pool_growths: null argument");
  { const int rc = fetch_info(d); if (rc) return rc; }
  for (size_t k = 0; k < d->sel.size(); k++) h_growths[k] = d->h_info[d->sel[k]].pool_grows;
  return K3_OK;
}

extern "C" int k3_decoder_order_sensitive_events(k3_decoder *d, int64_t *h_events) {
  K3_REQUIRE(d && h_events, "k3_decoder_order_sensitive_events: null argument");
  { const int rc = fetch_info(d); if (rc) return rc; }
  for (size_t k = 0; k < d->sel.size(); k++) h_events[k] = d->h_info[d->sel[k]].n_order_sensitive;
  return K3_OK;
}

extern "C" int k3_decoder_get_raw_lattices(k3_decoder *d, int32_t *st_frame, int32_t *st_state, float *st_cost, float *st_final, int32_t *arc_src, int32_t *arc_dst,
                                           int32_t *arc_il, int32_t *arc_ol, float *arc_g, float *arc_ac) {
  K3_REQUIRE(d && st_frame && st_state && st_cost && st_final && arc_src && arc_dst && arc_il && arc_ol && arc_g && arc_ac, "k3_decoder_get_raw_lattices: null argument");
  { const int rc = fetch_info(d); if (rc) return rc; }
  const int U = (int)d->sel.size();
  K3_REQUIRE(U > 0, "k3_decoder_get_raw_lattices: no finalised lanes");
  std::vector<long long> so(U + 1, 0), ao(U + 1, 0);
  for (int u = 0; u < U; u++) {
    const LaneInfo &li = d->h_info[d->sel[u]];
    const bool ok = li.status == kStOk;
    so[u + 1] = so[u] + (ok ? li.out_states : 0);
    ao[u + 1] = ao[u] + (ok ? li.out_arcs : 0);
  }
  const size_t NS = (size_t)so[U], NA = (size_t)ao[U];
  // one grow-only device staging area: [offsets | 4 state arrays | 6 arc arrays]
  const size_t need = sizeof(long long) * 2 * (U + 1) + 4 * (4 * NS + 6 * NA) + 256;
  if (need > d->out_bytes) {
    if (d->out_buf) (void)hipFree(d->out_buf);
    d->out_buf = nullptr; d->out_bytes = 0;
    K3_HIP_CHECK(hipMalloc(&d->out_buf, need + need / 4));
    d->out_bytes = need + need / 4;
  }
  OutParams o{};
  char *base = (char *)d->out_buf;
  long long *d_so = (long long *)base, *d_ao = d_so + (U + 1);
  char *q = (char *)(d_ao + (U + 1));
  o.st_frame = (int *)q; q += 4 * NS; o.st_state = (int *)q; q += 4 * NS; o.st_cost = (float *)q; q += 4 * NS; o.st_final = (float *)q; q += 4 * NS;
  o.arc_src = (int *)q; q += 4 * NA; o.arc_dst = (int *)q; q += 4 * NA; o.arc_il = (int *)q; q += 4 * NA; o.arc_ol = (int *)q; q += 4 * NA;
  o.arc_g = (float *)q; q += 4 * NA; o.arc_ac = (float *)q; q += 4 * NA;
  auto cleanup = [&]() {};
  o.st_off = d_so; o.arc_off = d_ao;
  hipStream_t st = d->last_stream;
#define K3_TRY(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { cleanup(); k3::set_error("HIP error %s: %s", hipGetErrorName(e__), #e); return K3_ERR_HIP; } } while (0)
  so.insert(so.end(), ao.begin(), ao.end());        // d_so and d_ao are adjacent: one upload
  K3_TRY(hipMemcpyAsync(d_so, so.data(), sizeof(long long) * 2 * (U + 1), hipMemcpyHostToDevice, st));
  d->p.lane_ids = d->d_lane_ids;                     // still holds d->sel (only a finalize call rewrites it)
  hipLaunchKernelGGL(k3_decode_output_kernel, dim3(U), dim3(kPBlock), 0, st, d->p, o);
  K3_TRY(hipGetLastError());
  // the caller's ten arrays laid out back to back in this order (kaldi_amd/decoder.py carves them out of one pinned buffer):
  // a single device-to-host copy
  const bool contiguous = (char *)st_state == (char *)st_frame + 4 * NS && (char *)st_cost == (char *)st_state + 4 * NS && (char *)st_final == (char *)st_cost + 4 * NS &&
                          (char *)arc_src == (char *)st_final + 4 * NS && (char *)arc_dst == (char *)arc_src + 4 * NA && (char *)arc_il == (char *)arc_dst + 4 * NA &&
                          (char *)arc_ol == (char *)arc_il + 4 * NA && (char *)arc_g == (char *)arc_ol + 4 * NA && (char *)arc_ac == (char *)arc_g + 4 * NA;
  if (contiguous) {
    K3_TRY(hipMemcpyAsync(st_frame, o.st_frame, 4 * (4 * NS + 6 * NA), hipMemcpyDeviceToHost, st));
    K3_TRY(hipStreamSynchronize(st));
    return K3_OK;
  }
  K3_TRY(hipStreamSynchronize(st));
  K3_TRY(hipMemcpy(st_frame, o.st_frame, 4 * NS, hipMemcpyDeviceToHost)); K3_TRY(hipMemcpy(st_state, o.st_state, 4 * NS, hipMemcpyDeviceToHost));
  K3_TRY(hipMemcpy(st_cost, o.st_cost, 4 * NS, hipMemcpyDeviceToHost)); K3_TRY(hipMemcpy(st_final, o.st_final, 4 * NS, hipMemcpyDeviceToHost));
  K3_TRY(hipMemcpy(arc_src, o.arc_src, 4 * NA, hipMemcpyDeviceToHost)); K3_TRY(hipMemcpy(arc_dst, o.arc_dst, 4 * NA, hipMemcpyDeviceToHost));
  K3_TRY(hipMemcpy(arc_il, o.arc_il, 4 * NA, hipMemcpyDeviceToHost)); K3_TRY(hipMemcpy(arc_ol, o.arc_ol, 4 * NA, hipMemcpyDeviceToHost));
  K3_TRY(hipMemcpy(arc_g, o.arc_g, 4 * NA, hipMemcpyDeviceToHost)); K3_TRY(hipMemcpy(arc_ac, o.arc_ac, 4 * NA, hipMemcpyDeviceToHost));
#undef K3_TRY
  cleanup();
  return K3_OK;
}


// GetBestPath (cuda-decoder.h:306) / the traceback of GetPartialHypothesis (:286): one-best path of each listed lane from the tokens it holds
// NOW -- after k3_decoder_advance_decoding (partial result: call it with use_final_probs = 0) or after finalisation.  Path u owns entries
// h_offsets[u] .. h_offsets[u+1] of the four arc arrays, in path order (first arc first); arc weight = LatticeWeight(graph, acoustic) as in
// GetRawLattice.  h_final_cost[u]: the final cost that was added (0 when none); h_relative_cost[u] = FinalRelativeCost() = min(cost + final)
// - min(cost) over the newest frame (+inf: no final state active) -- what kaldi::EndpointDetected takes; cap_arcs = capacity of the arc arrays.
extern "C" int k3_decoder_get_best_path(k3_decoder *d, const int32_t *channels, int32_t n, int32_t use_final_probs, int64_t *h_offsets, int64_t cap_arcs,
                                        int32_t *h_ilabel, int32_t *h_olabel, float *h_graph, float *h_ac, float *h_final_cost, float *h_relative_cost, int32_t *h_reached_final) {
  K3_REQUIRE(d && channels && n > 0 && n <= d->nlanes && h_offsets && h_ilabel && h_olabel && h_graph && h_ac && d->started,
      "k3_decoder_get_best_path: bad argument or nothing decoded");
  for (int i = 0; i < n; i++) K3_REQUIRE(channels[i] >= 0 && channels[i] < d->last_utts && !d->fresh[channels[i]],
      "k3_decoder_get_best_path: channel out of range or never advanced");
  hipStream_t st = d->last_stream; K3_HIP_CHECK(hipStreamSynchronize(st));
  int maxT = 0; for (int i = 0; i < n; i++) maxT = std::max(maxT, d->last_frames[channels[i]]);
  const int cap = 4 * maxT + 64;
  int *d_ids = nullptr, *d_il = nullptr, *d_ol = nullptr, *d_len = nullptr, *d_rf = nullptr; float *d_g = nullptr, *d_ac = nullptr, *d_fc = nullptr, *d_rc = nullptr;
  std::vector<void *> tmp; auto cleanup = [&]() { for (void *q : tmp) (void)hipFree(q); };
  int rc; const size_t nc = (size_t)n * cap;
  if ((rc = dmalloc(&tmp, &d_ids, (size_t)n)) || (rc = dmalloc(&tmp, &d_il, nc)) || (rc = dmalloc(&tmp, &d_ol, nc)) || (rc = dmalloc(&tmp, &d_g, nc)) ||
      (rc = dmalloc(&tmp, &d_ac, nc)) ||
      (rc = dmalloc(&tmp, &d_len, (size_t)n)) || (rc = dmalloc(&tmp, &d_fc, (size_t)n)) || (rc = dmalloc(&tmp, &d_rc, (size_t)n)) ||
          (rc = dmalloc(&tmp, &d_rf, (size_t)n))) { cleanup(); return rc; }
#define K3_TRY(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { cleanup(); k3::set_error("HIP error %s: %s", hipGetErrorName(e__), #e); return K3_ERR_HIP; } } while (0)
  K3_TRY(hipMemcpy(d_ids, channels, sizeof(int) * n, hipMemcpyHostToDevice));
  DecParams p = d->p; p.lane_ids = d_ids;
  BestPathParams o{d_il, d_ol, d_g, d_ac, d_len, d_fc, d_rc, d_rf, cap, use_final_probs ? 1 : 0};
  hipLaunchKernelGGL(k3_decode_best_path_kernel, dim3(n), dim3(kPBlock), 0, st, p, o);
  K3_TRY(hipGetLastError());
  std::vector<int> len(n), rf(n), il(nc), ol(nc); std::vector<float> g(nc), ac(nc), fc(n), rcst(n);
  K3_TRY(hipMemcpyAsync(len.data(), d_len, sizeof(int) * n, hipMemcpyDeviceToHost, st)); K3_TRY(hipMemcpyAsync(rf.data(), d_rf, sizeof(int) * n, hipMemcpyDeviceToHost, st));
  K3_TRY(hipMemcpyAsync(il.data(), d_il, sizeof(int) * nc, hipMemcpyDeviceToHost, st)); K3_TRY(hipMemcpyAsync(ol.data(), d_ol, sizeof(int) * nc, hipMemcpyDeviceToHost, st));
  K3_TRY(hipMemcpyAsync(g.data(), d_g, sizeof(float) * nc, hipMemcpyDeviceToHost, st)); K3_TRY(hipMemcpyAsync(ac.data(), d_ac, sizeof(float) * nc, hipMemcpyDeviceToHost, st));
  K3_TRY(hipMemcpyAsync(fc.data(), d_fc, sizeof(float) * n, hipMemcpyDeviceToHost, st)); K3_TRY(hipMemcpyAsync(rcst.data(), d_rc, sizeof(float) * n, hipMemcpyDeviceToHost, st));
  K3_TRY(hipStreamSynchronize(st));
#undef K3_TRY
  int64_t total = 0; h_offsets[0] = 0;
  for (int u = 0; u < n; u++) {
    if (len[u] < 0) {
      cleanup();
      k3::set_error("k3_decoder_get_best_path: path of channel %d longer than %d arcs", channels[u], cap);
      return K3_ERR_OVERFLOW;
    }
    total += len[u];
    h_offsets[u + 1] = total;
  }
  if (total > cap_arcs) { cleanup(); k3::set_error("k3_decoder_get_best_path: %lld arcs but room for %lld", (long long)total, (long long)cap_arcs); return K3_ERR_OVERFLOW; }
  for (int u = 0; u < n; u++)
    for (int k = 0; k < len[u]; k++) {      // the kernel wrote the last arc first
      const size_t src = (size_t)u * cap + (size_t)(len[u] - 1 - k); const int64_t q = h_offsets[u] + k;
      h_ilabel[q] = il[src]; h_olabel[q] = ol[src]; h_graph[q] = g[src]; h_ac[q] = ac[src];
    }
  for (int u = 0; u < n; u++) { if (h_final_cost) h_final_cost[u] = fc[u]; if (h_relative_cost) h_relative_cost[u] = rcst[u]; if (h_reached_final) h_reached_final[u] = rf[u]; }
  cleanup();
  return K3_OK;
}

extern "C" int k3_decoder_phase_cycles(k3_decoder *d, int64_t *h_cycles /* [16] summed over lanes, reset */) {
  K3_REQUIRE(d && h_cycles, "k3_decoder_phase_cycles: null argument");
  std::vector<long long> h((size_t)d->nlanes * 16);
  K3_HIP_CHECK(hipMemcpy(h.data(), d->p.prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  const bool mx = getenv("K3_PROF_MAX") != nullptr;      // profiling builds: the slowest lane instead of the sum
  for (int i = 0; i < 16; i++) {
    h_cycles[i] = 0;
    for (int l = 0; l < d->nlanes; l++) h_cycles[i] = mx ? std::max<int64_t>(h_cycles[i], h[(size_t)l * 16 + i]) : h_cycles[i] + h[(size_t)l * 16 + i];
  }
  K3_HIP_CHECK(hipMemset(d->p.prof, 0, h.size() * sizeof(long long)));
  return K3_OK;
}

extern "C" int k3_decoder_frame_stats(k3_decoder *d, int32_t utt, int32_t *ntoks, float *cur, float *ab, float *next, float *co) {
  K3_REQUIRE(d && utt >= 0 && utt < d->last_utts, "k3_decoder_frame_stats: bad utterance index");
  { const int rc = fetch_info(d); if (rc) return rc; }
  const int T = d->last_frames[utt]; const size_t off = (size_t)utt * d->fstride;
  if (ntoks) K3_HIP_CHECK(hipMemcpy(ntoks, d->p.st_ntoks + off, 4 * T, hipMemcpyDeviceToHost));
  if (cur) K3_HIP_CHECK(hipMemcpy(cur, d->p.st_cur + off, 4 * T, hipMemcpyDeviceToHost));
  if (ab) K3_HIP_CHECK(hipMemcpy(ab, d->p.st_ab + off, 4 * T, hipMemcpyDeviceToHost));
  if (next) K3_HIP_CHECK(hipMemcpy(next, d->p.st_next + off, 4 * T, hipMemcpyDeviceToHost));
  if (co) K3_HIP_CHECK(hipMemcpy(co, d->p.st_co + off, 4 * T, hipMemcpyDeviceToHost));
  return K3_OK;
}
