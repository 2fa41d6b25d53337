// k3_decoder_dev.h -- device-side definitions shared by the decoder translation units: k3_decoder.hip (two-pass token passing, pruning, output, host API;
// K3_DEC_BLOCK threads per lane) and k3_decoder_lit.hip (literal_order token passing, its own K3_DEC_BLOCK).  Everything lives in an anonymous namespace: each
// translation unit gets its own copy, parametrised by the K3_DEC_BLOCK it was compiled with.
#ifndef K3_DECODER_DEV_H_
#define K3_DECODER_DEV_H_
#include "k3_common.h"
#pragma clang fp contract(off)   // costs must be formed add-by-add like the reference (no FMA contraction)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

namespace {


#ifndef K3_DEC_BLOCK
#define K3_DEC_BLOCK 512
#endif
constexpr int kBlock = K3_DEC_BLOCK;      // threads per lane-workgroup
constexpr int kPBlock = 512;                 // threads per lane-workgroup of the prune / output kernels
constexpr int kWaves = (kBlock > kPBlock ? kBlock : kPBlock) / 64;
constexpr unsigned kEncInf = 0xFF800000u;   // enc(+inf)
constexpr unsigned kEncMax = 0xFFFFFFFFu;
constexpr int kEmpty = -1;

struct Slot { int key; unsigned cost; int tok; int stamp; };          // 16 B hash slot
constexpr unsigned kEpsFlag = 0x80000000u;     // ArcRec::next bit 31: the destination state has epsilon arcs (set by k3_fst_create)
struct ArcRec { int next; float w; int pdf; int olabel; };            // 16 B graph arc
// 16 B forward link (token indices are lane-pool indices): `tot` = (src cost + acoustic) + graph exactly as the forward pass formed
// it, which is the term PruneForwardLinks needs (:339-341), so pruning never touches the graph; the arc id (labels, graph cost for
// the lattice writer) lives in a parallel 4 B array that only the output kernel reads
struct Link { unsigned src, dst; float tot; float ac; };
// the link / token pools are written once per frame and read by the pruning kernel much later: streaming stores keep them from evicting the per-lane scratch out of L2
#ifndef K3_DEC_NT
#define K3_DEC_NT 1
#endif
__device__ __forceinline__ void store_link(Link *dst, const Link &v) {
#if K3_DEC_NT
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(u32x4{v.src, v.dst, __float_as_uint(v.tot), __float_as_uint(v.ac)}, reinterpret_cast<u32x4 *>(dst));
#else
  *dst = v;
#endif
}
template <typename T> __device__ __forceinline__ void store_stream(T *dst, T v) {
#if K3_DEC_NT
  __builtin_nontemporal_store(v, dst);
#else
  *dst = v;
#endif
}

enum { kStOk = 0, kStNoTokens = 1 };
// internal to the forward kernels: a lane's token / link pool is full -- grow it and go on (never leaves the kernel; K3_ERR_OVERFLOW when the spare arena is
// exhausted)
constexpr int kErrPool = -1000;

// A lane's token / link pools.  The configured capacities (k3_decoder_config::lane_tokens_cap / lane_links_cap = the reference's ntokens_pre_allocated) are a RESERVATION like the
// reference's (cuda-decoder.cc:232-238 reserves, the per-channel vectors grow): a lane that outgrows them moves, inside the token-passing kernel and without
// the host, to a block of
// at least twice the size carved off the decoder's spare arena (grow_lane_pools below); every kernel reads a lane's pointers and capacities from this record.
struct LanePool { int *tok_state; unsigned *tok_cost; float *tok_extra; int *newidx; Link *links; int *link_arc; long long tcap, lcap; };

struct LaneInfo {            // per lane, written by the kernels, read by the host
  long long n_tokens, n_links, n_cands, n_eps;   // created / emitting arcs examined / eps arcs examined
  int status, reached_final, max_frame_tokens, num_frames;
  int out_states, out_arcs;               // after pruning
  int live_overflow;                      // the survivor lists were too small: the output kernel rescans the pools
  long long cur_base; int n_cur;          // resume point of AdvanceDecoding: first token / token count of the newest frame
  float final_best_cost; int final_empty;
  // SURVEY 9.1 "order-sensitive events".  literal_order: forward links the serial algorithm creates only because next_cutoff was still loose when
  // their arc was examined (tot >= the frame's final next_cutoff) -- the oracle's extra_links.  Default (two-pass) mode: emitting arcs below the
  // pre-pass bound but not below the final bound, an upper bound on the arcs the two rules can disagree on.
  long long n_order_sensitive;
  // literal_order: HashList bucket count (PossiblyResizeHash) and which half of lt_order holds the newest frame, carried across AdvanceDecoding calls
  int hash_size, order_sel;
  int pool_grows;                         // how often this lane moved to bigger pools (since the decoder was created)
};

#ifdef K3_DEC_PROF
#ifndef K3_DEC_PROF_MAXTOK
#define K3_DEC_PROF_MAXTOK 0x7FFFFFFF      /* count only frames built from at most this many tokens */
#endif
#ifndef K3_DEC_PROF_MINTOK
#define K3_DEC_PROF_MINTOK 0
#endif
#define K3_T(i) do { if (threadIdx.x == 0) { const long long now__ = (long long)__builtin_readcyclecounter(); if (sh.prof_n <= K3_DEC_PROF_MAXTOK && sh.prof_n >= K3_DEC_PROF_MINTOK) sh.prof[i] += now__ - t_last__; t_last__ = now__; } } while (0)
#define K3_TW(i) do { __builtin_amdgcn_s_waitcnt(0); K3_T(i); } while (0)      /* drain this wave's memory ops first: attributes load latency to the segment */
#else
#define K3_T(i) do { } while (0)
#define K3_TW(i) do { } while (0)
#endif
#ifdef K3_PRUNE_PROF
#define K3_PT(i) do { if (threadIdx.x == 0) { const long long now__ = (long long)__builtin_readcyclecounter(); s_pprof[i] += now__ - pt_last__; pt_last__ = now__; } } while (0)
#else
#define K3_PT(i) do { } while (0)
#endif
struct DecParams {
  long long *prof;   // [nlanes x 16] cycle counters per phase (only with -DK3_DEC_PROF)
  // graph
  const int2 *offs; const ArcRec *arcs; const float *final_cost; const int *arc_ilabel; int start;
  // per arc, what a token created through it needs to know about its state: {first arc of the destination, emitting arcs | eps arcs << 16 (65535 = more: look the
  // state up)} -- derived from offs / arcs when the decoder is created, loaded NEXT TO the arc instead of behind it (literal_order's LDS-resident frames)
  const int2 *dinfo;
  // config
  float beam, lattice_beam, beam_delta; int max_active, min_active;
  int frame_tokens_cap, frame_cands_cap, hash_mask;
  // input
  const float *loglikes; long long ld; const long long *row_off; int num_pdfs; int use_lds_row;
  const float *const *lane_rows;          // non-null: lane l's next frames start at lane_rows[l] (rows ld apart) instead of row row_off[l] of `loglikes`
  // [nlanes] 1: InitDecoding first (start token + eps closure), frames start at 0; 0: continue after LaneInfo::num_frames frames (AdvanceDecoding)
  const int *fresh;
  const int *lane_ids;                    // prune / output kernels: the lanes being finalised (workgroup b works on lane lane_ids[b]); null = lane b
  // per-lane pools: pools[l] (initially lane l's slice of one allocation per array; after a growth a block of the spare arena)
  // spare arena: bump-allocated by the lanes that outgrow their pools, never freed before the decoder is destroyed
  LanePool *pools;
  char *spare;
  unsigned long long *spare_used;
  long long spare_bytes;
  int *live_tok; long long *live_link; int live_cap;   // survivors of the pruning pass (pool indices), per lane
  Slot *hash; int *tok_slot; int *wl;            // wl: 2 x frame_tokens_cap
  float *c_tot, *c_ac; int *c_dst, *c_arc, *c_src;
  // per-lane per-frame arrays, stride fstride = max_frames + 2
  long long fstride; long long *tok_off; long long *link_off_e, *link_off_n;
  int *st_ntoks; float *st_cur, *st_ab, *st_next, *st_co;
  LaneInfo *info;
  // literal_order scratch (k3_decoder_literal.h), per lane
  // the lt_* pointers below are lane 0's; lane l's arrays start lt_lane_bytes * l further on
  int literal, fast_cap, lit_force_hbm_order;
  float hash_ratio;
  int hash_cap, seq_words_cap, eps_cap, stack_cap;
  long long lt_lane_bytes;
  int *lt_order;          // [2 x frame_tokens_cap] HashList order of the current / the next frame (local token indices)
  int *lt_by_ins;         // [frame_tokens_cap] tokens of the newest frame in creation order (the final-frame sweeps walk it backwards)
  unsigned *lt_label;     // [frame_tokens_cap] creation label of a token being built (all 0xFFFFFFFF between frames)
  int *lt_dense, *lt_grp; unsigned *lt_lead;            // [frame_tokens_cap (+1)]
  unsigned *lt_bm, *lt_wpre;                             // [seq_words_cap] label bitmap (all 0 between uses) / word prefix counts
  unsigned *lt_cmin; int *lt_ccnt;                       // [frame_tokens_cap / 64 + 2] per 64-token chunk: min (tot + adaptive_beam), emitting arcs
  float *lt_c0;                                          // [frame_tokens_cap] token costs right after ProcessEmitting
  int2 *lt_crng; int *lt_cdst; float *lt_cw;             // closure sub-graph in token space: per token (first, count), per eps arc (dst token | -1, weight)
  float *lt_rcost; int *lt_rflag, *lt_rown, *lt_stack, *lt_iq, *lt_c2t; int2 *lt_arcs2; int4 *lt_meta;   // replay state (global copies; small frames use LDS)
  // [frame_tokens_cap] component replay: union-find parents, roots grouped by component, workers, per-component counts / offsets, per-root (first, count) of
  // the tokens it created
  int *lt_par, *lt_rtmp;
  int2 *lt_rlist, *lt_rinfo;
  int4 *lt_cinfo, *lt_coffs, *lt_wrec, *lt_vis, *lt_btab;
  // literal_order kernel: q_lanes = the call's lanes (those with frames or a fresh start), longest first, q_n of them: workgroup b decodes lane q_lanes[b] (null: lane b).
  // q_head (work-queue builds, fewer workgroups than lanes): the cursor through which a workgroup takes its next entry.
  int *q_head; const int *q_lanes; long long q_n;
  // literal_order kernel: what InitDecoding (the start token and its epsilon closure, lattice-faster-decoder.cc:63-81) leaves in a lane -- the same for every utterance of a
  // decoder (graph, beam and hash ratio decide it), computed once when the decoder is created (one lane, no frames) and copied into a fresh lane instead of being worked out
  // again (0.9 M cycles of a lane's ~98 M): tokens (state, cost) in creation order, the closure's forward links, the HashList visit order and the creation order.  tpl_n = 0: none.
  const int *tpl_state; const unsigned *tpl_cost; const Link *tpl_links; const int *tpl_arc; const int *tpl_order, *tpl_by_ins;
  int tpl_n, tpl_nl; long long tpl_eps;
  // ... and the FIRST FRAME's structure (k3_decode_frame0_from_template_kernel).  With n_cur <= min_active tokens after InitDecoding the adaptive beam of frame 0 is +inf
  // (lattice-faster-decoder.cc:700-716): every emitting arc of those tokens is accepted and every epsilon arc of the closure passes, whatever the utterance -- so the tokens
  // of frame 1, the forward links (as source / destination / arc), the emitting tokens' creation ranks, the closure sub-graph with its components, root lists and worker
  // records are the same for every utterance of a decoder; only costs (hence the ORDER the closure creates its tokens in) depend on the log-likelihoods.  Captured once when
  // the decoder is created (one lane, one frame, the capture build of the token-passing kernel).  t0_n = 0: none.
  const int4 *t0_elinks, *t0_xlinks;      // emitting links {src << 16 | dst, arc, pdf, weight bits}; epsilon links {src << 16 | dst, arc, 0, weight bits}; token indices frame-local
  const int *t0_state, *t0_rank1, *t0_c2t; const int4 *t0_meta; const int2 *t0_ar, *t0_rlist; const int4 *t0_wrec;
  int t0_n_e, t0_n, t0_m_e, t0_nle, t0_nlx, t0_ncid, t0_narc, t0_niq, t0_nworkers; unsigned t0_hash_size; float t0_best; int t0_pad; long long t0_eps;
  int *row_skip;        // [nlanes] rows of this call's log-likelihoods a lane has already consumed in front of the token-passing kernel (0 or 1: frame 0 from the template)
  long long *cap;       // capture build only: [16] scalars of the frame just decoded by lane 0
};

__device__ __forceinline__ unsigned enc(float x) { unsigned b = __float_as_uint(x); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ float dec(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }
// Every atomic below works on the state of ONE lane, and a lane is owned by one workgroup in every kernel: workgroup scope is enough.  (Agent
// scope makes gfx950 resolve the operation beyond the XCD's L2 -- the coherence point of the eight XCDs -- at several times the latency.)
#ifndef K3_DEC_SCOPE
#define K3_DEC_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
#define K3_ALD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, K3_DEC_SCOPE)
#define K3_AST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, K3_DEC_SCOPE)
template <typename T, typename V> __device__ __forceinline__ T k3a_add(T *p, V v) { return __hip_atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED, K3_DEC_SCOPE); }
template <typename T, typename V> __device__ __forceinline__ T k3a_min(T *p, V v) { return __hip_atomic_fetch_min(p, (T)v, __ATOMIC_RELAXED, K3_DEC_SCOPE); }
template <typename T, typename V> __device__ __forceinline__ T k3a_or(T *p, V v) { return __hip_atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED, K3_DEC_SCOPE); }
template <typename T, typename V> __device__ __forceinline__ T k3a_exch(T *p, V v) { return __hip_atomic_exchange(p, (T)v, __ATOMIC_RELAXED, K3_DEC_SCOPE); }
template <typename T, typename V, typename W> __device__ __forceinline__ T k3a_cas(T *p, V expected, W desired) {
  T e = (T)expected; __hip_atomic_compare_exchange_strong(p, &e, (T)desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, K3_DEC_SCOPE); return e;
}

__device__ __forceinline__ unsigned hash_state(int s) { unsigned x = (unsigned)s * 2654435761u; return x ^ (x >> 15); }

struct Shared {
  unsigned long long red64[kWaves];
  int redi[kWaves];
  int hist[256];
  int n_next, n_cand, n_wl[3], err_r[4], err, sel_digit, sel_k, flag;      // n_wl / err_r rotate over the eps rounds (one barrier per round)
  unsigned long long n_eps, n_emit, n_os;
  unsigned min_tot;
  long long n_link;
  unsigned long long bcast64;
  int prof_n; long long prof[16];
};

// Cross-lane primitives on DPP row shifts / row broadcasts (gfx9: row_shr:n 0x11n, row_bcast:15 0x142, row_bcast:31 0x143, wave_shr:1 0x138): a few cycles per step, where
// __shfl* compiles to ds_bpermute -- an LDS-crossbar round trip per step and word, on the critical path of every reduction and scan of a frame.  `idn` is what a lane
// without a source keeps (the operation's identity).  All 64 lanes must be active.
#define K3_DPP(idn, v, ctrl, rows) __builtin_amdgcn_update_dpp((int)(idn), (int)(v), ctrl, rows, 0xf, false)
__device__ __forceinline__ int wave_incl_sum_i32(int v) {
  v += K3_DPP(0, v, 0x111, 0xf); v += K3_DPP(0, v, 0x112, 0xf); v += K3_DPP(0, v, 0x114, 0xf); v += K3_DPP(0, v, 0x118, 0xf);
  v += K3_DPP(0, v, 0x142, 0xa); v += K3_DPP(0, v, 0x143, 0xc);
  return v;
}
__device__ __forceinline__ unsigned wave_incl_min_u32(unsigned v) {
  unsigned t;
  t = (unsigned)K3_DPP(~0u, v, 0x111, 0xf); v = t < v ? t : v; t = (unsigned)K3_DPP(~0u, v, 0x112, 0xf); v = t < v ? t : v;
  t = (unsigned)K3_DPP(~0u, v, 0x114, 0xf); v = t < v ? t : v; t = (unsigned)K3_DPP(~0u, v, 0x118, 0xf); v = t < v ? t : v;
  t = (unsigned)K3_DPP(~0u, v, 0x142, 0xa); v = t < v ? t : v; t = (unsigned)K3_DPP(~0u, v, 0x143, 0xc); v = t < v ? t : v;
  return v;
}
__device__ __forceinline__ unsigned wave_shr1_u32(unsigned v, unsigned idn) { return (unsigned)K3_DPP(idn, v, 0x138, 0xf); }      // lane i gets lane i - 1's value, lane 0 gets idn
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
  unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
#define K3_MIN64_STEP(ctrl, rows) { const unsigned th = (unsigned)K3_DPP(~0u, hi, ctrl, rows), tl = (unsigned)K3_DPP(~0u, lo, ctrl, rows); const bool take = th < hi || (th == hi && tl < lo); hi = take ? th : hi; lo = take ? tl : lo; }
  K3_MIN64_STEP(0x111, 0xf) K3_MIN64_STEP(0x112, 0xf) K3_MIN64_STEP(0x114, 0xf) K3_MIN64_STEP(0x118, 0xf) K3_MIN64_STEP(0x142, 0xa) K3_MIN64_STEP(0x143, 0xc)
#undef K3_MIN64_STEP
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)wave_incl_min_u32(v), 63); }
__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_incl_sum_i32(v), 63); }
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#define K3_SUM64_STEP(ctrl, rows) { const unsigned th = (unsigned)K3_DPP(0, (unsigned)(v >> 32), ctrl, rows), tl = (unsigned)K3_DPP(0, (unsigned)v, ctrl, rows); v += ((unsigned long long)th << 32) | tl; }
  K3_SUM64_STEP(0x111, 0xf) K3_SUM64_STEP(0x112, 0xf) K3_SUM64_STEP(0x114, 0xf) K3_SUM64_STEP(0x118, 0xf) K3_SUM64_STEP(0x142, 0xa) K3_SUM64_STEP(0x143, 0xc)
#undef K3_SUM64_STEP
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
}
// Position j0 + lane of a wavefront's arc sequence (lane i owns positions [excl_i, excl_i + deg_i)) -> its owner lane.  The binary search over the scan it replaces was six
// dependent ds_bpermute round trips per 64 arcs.  Here: the starts that fall into this chunk of 64 positions as a bit set (OR-reduced with DPP), the owner's rank among the
// lanes with arcs = (owners that start before the chunk) + (starts at or before my position) - 1, and the rank-th set bit of the ballot of those lanes by a popcount descent.
__device__ __forceinline__ int wave_owner_of(int j0, int deg, int excl, unsigned long long has_arcs) {
  const int lane = threadIdx.x & 63, s_ = excl - j0; const bool pos = deg > 0;
  const bool in_chunk = pos && s_ >= 0 && s_ < 64;
  unsigned lo = in_chunk && s_ < 32 ? 1u << s_ : 0u, hi = in_chunk && s_ >= 32 ? 1u << (s_ - 32) : 0u;
#define K3_OR_STEP(ctrl, rows) { lo |= (unsigned)K3_DPP(0, lo, ctrl, rows); hi |= (unsigned)K3_DPP(0, hi, ctrl, rows); }
  K3_OR_STEP(0x111, 0xf) K3_OR_STEP(0x112, 0xf) K3_OR_STEP(0x114, 0xf) K3_OR_STEP(0x118, 0xf) K3_OR_STEP(0x142, 0xa) K3_OR_STEP(0x143, 0xc)
#undef K3_OR_STEP
  const unsigned long long starts = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
  const int before = __popcll(__ballot(pos && s_ < 0));
  int r = before + __popcll(starts & ((2ull << lane) - 1ull)) - 1;      // rank of my owner among the lanes with arcs
  unsigned w = (unsigned)has_arcs; int base = 0, c = __popc(w);
  if (r >= c) { r -= c; w = (unsigned)(has_arcs >> 32); base = 32; }
  c = __popc(w & 0xFFFFu); if (r >= c) { r -= c; w >>= 16; base += 16; }
  c = __popc(w & 0xFFu); if (r >= c) { r -= c; w >>= 8; base += 8; }
  c = __popc(w & 0xFu); if (r >= c) { r -= c; w >>= 4; base += 4; }
  c = __popc(w & 0x3u); if (r >= c) { r -= c; w >>= 2; base += 2; }
  c = (int)(w & 1u); if (r >= c) base += 1;
  return base > 63 ? 63 : (base < 0 ? 0 : base);      // (positions beyond the total: any lane; the caller masks them)
}
__device__ unsigned long long block_min_u64(unsigned long long v, Shared &sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_min_u64(v);
  __syncthreads();
  if (lane == 0) sh.red64[wave] = v;
  __syncthreads();
  unsigned long long r = sh.red64[0];
  const int nw = (int)blockDim.x >> 6;
  for (int w = 1; w < nw; w++) r = sh.red64[w] < r ? sh.red64[w] : r;
  return r;
}
__device__ int block_sum_i32(int v, Shared &sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum_i32(v);
  __syncthreads();
  if (lane == 0) sh.redi[wave] = v;
  __syncthreads();
  int r = 0;
  const int nw = (int)blockDim.x >> 6;
  for (int w = 0; w < nw; w++) r += sh.redi[w];
  return r;
}

// uniform snapshot of the lane's error flag (read between two barriers so that no wavefront can race ahead and set it)
__device__ __forceinline__ int block_err(Shared &sh) { __syncthreads(); const int e = sh.err; __syncthreads(); return e; }

// ballot/popcount aggregated append: returns the slot index for lanes with pred (one LDS atomic per wavefront)
__device__ __forceinline__ int wave_append(bool pred, int *counter) {
  const int lane = threadIdx.x & 63;
  const unsigned long long m = __ballot(pred);
  if (m == 0) return 0;
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = k3a_add(counter, __popcll(m));
  base = __builtin_amdgcn_readlane(base, leader);          // leader is wave-uniform: a v_readlane, not an LDS-crossbar shuffle
  return base + __popcll(m & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ long long wave_append64(bool pred, long long *counter) {
  const int lane = threadIdx.x & 63;
  const unsigned long long m = __ballot(pred);
  if (m == 0) return 0;
  const int leader = __ffsll((long long)m) - 1;
  long long base = 0;
  if (lane == leader) base = (long long)k3a_add((unsigned long long *)counter, (unsigned long long)__popcll(m));
  base = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)base >> 32), leader) << 32) |
                     (unsigned)__builtin_amdgcn_readlane((int)(unsigned)base, leader));
  return base + __popcll(m & ((1ull << lane) - 1ull));
}

// 64 (begin, degree) pairs, one per lane -> f(valid, arc, owner_lane) once per arc, one arc per lane per step.
// Every lane of the wavefront must call this (uniform control flow); f must keep the wavefront converged.
template <typename F>
__device__ __forceinline__ void wave_expand(const ArcRec *arcs, int beg, int deg, F &&f) {
  const int lane = threadIdx.x & 63;
  // inclusive scan of the degrees with DPP row shifts / row broadcasts (a few cycles each; the ds_bpermute shuffle chain they replace
  // is six dependent LDS-crossbar round trips)
  int incl = deg;
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);      // row_shr:1
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);      // row_shr:2
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);      // row_shr:4
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);      // row_shr:8
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - deg;
  const unsigned long long has_arcs = __ballot(deg > 0);
  const int rel = beg - excl;      // (arc of position j of owner o = beg_o + (j - excl_o): one cross-lane read per arc instead of two)
  auto locate = [&](int j, int &arc, int &owner) {
    const int lo = wave_owner_of(j - lane, deg, excl, has_arcs);
    arc = __shfl(rel, lo) + j; owner = lo;
  };
  for (int j0 = 0; j0 < total; j0 += 64) {
    int a, o; ArcRec r{};
    locate(j0 + lane, a, o);
    if (j0 + lane < total) r = arcs[a];
    f(j0 + lane < total, a, o, r);
  }
}

// wave_expand with the destination record of every arc (DecParams::dinfo) loaded together with the arc
template <typename F>
__device__ __forceinline__ void wave_expand_d(const ArcRec *arcs, const int2 *dinfo, int beg, int deg, F &&f) {
  const int lane = threadIdx.x & 63;
  const int incl = wave_incl_sum_i32(deg);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - deg;
  const unsigned long long has_arcs = __ballot(deg > 0);
  for (int j0 = 0; j0 < total; j0 += 64) {
    const int lo = wave_owner_of(j0, deg, excl, has_arcs);
    const int a = __shfl(beg - excl, lo) + j0 + lane; ArcRec r{}; int2 di = make_int2(0, 0);
    if (j0 + lane < total) { r = arcs[a]; di = dinfo[a]; }
    f(j0 + lane < total, a, lo, r, di);
  }
}

// find the slot of `state` or claim an empty one.  All key accesses are atomics (performed at L2).
__device__ __forceinline__ int slot_find_or_claim(Slot *tab, unsigned mask, int state, bool *claimed) {
  unsigned h = hash_state(state) & mask;
  for (unsigned probe = 0; probe <= mask; probe++) {
    const int old = k3a_cas(&tab[h].key, kEmpty, state);
    if (old == kEmpty) { *claimed = true; return (int)h; }
    if (old == state) { *claimed = false; return (int)h; }
    h = (h + 1) & mask;
  }
  *claimed = false; return -1;
}
__device__ __forceinline__ int slot_find(Slot *tab, unsigned mask, int state) {
  unsigned h = hash_state(state) & mask;
  for (unsigned probe = 0; probe <= mask; probe++) {
    const int k = K3_ALD(&tab[h].key);
    if (k == state) return (int)h;
    if (k == kEmpty) return -1;
    h = (h + 1) & mask;
  }
  return -1;
}

// Two-level state -> token table of the frame being built.  Level 1 lives in LDS (kHL slots, SoA key/cost/token): a state
// is looked up in a kProbe-slot window; slots are never freed inside a frame, so once a window is full it stays full and
// every thread agrees that such a state belongs to level 2, the per-lane open-addressing table in HBM.  Slot ids < kHL are
// LDS slots, ids >= kHL are kHL + index of the HBM slot.  Typical frames (~1-3 k tokens) never leave LDS.
#ifndef K3_DEC_HL
#define K3_DEC_HL 4096
#endif
#ifndef K3_DEC_LDSROW
#define K3_DEC_LDSROW 1
#endif
constexpr int kHL = K3_DEC_HL, kProbe = 48;
constexpr int kRowRegs = K3_DEC_LDSROW ? (6400 + kBlock - 1) / kBlock : 1;   // registers that carry the next frame's log-likelihood row (LDS rows are <= 25 KB)
constexpr int kCurRegs = 4;       // frames of <= kCurRegs * kBlock tokens hand their (cost, state) pairs to the next frame in registers
constexpr int kWlLds = 1024;      // the first kWlLds work-list entries of an eps round live in LDS (16-bit LDS slot ids)
#define K3_LLD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
struct Table {
  int *lkey; unsigned *lcost; int *ltok; unsigned *lmark;     // LDS: [kHL], [kHL], [kHL], [3][kHL / 32] (mark bits of three consecutive eps rounds)
  Slot *g; unsigned gmask;                                     // HBM level
  __device__ __forceinline__ int claim(int state, bool *claimed) const {
    unsigned h = hash_state(state) & (kHL - 1);
    for (int probe = 0; probe < kProbe; probe++) {
      int k = K3_LLD(&lkey[h]); bool cl = false;
      if (k == kEmpty) { const int old = k3a_cas(&lkey[h], kEmpty, state); if (old == kEmpty) { cl = true; k = state; } else k = old; }
      if (k == state) { *claimed = cl; return (int)h; }
      h = (h + 1) & (kHL - 1);
    }
    const int gs = slot_find_or_claim(g, gmask, state, claimed);
    return gs < 0 ? -1 : kHL + gs;
  }
  __device__ __forceinline__ int find(int state) const {
    unsigned h = hash_state(state) & (kHL - 1);
    for (int probe = 0; probe < kProbe; probe++) {
      const int k = K3_LLD(&lkey[h]);
      if (k == state) return (int)h;
      if (k == kEmpty) return -1;           // a window with a hole was never full: the state cannot be in level 2
      h = (h + 1) & (kHL - 1);
    }
    const int gs = slot_find(g, gmask, state);
    return gs < 0 ? -1 : kHL + gs;
  }
  __device__ __forceinline__ unsigned cost_min(int id, unsigned e) const { return id < kHL ? k3a_min(&lcost[id], e) : k3a_min(&g[id - kHL].cost, e); }
  __device__ __forceinline__ unsigned cost(int id) const { return id < kHL ? K3_LLD(&lcost[id]) : K3_ALD(&g[id - kHL].cost); }
  __device__ __forceinline__ int key(int id) const { return id < kHL ? K3_LLD(&lkey[id]) : K3_ALD(&g[id - kHL].key); }
  __device__ __forceinline__ int tok(int id) const { return id < kHL ? K3_LLD(&ltok[id]) : K3_ALD(&g[id - kHL].tok); }
  __device__ __forceinline__ void set_tok(int id, int t) const {
    if (id < kHL) __hip_atomic_store(&ltok[id], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else K3_AST(&g[id - kHL].tok, t);
  }
  // token index of a slot some other thread claimed a moment ago: the claimer publishes it right after its claim (same
  // program point for every lane of a wave, so lanes of one wave never wait on each other); bounded in case of a bug
  __device__ __forceinline__ int wait_tok(int id, int *err) const {
    for (int spin = 0; spin < (1 << 22); spin++) {
      const int t = tok(id);
      if (t >= 0) return t;
      __builtin_amdgcn_s_sleep(1);
    }
    *err = K3_ERR_HIP; return -1;
  }
  // true if the slot was not yet queued for round `stamp` (LDS slots: one bit per slot, cleared at the start of every round)
  __device__ __forceinline__ bool mark(int id, int stamp) const {
    if (id < kHL) { const unsigned bit = 1u << (id & 31); return (k3a_or(&lmark[(stamp % 3) * (kHL / 32) + (id >> 5)], bit) & bit) == 0; }
    return k3a_exch(&g[id - kHL].stamp, stamp) != stamp;
  }
  __device__ __forceinline__ void clear(int id) const {
    if (id < kHL) { lkey[id] = kEmpty; lcost[id] = kEncMax; ltok[id] = -1; }
    else { Slot *q = &g[id - kHL]; K3_AST(&q->cost, kEncMax); K3_AST(&q->stamp, 0); K3_AST(&q->tok, -1); K3_AST(&q->key, kEmpty); }
  }
};

// exact k-th smallest (0-based) of the block's keys -- the value std::nth_element leaves at position k.  for_keys(fn) calls
// fn(valid, key) the same number of times in every thread of a wavefront (valid = false pads the tail); MSB-first radix
// select, 8 bits per pass; the digit is located by a 64-lane scan of the 256-bin histogram (4 bins per lane).
template <typename ForKeys>
__device__ __forceinline__ unsigned block_select_kth(ForKeys &&for_keys, int k, Shared &sh) {
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    __syncthreads();
    for (int i = tid; i < 256; i += kBlock) sh.hist[i] = 0;
    __syncthreads();
    for_keys([&](bool v, unsigned key) {
      v = v && (key & mask) == prefix;
      const int d = (key >> shift) & 255;
      const unsigned long long mv = __ballot(v);
      const int d0 = __builtin_amdgcn_readlane(d, mv ? __ffsll((long long)mv) - 1 : 0);      // (the index is wave-uniform)
      const unsigned long long diff = __ballot(v && d != d0);
      if (mv != 0) {
        if (diff == 0) { if (lane == __ffsll((long long)mv) - 1) k3a_add(&sh.hist[d0], __popcll(mv)); }
        else if (v) k3a_add(&sh.hist[d], 1);
      }
    });
    __syncthreads();
    if (tid < 64) {
      const int c0 = sh.hist[4 * lane], c1 = sh.hist[4 * lane + 1], c2 = sh.hist[4 * lane + 2], c3 = sh.hist[4 * lane + 3];
      int incl = c0 + c1 + c2 + c3;
      const int own = incl;
      incl = wave_incl_sum_i32(incl);
      const int excl = incl - own;
      const unsigned long long hit = __ballot(k >= excl && k < incl);
      const int owner = hit ? __ffsll((long long)hit) - 1 : 63;        // k beyond the total (cannot happen for k < n): last bin, like a serial scan would
      if (lane == owner) {
        int d = 4 * lane, cum = excl;
        if (hit) { if (k >= cum + c0) { cum += c0; d++; if (k >= cum + c1) { cum += c1; d++; if (k >= cum + c2) { cum += c2; d++; } } } }
        else { d = 255; cum = incl - c3; }
        sh.sel_digit = d; sh.sel_k = k - cum;
      }
    }
    __syncthreads();
    prefix |= (unsigned)sh.sel_digit << shift; mask |= 0xFFu << shift; k = sh.sel_k;
  }
  return prefix;
}

// A lane's pool record is the same for every thread of its workgroup: pin it to scalar registers (read through LDS or a vector load it would sit in 16 vector registers per thread,
// in kernels that have none to spare, and every pool access would form its address on the vector unit)
template <typename T> __device__ __forceinline__ T *k3_uniform_ptr(T *q) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long k3_uniform_i64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v),
      hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ LanePool k3_uniform_pool(const LanePool &a) {
  return LanePool{k3_uniform_ptr(a.tok_state), k3_uniform_ptr(a.tok_cost), k3_uniform_ptr(a.tok_extra), k3_uniform_ptr(a.newidx), k3_uniform_ptr(a.links),
      k3_uniform_ptr(a.link_arc), k3_uniform_i64(a.tcap), k3_uniform_i64(a.lcap)};
}

// Move lane L to pools in which at least need_t tokens and need_l links fit behind the n_tok tokens / n_link links it holds: a block of the spare arena with both capacities at
// least doubled (bump allocation, one device-scope atomic), the lane's contents copied by its own workgroup, the lane's record updated for the kernels that follow.  Called by all
// threads of the workgroup at a point where nothing of the frame being built is in the pools yet.  false: the arena is exhausted (the caller reports K3_ERR_OVERFLOW).
__device__ __forceinline__ bool grow_lane_pools(const DecParams &p, int L, LanePool &lp, long long n_tok, long long n_link, long long need_t, long long need_l, LanePool *s_new) {
  const int tid = threadIdx.x, nthr = (int)blockDim.x;
  __syncthreads();
  if (tid == 0) {
    LanePool np = lp; long long nt = 2 * lp.tcap, nl = 2 * lp.lcap;
    while (nt - n_tok < need_t) nt *= 2;
    while (nl - n_link < need_l) nl *= 2;
    const unsigned long long bt = ((unsigned long long)nt * 4 + 255) & ~255ull, bl4 = ((unsigned long long)nl * 4 + 255) & ~255ull,
        bl16 = ((unsigned long long)nl * 16 + 255) & ~255ull;
    const unsigned long long bytes = 4 * bt + bl4 + bl16;
    np.tcap = -1;
    if (p.spare && nt < (1ll << 31)) {
      // reserve with a compare-and-swap: a request that does not fit leaves the cursor where it was, so a later, smaller growth of another lane still succeeds
      // (a fetch-add before the capacity check pushed the cursor past the arena for good)
      unsigned long long off = __hip_atomic_load(p.spare_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bool got = false;
      while (off + bytes <= (unsigned long long)p.spare_bytes) {
        if (__hip_atomic_compare_exchange_strong(p.spare_used, &off, off + bytes, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { got = true; break; }
      }
      if (got) {
        char *b = p.spare + off;
        np.links = reinterpret_cast<Link *>(b); b += bl16; np.tok_state = reinterpret_cast<int *>(b); b += bt; np.tok_cost = reinterpret_cast<unsigned *>(b); b += bt;
        np.tok_extra = reinterpret_cast<float *>(b); b += bt; np.newidx = reinterpret_cast<int *>(b); b += bt; np.link_arc = reinterpret_cast<int *>(b);
        np.tcap = nt; np.lcap = nl;
      }
    }
    *s_new = np;
  }
  __syncthreads();
  const LanePool np = k3_uniform_pool(*s_new);
  if (np.tcap < 0) return false;
  for (long long i = tid; i < n_tok; i += nthr) { np.tok_state[i] = lp.tok_state[i]; np.tok_cost[i] = K3_ALD(&lp.tok_cost[i]); }
  { const int4 *src = reinterpret_cast<const int4 *>(lp.links); int4 *dst = reinterpret_cast<int4 *>(np.links); for (long long i = tid; i < n_link; i += nthr) dst[i] = src[i]; }
  for (long long i = tid; i < n_link; i += nthr) np.link_arc[i] = lp.link_arc[i];
  __threadfence();
  __syncthreads();
  if (tid == 0) { p.pools[L] = np; p.info[L].pool_grows += 1; }
  lp = np;
  __syncthreads();
  return true;
}

// ---- epsilon closure + eps links + frame finalisation for the frame being built (tokens [nb, nb + n_next)) ----
// ProcessNonemitting (lattice-faster-decoder.cc:830-897): relax eps arcs until no cost changes, with tot < cutoff;
// the links a token ends up with are exactly its eps arcs with cur + graph < cutoff at its final cost.
template <bool kPublish = true>
__device__ __forceinline__ void finish_frame(const DecParams &p, Shared &sh, const Table &tb, float cutoff, long long nb, int *tok_state, unsigned *tok_cost,
                                             Link *links, int *link_arc, long long tcap, long long lcap, int *tok_slot, int *wl, unsigned short (*lwl)[kWlLds],
                                                 unsigned (&creg)[kCurRegs], int (&sreg)[kCurRegs],
                                             long long &t_last__, unsigned &cnt_eps) {
  const int tid = threadIdx.x, lane = tid & 63;
  // One barrier per round.  Work-list counters n_wl[3], error flags err_r[4] and LDS mark bits [3] rotate: round r reads list r,
  // appends to list r + 1 (count n_wl[(r+1) % 3], marks (r+1) % 3, data buffer (r+1) & 1) and recycles the slots of round r + 2,
  // which nobody touches any more; what is read after a round's barrier (its error flag, the next count) is not written again
  // before the following barrier, so every wavefront takes the same decision.
  // (the counters, flags and marks were reset before pass 2, which fills list 1 = the frame's tokens whose state has eps arcs)
  K3_T(7);
  int n = sh.n_wl[1];
  for (int round = 1; n > 0; round++) {
    if (round > 100000) { sh.err = K3_ERR_HIP; break; }        // an epsilon cycle with negative weight: cannot converge
    const int cur = round & 1, nxt_buf = cur ^ 1;
    const int *wl_cur = wl + (long long)cur * p.frame_tokens_cap;
    int *wl_nxt = wl + (long long)nxt_buf * p.frame_tokens_cap;
    int *n_nxt = &sh.n_wl[(round + 1) % 3];
    auto fail = [&](int code) { sh.err = code; sh.err_r[round & 3] = 1; };
    K3_T(11);
    // Software pipeline over the work-list (it matters for frames with thousands of tokens): while item i is expanded, the
    // arc range of item i + kBlock is already on its way and the slot of item i + 2 kBlock is being read, so an iteration
    // exposes one memory round trip (the arcs) instead of three.
    auto fetch_slot = [&](int i) {
      int s_ = -1;
      if (i < n) {
        s_ = i < kWlLds ? (int)lwl[cur][i] : 0xFFFF; if (s_ == 0xFFFF) s_ = wl_cur[i];
      }
      return s_;
    };
    struct Pending { int ay, bx, ti; unsigned cb, prev; };      // what stage B requested; beg / deg follow once it has arrived
    auto request = [&](int slot) {
      Pending q{0, 0, 0, 0u, 0u};
      if (slot >= 0) {
        q.cb = tb.cost(slot); q.prev = q.cb;
        if (dec(q.cb) < cutoff) {
          const int st = tb.key(slot); q.ti = tb.tok(slot);
          const int2 a = p.offs[st], b = p.offs[st + 1]; q.ay = a.y; q.bx = b.x;
          // a token is expanded once per cost value: tok_cost holds the cost of its latest expansion until the frame is published
          q.prev = k3a_exch(&tok_cost[nb + q.ti], q.cb);
        }
      }
      return q;
    };
    Pending pend = request(fetch_slot(tid));
    int slot_next = fetch_slot(tid + kBlock);
    for (int i0 = 0; i0 < n; i0 += kBlock) {
      const Pending pn = request(slot_next);
      slot_next = fetch_slot(i0 + 2 * kBlock + tid);
      K3_TW(13);
      const unsigned cb = pend.cb; const int ti = pend.ti;
      int beg = 0, deg = 0;
      if (pend.prev != cb) { beg = pend.ay; deg = pend.bx - pend.ay; }
      pend = pn;
      wave_expand(p.arcs, beg, deg, [&](bool valid, int arc, int owner, const ArcRec &r) {
        const unsigned ocb = __shfl(cb, owner); const int oti = __shfl(ti, owner);
        const float oc = dec(ocb);
        bool claimed = false, push = false, mk = false; int slot2 = -1, nxt = 0; float tot = 0.0f;
        cnt_eps += valid;
        if (valid) {
          tot = oc + r.w; nxt = (int)((unsigned)r.next & ~kEpsFlag);
          if (tot < cutoff) {
            slot2 = tb.claim(nxt, &claimed);
            if (slot2 < 0) { fail(K3_ERR_OVERFLOW); claimed = false; }
            else {
              mk = true;
              const unsigned e = enc(tot);
              const unsigned old = tb.cost_min(slot2, e);
              if (e < old && r.next < 0) push = tb.mark(slot2, round + 1);      // only tokens whose state has eps arcs are queued
            }
          }
        }
        int idx = wave_append(claimed, &sh.n_next);
        if (claimed) {
          if (idx < p.frame_tokens_cap && nb + idx < tcap) { tok_slot[idx] = slot2; tok_state[nb + idx] = nxt; K3_AST(&tok_cost[nb + idx], kEncMax); }
          else { fail(K3_ERR_OVERFLOW); idx = 0; }      // (the callers make room for a whole frame before it starts: the pool cannot be what is full)
          tb.set_tok(slot2, idx);
        }
        const int pos = wave_append(push, n_nxt);
        if (push) {
          if (pos < kWlLds) lwl[nxt_buf][pos] = slot2 < kHL ? (unsigned short)slot2 : (unsigned short)0xFFFF;
          if (pos >= kWlLds || slot2 >= kHL) { if (pos < p.frame_tokens_cap) wl_nxt[pos] = slot2; else fail(K3_ERR_OVERFLOW); }
        }
        // the eps link of this arc at the source's present cost; links written at a cost the source later improves on
        // are recognised as stale by their stamp (Link::ac of an eps link = the source cost it was created at)
        if (mk && !claimed) { idx = tb.wait_tok(slot2, &sh.err); if (idx < 0) { fail(K3_ERR_HIP); idx = 0; } }
        const long long lp = wave_append64(mk, &sh.n_link);
        if (mk) {
          if (lp < lcap) {
            store_link(&links[lp], Link{(unsigned)(nb + oti), (unsigned)(nb + idx), tot, __uint_as_float(ocb)});
            store_stream(&link_arc[lp], arc);
          } else fail(K3_ERR_OVERFLOW);
        }
      });
      K3_TW(14);
    }
    // recycle the slots round + 2 will append to / flag / mark
    if (tid == 0) { sh.n_wl[(round + 2) % 3] = 0; sh.err_r[(round + 2) & 3] = 0; }
    for (int i = tid; i < kHL / 32; i += kBlock) tb.lmark[((round + 2) % 3) * (kHL / 32) + i] = 0;
    __syncthreads();
    K3_T(15);
    if (sh.err_r[round & 3]) break;
    n = sh.n_wl[(round + 1) % 3];
  }
  if (block_err(sh)) return;
  K3_T(9);
  if (!kPublish) return;      // literal_order publishes the frame itself (it still needs the table)
  // final costs into the pool, clear the table
  {
    const int n = sh.n_next;
    if (n <= kCurRegs * kBlock) {       // the next frame starts from these registers instead of re-reading the pool
#pragma unroll
      for (int k = 0; k < kCurRegs; k++) {
        const int i = tid + k * kBlock;
        if (i < n) { const int slot = tok_slot[i]; const unsigned c = tb.cost(slot); creg[k] = c; sreg[k] = tb.key(slot); tok_cost[nb + i] = c; if (slot >= kHL) tb.clear(slot); }
      }
    } else {
      for (int i0 = tid; i0 < n; i0 += 4 * kBlock) {      // four slot reads in flight per thread (frames of this size pay a round trip per iteration)
        int sl[4]; unsigned c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int i = i0 + j * kBlock; sl[j] = i < n ? tok_slot[i] : -1; }
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = sl[j] >= 0 ? tb.cost(sl[j]) : 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int i = i0 + j * kBlock; if (sl[j] >= 0) { tok_cost[nb + i] = c[j]; if (sl[j] >= kHL) tb.clear(sl[j]); } }
      }
    }
    __syncthreads();
    for (int i = tid; i < kHL; i += kBlock) { tb.lkey[i] = kEmpty; tb.lcost[i] = kEncMax; tb.ltok[i] = -1; }
  }
  (void)lane;
  __syncthreads();
  K3_T(10);
}

}  // namespace
#endif  // K3_DECODER_DEV_H_
