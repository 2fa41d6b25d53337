// k3_nnet.hip -- nnet3 TDNN / TDNN-F forward for gfx950: one fused MFMA kernel per affine layer.
//
//   out[r, :] = epilogue( sum_{o in offsets} in[row(r) + shift_o, :] * W_o^T )
//
// * The time-stride splice of TdnnComponent (nnet3/nnet-tdnn-component.cc:181-211: one GEMM per offset on a
//   strided sub-matrix view) is never materialised: the A-tile loader walks K over (offset, column) pairs and
//   applies the row shift on the fly, so a 2-offset TDNN-F half-layer is ONE GEMM with K = 2*D.
// * GEMMs run on the FP32 matrix core (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf chain)
//   -- the only MFMA class that meets the 1e-4 log-likelihood parity bound; 157.3 TFLOP/s peak.
// * Bias, ReLU, test-mode BatchNorm (y = x*scale + offset, nnet-normalize-component.cc:460-462) and the
//   0.75-scaled bypass Sum(Scale(0.75, x), y) are executed in the GEMM epilogue on the accumulators.
// * Whole utterances are evaluated at once (no chunking): edge frames are replicated by CLAMPING the row
//   index of the first layer's loads (DecodableNnetSimple, nnet-am-decodable-simple.cc:154-163), every
//   layer computes exactly the arithmetic progression of time steps its consumers need (the compiler's
//   time-step pruning), rows are utterance-major so ragged batches need no padding.
// * Tiling: 256 threads = 4 wavefronts; block tile 128 x {128|96} x 32, LDS rows padded to 36 floats so
//   ds_read_b128 fragment loads and ds_write_b128 stages are bank-conflict free; K is permuted inside each
//   8-wide group so one b128 read feeds 4 consecutive MFMAs; global->register->LDS double buffering;
//   blockIdx is remapped so that the N-tiles sharing an A panel run on the same XCD (shared L2).
#include "k3_common.h"
#include <mutex>
#include "k3_nnet_model.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

// developer-only timing / short-circuit switches of the GEMM kernel: compiled in only with -DK3_GEMM_PROF (never in the shipped library)
#ifdef K3_GEMM_PROF
#define K3_GDBG (p.dbg)
#else
#define K3_GDBG 0
#endif
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kBK = 32, kLdsLd = 36;
constexpr int kMaxOffsets = 8, kMaxOps = 6;

struct TileDesc {      // one per 128-row slab of a node's output.  A slab is cut from ONE sequence (utterance / chunk), or -- when a sequence ends inside
  // it -- continues with the first rows of the next one (rows [split, nrows)): sequences of 1030 or 340 rows would otherwise leave every ninth / third
  // slab almost empty, and an empty row costs the same MFMA time as a full one.
  int out_row0, nrows; // rows [out_row0, out_row0 + nrows) of the output buffer (consecutive across the two sequences)
  int split;           // local rows >= split belong to the second sequence (== nrows: none)
  int in_base;         // input row of local index 0 with shift 0
  int in_lo, in_hi;    // clamp bounds (absolute input rows) -- edge-frame replication for the first layer
  int res_base;        // residual row of local index 0
  int bias_row;        // row of GemmParams::seq_bias this slab starts from instead of the bias (a node fed by the chunk's i-vector)
  int in_base2, in_lo2, in_hi2, res_base2, bias_row2;      // the same for the second sequence; the bases are pre-shifted so that local row r maps to base2 + r * stride as well
  int pad[3];
};

struct GemmParams {
  const float *A; long long lda; int in_dim, noff, row_stride; int shifts[kMaxOffsets];
  int tiles_per_off, tiles_per_seg;   // k-tiles per time offset (0: offsets are not tile aligned) / per accumulation segment
  const float *W; int ldw, Ktot;
  float *C; long long ldc; int N;
  const float *bias;
  const float *seq_bias; long long ld_seq_bias;      // non-null: per-sequence rows "bias + W_iv . ivector" (k3_seq_bias_kernel), picked by TileDesc::bias_row
  int nops; int op_kind[kMaxOps]; const float *op_scale[kMaxOps]; const float *op_offset[kMaxOps];
  const float *R; long long ldr; int res_row_stride; float res_scale;
  const TileDesc *tiles; int num_m_tiles, num_n_tiles; int dbg; long long *dbg_buf;
  // split-bf16 with producer-side planes (k3_nnet_batch_set_precision(.., 2)): the three bf16 planes [3][rows][ld] of the input (same rows and ld as A; null: the loader splits A)
  // and of the output (same rows and ld as C; null: not wanted), plane strides in elements
  const unsigned short *Ap; long long ap_stride; unsigned short *Cp; long long cp_stride;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// kAligned: every k-tile lies inside one time offset and inside K (tiles_per_off > 0) -- a separate instantiation, because a
// runtime choice between the two loaders makes the compiler merge their registers with moves that wait for the loads at once.
// EPI: the epilogue program known at compile time -- 0: any (run-time dispatch per op), 1: ReLU, scale/offset, residual (a TDNN-F
// affine + ReLU + BatchNorm + bypass), 2: none (linear bottleneck), 3: ReLU, scale/offset.  The fixed programs are straight-line
// code; the run-time dispatch costs a register shuffle per op when it merges the branches.
// kEpiAnyMap: kEpiAny + the sigmoid / tanh operations (their own instantiation: the exp code costs the generic one registers)
enum { kEpiAny = 0, kEpiReluScaleRes = 1, kEpiNone = 2, kEpiReluScale = 3, kEpiAnyMap = 4 };
// SigmoidComponent / TanhComponent: the overflow-safe forms of matrix/kaldi-vector.cc:900-960
__device__ __forceinline__ float epi_sigmoid(float x) { if (x > 0.0f) return 1.0f / (1.0f + expf(-x)); const float e = expf(x); return e / (e + 1.0f); }
__device__ __forceinline__ float epi_tanh(float x) {
  if (x > 0.0f) {
    const float e = expf(-x);
    return -1.0f + 2.0f / (1.0f + e * e);
  }
  const float e = expf(x);
  return 1.0f - 2.0f / (1.0f + e * e);
}
template <int BN, int WM, int WN, bool kAligned, int EPI>
__global__ __launch_bounds__((kBM / WM) * (BN / WN) * 64, (kBM / WM) * (BN / WN) == 8 ? 4 : 2) void k3_tdnn_gemm_kernel(GemmParams p) {
  constexpr int NT = (kBM / WM) * (BN / WN) * 64, LR = NT / 8;      // threads per workgroup (4 or 8 wavefronts); rows the loader covers per pass
  constexpr int MI = WM / 32, NI = WN / 32, WAVES_N = BN / WN;
  constexpr int A_LOADS = kBM * kBK / 4 / NT;   // float4 loads per thread per k-tile
  constexpr int B_LOADS = BN * kBK / 4 / NT;
  static_assert(kBM % LR == 0 && BN % LR == 0, "loader passes must tile the block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *As = reinterpret_cast<float *>(smem);                         // [2][kBM][kLdsLd]
  float *Bs = As + 2 * kBM * kLdsLd;                                   // [2][BN][kLdsLd]

  // XCD-aware remap (T1): the dispatcher places block b on XCD b % 8; give each XCD a contiguous logical range
  const int nblocks = p.num_m_tiles * p.num_n_tiles;
  int bid = blockIdx.x;
  { const int per = nblocks / 8; if (bid < per * 8) bid = (bid % 8) * per + bid / 8; }
  const int m_tile = bid / p.num_n_tiles, n_tile = bid % p.num_n_tiles;
  const TileDesc td = p.tiles[m_tile];
  const int n0 = n_tile * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int ld_row = tid >> 3, ld_kv = (tid & 7) * 4;

  // per-thread row bookkeeping for the A loader (rows beyond nrows re-read the last valid row; never stored)
  int a_row_local[A_LOADS], a_lo[A_LOADS], a_hi[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; i++) {
    const int r = min(i * LR + ld_row, td.nrows - 1); const bool s2 = r >= td.split;
    a_row_local[i] = r * p.row_stride + (s2 ? td.in_base2 : td.in_base); a_lo[i] = s2 ? td.in_lo2 : td.in_lo; a_hi[i] = s2 ? td.in_hi2 : td.in_hi;
  }

  f32x4 ra[A_LOADS], rb[B_LOADS];
  // `oi_u` = time offset the k-tile lies in when offsets are tile aligned (tiles_per_off > 0): uniform over the block, so the
  // row shift is picked with scalar selects.  (Indexing the kernel-argument array with a per-lane value costs a dependent
  // global load at the top of every k-tile, and the waits the compiler puts around it serialise the whole tile prefetch.)
  const float *a_ptr[A_LOADS];      // kAligned: this thread's A addresses for the next tile; recomputed when the time offset changes
  auto load_tiles = [&](int kt, int oi_u, int w_u, bool dummy = false) {      // dummy (block-uniform): the prefetch slot of the last k-tile -- nothing left to fetch
    const int kglob = kt * kBK + ld_kv;
    if constexpr (kAligned) {        // straight-line loads, nothing to wait for in between
      if (w_u == 0) {                // first tile of a time offset (block-uniform): row shift and clamped rows change
        int sh = p.shifts[0];
#pragma unroll
        for (int o = 1; o < kMaxOffsets; o++) sh = oi_u == o ? p.shifts[o] : sh;
#pragma unroll
        for (int i = 0; i < A_LOADS; i++) a_ptr[i] = p.A + (long long)clampi(a_row_local[i] + sh, a_lo[i], a_hi[i]) * p.lda + ld_kv;
      }
#pragma unroll
      // (dummy: every lane the same 16 bytes -- one cache line per instruction, and no address past the last row)
      for (int i = 0; i < A_LOADS; i++) {
        ra[i] = *reinterpret_cast<const f32x4 *>(dummy ? p.W : a_ptr[i]);
        a_ptr[i] += kBK;
      }
    } else {
      const bool kvalid = kglob < p.Ktot;
      const int oi = kglob / p.in_dim, col = kglob - oi * p.in_dim;
      const int shift = kvalid ? p.shifts[oi] : 0;
#pragma unroll
      for (int i = 0; i < A_LOADS; i++) {
        if (kvalid) {
          const int row = clampi(a_row_local[i] + shift, a_lo[i], a_hi[i]);
          ra[i] = *reinterpret_cast<const f32x4 *>(p.A + (long long)row * p.lda + col);
        } else {
          ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; i++)   // W is zero-padded to [Npad x Kpad]: no bounds checks
      rb[i] = *reinterpret_cast<const f32x4 *>(dummy ? p.W : p.W + (long long)(n0 + i * LR + ld_row) * p.ldw + kglob);
  };
  // LDS K layout: inside every group of 8 k-values position p holds k = 2 * (p & 3) + (p >> 2), so that the b128 fragment a
  // lane of half h = lane >> 5 reads (positions 4h .. 4h+3) is k = h, 2+h, 4+h, 6+h and MFMA j (A column k = lane >> 5)
  // consumes k = 2j, 2j+1: the accumulation runs over k in ASCENDING order, the order a CPU sgemm kernel sums in -- with a
  // different order every partial sum rounds differently and the result drifts ~5x further from the reference.
  // A thread holding k..k+3 therefore writes (k, k+2) and (k+1, k+3) as two 8-byte pieces.
  const int st_col = (ld_kv & ~7) + ((ld_kv >> 2) & 1) * 2;
  auto store_tiles = [&](int buf) {
    float *a = As + buf * kBM * kLdsLd, *b = Bs + buf * BN * kLdsLd;
#pragma unroll
    for (int i = 0; i < A_LOADS; i++) {
      float *q = a + (i * LR + ld_row) * kLdsLd + st_col;
      q[0] = ra[i][0]; q[4] = ra[i][1]; q[1] = ra[i][2]; q[5] = ra[i][3];       // two ds_write2_b32 straight from the load registers
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; i++) {
      float *q = b + (i * LR + ld_row) * kLdsLd + st_col;
      q[0] = rb[i][0]; q[4] = rb[i][1]; q[1] = rb[i][2]; q[5] = rb[i][3];
    }
  };

  // Association mirrors the reference (out = bias; out += in_o . W_o^T per time offset, nnet-tdnn-component.cc:199-207, each
  // an sgemm that sums k in ascending order inside blocks of K and adds the block into C): `tot` starts at the bias, `acc` is
  // an ascending-k fma chain over one segment (a time offset, cut every tiles_per_seg k-tiles) added into `tot` at its end.
  f32x16 acc[MI][NI], tot[MI][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ni++) {
    const int col = n0 + wn * WN + ni * 32 + (lane & 31);
    const float *bias = p.seq_bias ? p.seq_bias + (long long)td.bias_row * p.ld_seq_bias : p.bias;
    const float b0 = (bias && col < p.N) ? bias[col] : 0.0f;
    if (p.seq_bias && td.split < td.nrows) {      // two sequences in the slab, each with its own "bias + W_iv . ivector" row (block-uniform branch)
      const float b1 = col < p.N ? p.seq_bias[(long long)td.bias_row2 * p.ld_seq_bias + col] : 0.0f;
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[mi][ni][r] = 0.0f; tot[mi][ni][r] = (wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) < td.split ? b0 : b1; }
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[mi][ni][r] = 0.0f; tot[mi][ni][r] = b0; }
    }
  }

  const int nk = (p.Ktot + kBK - 1) / kBK;
  long long t0 = 0, t1 = 0, t2 = 0;
  if (K3_GDBG & 8) t0 = (long long)__builtin_readcyclecounter();
  int oi_next = 0, w_next = 0;       // time offset of tile kt + 1 and its index inside the offset (tiles_per_off > 0)
  load_tiles(0, 0, 0);
  store_tiles(0);
  __syncthreads();
  if (K3_GDBG & 8) t1 = (long long)__builtin_readcyclecounter();
  const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
  for (int kt = 0; kt < ((K3_GDBG & 4) ? 1 : nk); kt++) {
    const int buf = kt & 1;
    // prefetch tile kt + 1 (the last iteration issues dummy loads of one line: no branch in the loop body, so the compiler's
    // s_waitcnt placement stays exact -- with conditional prefetches it put vmcnt(0) between the loads of one tile)
    const int w_cur = w_next, oi_cur = oi_next;        // index of tile kt inside its offset
    const bool more = kt + 1 < nk;
    if (++w_next == p.tiles_per_off) { w_next = 0; oi_next++; }
    const float *a = As + buf * kBM * kLdsLd + (wm * WM + frag_row) * kLdsLd + frag_k;
    const float *b = Bs + buf * BN * kLdsLd + (wn * WN + frag_row) * kLdsLd + frag_k;
    // fragments of step kk + 1 are requested before the MFMAs of step kk: a wavefront issues in order, so reads placed after the 16 MFMAs of a step
    // reach the LDS only when the last of them has been accepted, and the pipe then idles for the LDS round trip (a lone workgroup ran at 72 %)
    // The stretch between the last MFMA of one k-tile and the first of the next is what a workgroup cannot hide by itself, so only the barrier and one
    // LDS read stay on it: the next tile's global loads are issued while the first fragments are on their way, and the tile is written to the other
    // buffer before the LAST step's MFMAs (its loads were issued three steps earlier), not after them.
    f32x4 fa[2][MI], fb[2][NI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++) fa[0][mi] = *reinterpret_cast<const f32x4 *>(a + mi * 32 * kLdsLd);
#pragma unroll
    for (int ni = 0; ni < NI; ni++) fb[0][ni] = *reinterpret_cast<const f32x4 *>(b + ni * 32 * kLdsLd);
    __builtin_amdgcn_sched_barrier(0);
    load_tiles(more ? kt + 1 : kt, more ? oi_next : oi_cur, more ? w_next : 1, !more);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < kBK / 8; kk++) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < kBK / 8) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++) fa[nxt][mi] = *reinterpret_cast<const f32x4 *>(a + mi * 32 * kLdsLd + (kk + 1) * 8);
#pragma unroll
        for (int ni = 0; ni < NI; ni++) fb[nxt][ni] = *reinterpret_cast<const f32x4 *>(b + ni * 32 * kLdsLd + (kk + 1) * 8);
      }
      __builtin_amdgcn_sched_barrier(0);      // reads first, then this step's MFMAs (the scheduler otherwise sinks the reads to the end of the step)
      if (kk == kBK / 8 - 1) { store_tiles(buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][mi][j], fb[cur][ni][j], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);      // (keeps the scheduler from sinking the next step's reads back below these MFMAs)
    }
    {   // end of an accumulation segment?
      bool flush = kt + 1 == nk;
      if (p.tiles_per_off > 0) { const int w = w_cur + 1; flush = flush || w == p.tiles_per_off || w % p.tiles_per_seg == 0; }
      if (flush) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int r = 0; r < 16; r++) { tot[mi][ni][r] += acc[mi][ni][r]; acc[mi][ni][r] = 0.0f; }
      }
    }
    __syncthreads();
  }

  if (K3_GDBG & 8) t2 = (long long)__builtin_readcyclecounter();
  // ---- epilogue.  The MFMA C/D layout gives a lane 4 consecutive ROWS of one column (col = lane & 31,
  // row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)), i.e. dword stores of 128-byte row pieces: 64 store and 64 residual-load
  // instructions per lane, which is what the affine layers (K = 192 only) spent 60 % of their time on.  Instead each wavefront
  // transposes its WM x WN accumulator tile through LDS (the tile buffers are dead by now) and then works on float4 row
  // pieces: 4x fewer, 4x wider memory instructions, 256-byte runs per row.
  constexpr int kStLd = WN + 4, C4 = WN / 4, RPI = 64 / C4, ITERS = WM / RPI;      // RPI rows per iteration; lanes >= RPI * C4 idle (WN = 96: 48 of 64 busy)
  float *stage = reinterpret_cast<float *>(smem) + wave * (WM * kStLd);
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        stage[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * kStLd + ni * 32 + (lane & 31)] = tot[mi][ni][r];
  int res_kind = -1;
#pragma unroll
  for (int o = 0; o < kMaxOps; o++) if (o < p.nops && p.op_kind[o] == k3::kEpiResidual) res_kind = o;
  const float *__restrict__ R = p.R; float *__restrict__ C = p.C;
  const bool vec_ok = (p.N % 4 == 0) && (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                      (res_kind < 0 || ((p.ldr % 4 == 0) && ((reinterpret_cast<uintptr_t>(R) & 15) == 0)));
  // Everything the per-element program needs is fetched up front, all loads in flight together: the residual row pieces of the
  // tile and, per epilogue op, this lane's four columns of its scale / offset vectors (a lane keeps the same columns over all
  // its rows).  The op program is then applied op by op over the whole register tile: six block-uniform dispatches per tile
  // instead of per row piece, and no scalar / vector loads between the arithmetic.  (Fetching kinds, pointers and vectors
  // inside the row loop cost 42 k of the 48 k epilogue cycles of a K = 192 layer.)
  const int c4 = lane % C4, row0 = lane / C4;                 // row = it * RPI + row0, column group c4
  static_assert(WM % RPI == 0, "rows per iteration must divide the wave tile");
  const bool lane_on = lane < RPI * C4;
  const int col = lane_on ? n0 + wn * WN + c4 * 4 : p.N;      // idle lanes look like out-of-range columns
  if (vec_ok) {
    f32x4 res[ITERS], opS[kMaxOps], opO[kMaxOps], v[ITERS];
    const bool col_ok = col < p.N;
    const int colc = min(col, p.N - 4);
    const int row0c = lane_on ? row0 : 0;
    // a full tile (all of the 128 rows and BN columns exist: every tile but the last of an utterance / of N) needs no per-row
    // bounds logic and walks its rows with one pointer increment per row piece
    const bool full = td.nrows == kBM && n0 + BN <= p.N;
    if ((EPI == kEpiAny || EPI == kEpiAnyMap || EPI == kEpiReluScaleRes) && res_kind >= 0 && !(K3_GDBG & 1)) {
      if (full && td.split >= td.nrows) {
        const float *rp = R + (long long)(td.res_base + (wm * WM + row0c) * p.res_row_stride) * p.ldr + colc;
        const long long rstep = (long long)RPI * p.res_row_stride * p.ldr;
#pragma unroll
        for (int it = 0; it < ITERS; it++) res[it] = *reinterpret_cast<const f32x4 *>(rp + it * rstep);
      } else {
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
          const int lrow = min(wm * WM + it * RPI + row0, td.nrows - 1);
          res[it] = *reinterpret_cast<const f32x4 *>(R + (long long)((lrow < td.split ? td.res_base : td.res_base2) + lrow * p.res_row_stride) * p.ldr + colc);
        }
      }
    }
    if (EPI == kEpiAny || EPI == kEpiAnyMap) {
#pragma unroll
      for (int o = 0; o < kMaxOps; o++) {
        if (o < p.nops && p.op_kind[o] == k3::kEpiScaleOffset) {
          opS[o] = *reinterpret_cast<const f32x4 *>(p.op_scale[o] + colc); opO[o] = *reinterpret_cast<const f32x4 *>(p.op_offset[o] + colc);
        }
      }
    } else if (EPI == kEpiReluScaleRes || EPI == kEpiReluScale) {
      opS[1] = *reinterpret_cast<const f32x4 *>(p.op_scale[1] + colc); opO[1] = *reinterpret_cast<const f32x4 *>(p.op_offset[1] + colc);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; it++) v[it] = *reinterpret_cast<const f32x4 *>(stage + (it * RPI + row0c) * kStLd + c4 * 4);
    if (EPI == kEpiAny || EPI == kEpiAnyMap) {
#pragma unroll
      for (int o = 0; o < kMaxOps; o++) {
        if (o < p.nops) {
          const int kind = p.op_kind[o];
          if (kind == k3::kEpiRelu) {
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
              v[it][0] = fmaxf(v[it][0], 0.0f);
              v[it][1] = fmaxf(v[it][1], 0.0f);
              v[it][2] = fmaxf(v[it][2], 0.0f);
              v[it][3] = fmaxf(v[it][3], 0.0f);
            }
          } else if (kind == k3::kEpiScaleOffset) {
#pragma unroll
            for (int it = 0; it < ITERS; it++) v[it] = v[it] * opS[o] + opO[o];
          } else if (EPI == kEpiAnyMap && (kind == k3::kEpiSigmoid || kind == k3::kEpiTanh)) {
#pragma unroll
            for (int it = 0; it < ITERS; it++)
#pragma unroll
              for (int e = 0; e < 4; e++) v[it][e] = kind == k3::kEpiSigmoid ? epi_sigmoid(v[it][e]) : epi_tanh(v[it][e]);
          } else {
#pragma unroll
            for (int it = 0; it < ITERS; it++) v[it] = p.res_scale * res[it] + v[it];
          }
        }
      }
    } else if (EPI == kEpiReluScaleRes || EPI == kEpiReluScale) {
#pragma unroll
      for (int it = 0; it < ITERS; it++) {
        v[it][0] = fmaxf(v[it][0], 0.0f); v[it][1] = fmaxf(v[it][1], 0.0f); v[it][2] = fmaxf(v[it][2], 0.0f); v[it][3] = fmaxf(v[it][3], 0.0f);
        v[it] = v[it] * opS[1] + opO[1];
        if (EPI == kEpiReluScaleRes) v[it] = p.res_scale * res[it] + v[it];
      }
    }
    if (full && !(K3_GDBG & 2)) {
      if (lane_on) {
        float *cp = C + (long long)(td.out_row0 + wm * WM + row0) * p.ldc + col;
        const long long cstep = (long long)RPI * p.ldc;
#pragma unroll
        for (int it = 0; it < ITERS; it++) *reinterpret_cast<f32x4 *>(cp + it * cstep) = v[it];
      }
    } else {
#pragma unroll
      for (int it = 0; it < ITERS; it++) {
        const int lrow = wm * WM + it * RPI + row0;
        if (col_ok && lrow < td.nrows && (!(K3_GDBG & 2) || v[it][0] == 12345.678f)) *reinterpret_cast<f32x4 *>(C + (long long)(td.out_row0 + lrow) * p.ldc + col) = v[it];
      }
    }
  } else {                                             // unaligned / odd-width output: element-wise tail path
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
      const int row = it * RPI + row0, lrow = wm * WM + row;
      if (col >= p.N || lrow >= td.nrows) continue;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + row * kStLd + c4 * 4);
      for (int e = 0; e < 4; e++) {
        const int c = col + e;
        if (c >= p.N) break;
        float x = v[e];
        for (int o = 0; o < p.nops; o++) {
          const int kind = p.op_kind[o];
          if (kind == k3::kEpiRelu) x = fmaxf(x, 0.0f);
          else if (kind == k3::kEpiScaleOffset) x = x * p.op_scale[o][c] + p.op_offset[o][c];
          else if (EPI == kEpiAnyMap && kind == k3::kEpiSigmoid) x = epi_sigmoid(x);
          else if (EPI == kEpiAnyMap && kind == k3::kEpiTanh) x = epi_tanh(x);
          else x = p.res_scale * R[(long long)((lrow < td.split ? td.res_base : td.res_base2) + lrow * p.res_row_stride) * p.ldr + c] + x;
        }
        C[(long long)(td.out_row0 + lrow) * p.ldc + c] = x;
      }
    }
  }
  if ((K3_GDBG & 8) && tid == 0) {
    const long long t3 = (long long)__builtin_readcyclecounter();
    atomicAdd((unsigned long long *)&p.dbg_buf[0], (unsigned long long)(t1 - t0)); atomicAdd((unsigned long long *)&p.dbg_buf[1], (unsigned long long)(t2 - t1));
    atomicAdd((unsigned long long *)&p.dbg_buf[2], (unsigned long long)(t3 - t2)); atomicAdd((unsigned long long *)&p.dbg_buf[3], 1ull);
  }
}

// ---- the same product on the bf16 matrix core, operands split three ways (exploratory; never the default) ----------------------------------------------------
// x = x_hi + x_mid + x_lo with every part a bf16 (the top, middle and bottom 8 bits of the float's 24-bit significand, by truncation: the split is EXACT), and
//   x * w ~= hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid            (the three dropped terms are below 2^-24 of the product)
// six v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block instead of eight v_mfma_f32_32x32x2_f32: 0.375 of the FP32 matrix-core time at gfx950's 16 : 1 rate, every bf16 x bf16
// product exact in the fp32 accumulator.  The weights are split once on the host (three planes), the activations when a tile is staged (and / sub / and / sub per element + one
// v_perm per pair and plane).  What it is NOT: the reference's ascending-k fp32 summation order -- it is as accurate as an fp32 GEMM but rounds differently, so it is held to the
// float64 forward (no further from it than nnet3-compute is), not to the 1e-4 fixture gates of the FP32 kernel above ("split-bf16").
// One LDS buffer of three planes per operand (rows of 32 bf16 padded to 80 bytes: b128 fragment reads and b64 / b128 stage writes conflict-free), two workgroups per CU; the next
// tile's global loads are in flight under the current tile's MFMAs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// the exact three-way split of a float into bf16 bit patterns (upper halves of h, m, l): hi = top 16 bits, mid = top 16 bits of (x - hi), lo = x - hi - mid (8 significant bits left)
__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
  const unsigned xb = __float_as_uint(x);
  const float r1 = x - __uint_as_float(xb & 0xFFFF0000u); const unsigned r1b = __float_as_uint(r1);
  const float r2 = r1 - __uint_as_float(r1b & 0xFFFF0000u);
  h = xb; m = r1b; l = __float_as_uint(r2);
}
// four consecutive floats of a row -> their three planes (8 bytes each) at element offset `at` of plane 0
__device__ __forceinline__ void store_planes4(unsigned short *Cp, long long cp_stride, long long at, const f32x4 &v) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; e++) split3(v[e], h[e], m[e], l[e]);
  // (v_perm_b32 selector 0x07060302: the upper halves of the two operands, the second operand's in the low half)
  *reinterpret_cast<uint2 *>(Cp + at) = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
  *reinterpret_cast<uint2 *>(Cp + cp_stride + at) = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
  *reinterpret_cast<uint2 *>(Cp + 2 * cp_stride + at) = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}
// an activation matrix produced by something else than the x6 kernel (the first layer, element-wise nodes), split after the fact: one thread per four columns
__global__ __launch_bounds__(256) void k3_split_planes_kernel(const float *__restrict__ C, long long ldc, long long rows, int cols4, unsigned short *Cp, long long cp_stride) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i >= rows * cols4) return;
  const long long r = i / cols4; const int c = (int)(i - r * cols4) * 4;
  store_planes4(Cp, cp_stride, r * ldc + c, *reinterpret_cast<const f32x4 *>(C + r * ldc + c));
}
// PA: the input comes as three bf16 planes (GemmParams::Ap, written by the producing epilogue): the loader only loads -- 6 bytes per element instead of 4 and no arithmetic
template <int BN, int WM, int WN, int EPI, bool PA>
__global__ __launch_bounds__(256, 2) void k3_tdnn_gemm_x6_kernel(GemmParams p, const unsigned short *__restrict__ Wp, long long plane_stride) {
  constexpr int NT = 256, LR = NT / 8, MI = WM / 32, NI = WN / 32, WAVES_N = BN / WN, A_LOADS = PA ? 1 : kBM * kBK / 4 / NT;
  constexpr int AP_LOADS = PA ? kBM * 4 * 3 / NT : 1;      // 16-byte chunks of the input tile: 3 planes x kBM rows x 4
  constexpr int kRow = 80, B_CH = BN * 4 * 3, B_LOADS = (B_CH + NT - 1) / NT;      // bytes per LDS row of a plane; 16-byte chunks of the weight tile (3 planes x BN rows x 4)
  static_assert((kBM / WM) * (BN / WN) == 4 && kBK == 32, "four wavefronts, k-tiles of 32");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *As = smem, *Bs = smem + 3 * kBM * kRow;      // [3][kBM][kRow], [3][BN][kRow]
  const int nblocks = p.num_m_tiles * p.num_n_tiles;
  int bid = blockIdx.x;
  { const int per = nblocks / 8; if (bid < per * 8) bid = (bid % 8) * per + bid / 8; }
  const int m_tile = bid / p.num_n_tiles, n_tile = bid % p.num_n_tiles;
  const TileDesc td = p.tiles[m_tile];
  const int n0 = n_tile * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int ld_row = tid >> 3, ld_kv = (tid & 7) * 4;
  constexpr int A_ROWS = PA ? 2 : A_LOADS;      // PA: a thread's chunks lie in two rows of the tile (tid >> 2 and 64 + (tid >> 2)), each in all three planes
  int a_row_local[A_ROWS], a_lo[A_ROWS], a_hi[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; i++) {
    const int r = min(PA ? i * 64 + (tid >> 2) : i * LR + ld_row, td.nrows - 1); const bool s2 = r >= td.split;
    a_row_local[i] = r * p.row_stride + (s2 ? td.in_base2 : td.in_base); a_lo[i] = s2 ? td.in_lo2 : td.in_lo; a_hi[i] = s2 ? td.in_hi2 : td.in_hi;
  }
  f32x4 ra[A_LOADS]; uint4 rb[B_LOADS]; uint4 rap[AP_LOADS];
  const float *a_ptr[A_LOADS]; const unsigned short *ap_ptr[A_ROWS];
  auto load_tiles = [&](int kt, int oi_u, int w_u, bool dummy = false) {
    if (w_u == 0) {
      int sh = p.shifts[0];
#pragma unroll
      for (int o = 1; o < kMaxOffsets; o++) sh = oi_u == o ? p.shifts[o] : sh;
      if (PA) {
#pragma unroll
        for (int i = 0; i < A_ROWS; i++) ap_ptr[i] = p.Ap + (long long)clampi(a_row_local[i] + sh, a_lo[i], a_hi[i]) * p.lda + (tid & 3) * 8;
      } else {
#pragma unroll
        for (int i = 0; i < A_LOADS; i++) a_ptr[i] = p.A + (long long)clampi(a_row_local[i] + sh, a_lo[i], a_hi[i]) * p.lda + ld_kv;
      }
    }
    if (PA) {
#pragma unroll
      for (int i = 0; i < AP_LOADS; i++) rap[i] = *reinterpret_cast<const uint4 *>(dummy ? Wp : ap_ptr[i & 1] + (i >> 1) * p.ap_stride);      // chunk i * NT + tid: plane i / 2, row (i & 1) * 64 + tid / 4
#pragma unroll
      for (int i = 0; i < A_ROWS; i++) ap_ptr[i] += kBK;
    } else {
#pragma unroll
      for (int i = 0; i < A_LOADS; i++) { ra[i] = *reinterpret_cast<const f32x4 *>(dummy ? p.W : a_ptr[i]); a_ptr[i] += kBK; }
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; i++) {
      const int c = i * NT + tid;
      if (B_CH % NT == 0 || c < B_CH) {
        const int pl = c / (BN * 4), r = (c % (BN * 4)) >> 2, kc = c & 3;
        rb[i] = *reinterpret_cast<const uint4 *>(dummy ? Wp : Wp + pl * plane_stride + (long long)(n0 + r) * p.ldw + kt * kBK + kc * 8);
      }
    }
  };
  auto store_tiles = [&]() {
    if (PA) {
#pragma unroll
      for (int i = 0; i < AP_LOADS; i++) *reinterpret_cast<uint4 *>(As + ((i >> 1) * kBM + (i & 1) * 64 + (tid >> 2)) * kRow + (tid & 3) * 16) = rap[i];
    } else
#pragma unroll
    for (int i = 0; i < A_LOADS; i++) {      // split the four floats: hi = top 16 bits, mid = top 16 bits of (x - hi), lo = x - hi - mid (8 significant bits left: exact as bf16)
      unsigned h[4], m[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; e++) split3(ra[i][e], h[e], m[e], l[e]);
      char *q = As + (i * LR + ld_row) * kRow + ld_kv * 2;
      // (v_perm_b32 selector 0x07060302: the upper halves of the two operands, the second operand's in the low half)
      *reinterpret_cast<uint2 *>(q) = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
      *reinterpret_cast<uint2 *>(q + kBM * kRow) = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
      *reinterpret_cast<uint2 *>(q + 2 * kBM * kRow) = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; i++) {
      const int c = i * NT + tid;
      if (B_CH % NT == 0 || c < B_CH) { const int pl = c / (BN * 4), r = (c % (BN * 4)) >> 2, kc = c & 3; *reinterpret_cast<uint4 *>(Bs + (pl * BN + r) * kRow + kc * 16) = rb[i]; }
    }
  };
  f32x16 acc[MI][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ni++) {
    const int col = n0 + wn * WN + ni * 32 + (lane & 31);
    const float *bias = p.seq_bias ? p.seq_bias + (long long)td.bias_row * p.ld_seq_bias : p.bias;
    const float b0 = (bias && col < p.N) ? bias[col] : 0.0f;
    const float b1 = (p.seq_bias && td.split < td.nrows && col < p.N) ? p.seq_bias[(long long)td.bias_row2 * p.ld_seq_bias + col] : b0;
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[mi][ni][r] = (wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) < td.split ? b0 : b1;
  }
  const int nk = (p.Ktot + kBK - 1) / kBK;
  int oi_next = 0, w_next = 0;
  load_tiles(0, 0, 0);
  const int frag_row = lane & 31, frag_b = (lane >> 5) * 16;      // a lane's row of the 32 x 32 block and its 16-byte half of a 16-wide k step
  for (int kt = 0; kt < nk; kt++) {
    __syncthreads();      // (every wavefront is through with the previous tile's fragments)
    store_tiles();
    __syncthreads();
    const int oi_cur = oi_next; const bool more = kt + 1 < nk;
    if (++w_next == p.tiles_per_off) { w_next = 0; oi_next++; }
    load_tiles(more ? kt + 1 : kt, more ? oi_next : oi_cur, more ? w_next : 1, !more);      // in flight under this tile's MFMAs
    const char *a = As + (wm * WM + frag_row) * kRow + frag_b, *b = Bs + (wn * WN + frag_row) * kRow + frag_b;
#pragma unroll
    for (int s_ = 0; s_ < kBK / 16; s_++) {
      bf16x8 fa[3][MI], fb[3][NI];
#pragma unroll
      for (int pl = 0; pl < 3; pl++) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++) fa[pl][mi] = *reinterpret_cast<const bf16x8 *>(a + (pl * kBM + mi * 32) * kRow + s_ * 32);
#pragma unroll
        for (int ni = 0; ni < NI; ni++) fb[pl][ni] = *reinterpret_cast<const bf16x8 *>(b + (pl * BN + ni * 32) * kRow + s_ * 32);
      }
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {      // the small terms first
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][mi], fb[1][ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][mi], fb[0][ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][mi], fb[2][ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][mi], fb[0][ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][mi], fb[1][ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][mi], fb[0][ni], acc[mi][ni], 0, 0, 0);
        }
    }
  }
  __syncthreads();
  // ---- epilogue.  The MFMA C/D layout gives a lane 4 consecutive ROWS of one column (col = lane & 31,
  // row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)), i.e. dword stores of 128-byte row pieces: 64 store and 64 residual-load
  // instructions per lane, which is what the affine layers (K = 192 only) spent 60 % of their time on.  Instead each wavefront
  // transposes its WM x WN accumulator tile through LDS (the tile buffers are dead by now) and then works on float4 row
  // pieces: 4x fewer, 4x wider memory instructions, 256-byte runs per row.
  constexpr int kStLd = WN + 4, C4 = WN / 4, RPI = 64 / C4, ITERS = WM / RPI;      // RPI rows per iteration; lanes >= RPI * C4 idle (WN = 96: 48 of 64 busy)
  float *stage = reinterpret_cast<float *>(smem) + wave * (WM * kStLd);
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        stage[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * kStLd + ni * 32 + (lane & 31)] = acc[mi][ni][r];
  int res_kind = -1;
#pragma unroll
  for (int o = 0; o < kMaxOps; o++) if (o < p.nops && p.op_kind[o] == k3::kEpiResidual) res_kind = o;
  const float *__restrict__ R = p.R; float *__restrict__ C = p.C;
  const bool vec_ok = (p.N % 4 == 0) && (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                      (res_kind < 0 || ((p.ldr % 4 == 0) && ((reinterpret_cast<uintptr_t>(R) & 15) == 0)));
  // Everything the per-element program needs is fetched up front, all loads in flight together: the residual row pieces of the
  // tile and, per epilogue op, this lane's four columns of its scale / offset vectors (a lane keeps the same columns over all
  // its rows).  The op program is then applied op by op over the whole register tile: six block-uniform dispatches per tile
  // instead of per row piece, and no scalar / vector loads between the arithmetic.  (Fetching kinds, pointers and vectors
  // inside the row loop cost 42 k of the 48 k epilogue cycles of a K = 192 layer.)
  const int c4 = lane % C4, row0 = lane / C4;                 // row = it * RPI + row0, column group c4
  static_assert(WM % RPI == 0, "rows per iteration must divide the wave tile");
  const bool lane_on = lane < RPI * C4;
  const int col = lane_on ? n0 + wn * WN + c4 * 4 : p.N;      // idle lanes look like out-of-range columns
  if (vec_ok) {
    f32x4 res[ITERS], opS[kMaxOps], opO[kMaxOps], v[ITERS];
    const bool col_ok = col < p.N;
    const int colc = min(col, p.N - 4);
    const int row0c = lane_on ? row0 : 0;
    // a full tile (all of the 128 rows and BN columns exist: every tile but the last of an utterance / of N) needs no per-row
    // bounds logic and walks its rows with one pointer increment per row piece
    const bool full = td.nrows == kBM && n0 + BN <= p.N;
    if ((EPI == kEpiAny || EPI == kEpiAnyMap || EPI == kEpiReluScaleRes) && res_kind >= 0 && !(0 & 1)) {
      if (full && td.split >= td.nrows) {
        const float *rp = R + (long long)(td.res_base + (wm * WM + row0c) * p.res_row_stride) * p.ldr + colc;
        const long long rstep = (long long)RPI * p.res_row_stride * p.ldr;
#pragma unroll
        for (int it = 0; it < ITERS; it++) res[it] = *reinterpret_cast<const f32x4 *>(rp + it * rstep);
      } else {
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
          const int lrow = min(wm * WM + it * RPI + row0, td.nrows - 1);
          res[it] = *reinterpret_cast<const f32x4 *>(R + (long long)((lrow < td.split ? td.res_base : td.res_base2) + lrow * p.res_row_stride) * p.ldr + colc);
        }
      }
    }
    if (EPI == kEpiAny || EPI == kEpiAnyMap) {
#pragma unroll
      for (int o = 0; o < kMaxOps; o++) {
        if (o < p.nops && p.op_kind[o] == k3::kEpiScaleOffset) {
          opS[o] = *reinterpret_cast<const f32x4 *>(p.op_scale[o] + colc); opO[o] = *reinterpret_cast<const f32x4 *>(p.op_offset[o] + colc);
        }
      }
    } else if (EPI == kEpiReluScaleRes || EPI == kEpiReluScale) {
      opS[1] = *reinterpret_cast<const f32x4 *>(p.op_scale[1] + colc); opO[1] = *reinterpret_cast<const f32x4 *>(p.op_offset[1] + colc);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; it++) v[it] = *reinterpret_cast<const f32x4 *>(stage + (it * RPI + row0c) * kStLd + c4 * 4);
    if (EPI == kEpiAny || EPI == kEpiAnyMap) {
#pragma unroll
      for (int o = 0; o < kMaxOps; o++) {
        if (o < p.nops) {
          const int kind = p.op_kind[o];
          if (kind == k3::kEpiRelu) {
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
              v[it][0] = fmaxf(v[it][0], 0.0f);
              v[it][1] = fmaxf(v[it][1], 0.0f);
              v[it][2] = fmaxf(v[it][2], 0.0f);
              v[it][3] = fmaxf(v[it][3], 0.0f);
            }
          } else if (kind == k3::kEpiScaleOffset) {
#pragma unroll
            for (int it = 0; it < ITERS; it++) v[it] = v[it] * opS[o] + opO[o];
          } else if (EPI == kEpiAnyMap && (kind == k3::kEpiSigmoid || kind == k3::kEpiTanh)) {
#pragma unroll
            for (int it = 0; it < ITERS; it++)
#pragma unroll
              for (int e = 0; e < 4; e++) v[it][e] = kind == k3::kEpiSigmoid ? epi_sigmoid(v[it][e]) : epi_tanh(v[it][e]);
          } else {
#pragma unroll
            for (int it = 0; it < ITERS; it++) v[it] = p.res_scale * res[it] + v[it];
          }
        }
      }
    } else if (EPI == kEpiReluScaleRes || EPI == kEpiReluScale) {
#pragma unroll
      for (int it = 0; it < ITERS; it++) {
        v[it][0] = fmaxf(v[it][0], 0.0f); v[it][1] = fmaxf(v[it][1], 0.0f); v[it][2] = fmaxf(v[it][2], 0.0f); v[it][3] = fmaxf(v[it][3], 0.0f);
        v[it] = v[it] * opS[1] + opO[1];
        if (EPI == kEpiReluScaleRes) v[it] = p.res_scale * res[it] + v[it];
      }
    }
    if (full && !(0 & 2)) {
      if (lane_on) {
        float *cp = C + (long long)(td.out_row0 + wm * WM + row0) * p.ldc + col;
        const long long cstep = (long long)RPI * p.ldc;
#pragma unroll
        for (int it = 0; it < ITERS; it++) *reinterpret_cast<f32x4 *>(cp + it * cstep) = v[it];
        if (p.Cp) {      // the consumer's operand planes, written where the values are produced (block-uniform branch)
          const long long at0 = (long long)(td.out_row0 + wm * WM + row0) * p.ldc + col;
#pragma unroll
          for (int it = 0; it < ITERS; it++) store_planes4(p.Cp, p.cp_stride, at0 + it * cstep, v[it]);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < ITERS; it++) {
        const int lrow = wm * WM + it * RPI + row0;
        if (col_ok && lrow < td.nrows && (!(0 & 2) || v[it][0] == 12345.678f)) {
          *reinterpret_cast<f32x4 *>(C + (long long)(td.out_row0 + lrow) * p.ldc + col) = v[it];
          if (p.Cp) store_planes4(p.Cp, p.cp_stride, (long long)(td.out_row0 + lrow) * p.ldc + col, v[it]);
        }
      }
    }
  } else {                                             // unaligned / odd-width output: element-wise tail path
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
      const int row = it * RPI + row0, lrow = wm * WM + row;
      if (col >= p.N || lrow >= td.nrows) continue;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + row * kStLd + c4 * 4);
      for (int e = 0; e < 4; e++) {
        const int c = col + e;
        if (c >= p.N) break;
        float x = v[e];
        for (int o = 0; o < p.nops; o++) {
          const int kind = p.op_kind[o];
          if (kind == k3::kEpiRelu) x = fmaxf(x, 0.0f);
          else if (kind == k3::kEpiScaleOffset) x = x * p.op_scale[o][c] + p.op_offset[o][c];
          else if (EPI == kEpiAnyMap && kind == k3::kEpiSigmoid) x = epi_sigmoid(x);
          else if (EPI == kEpiAnyMap && kind == k3::kEpiTanh) x = epi_tanh(x);
          else x = p.res_scale * R[(long long)((lrow < td.split ? td.res_base : td.res_base2) + lrow * p.res_row_stride) * p.ldr + c] + x;
        }
        C[(long long)(td.out_row0 + lrow) * p.ldc + c] = x;
        if (p.Cp) { unsigned h_, m_, l_; split3(x, h_, m_, l_); const long long at = (long long)(td.out_row0 + lrow) * p.ldc + c; p.Cp[at] = (unsigned short)(h_ >> 16); p.Cp[p.cp_stride + at] = (unsigned short)(m_ >> 16); p.Cp[2 * p.cp_stride + at] = (unsigned short)(l_ >> 16); }
      }
    }
  }
}

// element-wise fused node without a GEMM (rare: a ReLU/BatchNorm/NoOp whose input has several consumers)
__global__ __launch_bounds__(256) void k3_elementwise_kernel(GemmParams p) {
  const TileDesc td = p.tiles[blockIdx.x];
  for (int idx = threadIdx.x; idx < td.nrows * p.N; idx += 256) {
    const int lrow = idx / p.N, col = idx - lrow * p.N;
    const bool s2 = lrow >= td.split;
    const int row = clampi((s2 ? td.in_base2 : td.in_base) + lrow * p.row_stride + p.shifts[0], s2 ? td.in_lo2 : td.in_lo, s2 ? td.in_hi2 : td.in_hi);
    float v = p.A[(long long)row * p.lda + col];
    for (int o = 0; o < p.nops; o++) {
      const int kind = p.op_kind[o];
      if (kind == k3::kEpiRelu) v = fmaxf(v, 0.0f);
      else if (kind == k3::kEpiScaleOffset) v = v * p.op_scale[o][col] + p.op_offset[o][col];
      else if (kind == k3::kEpiSigmoid) v = epi_sigmoid(v);
      else if (kind == k3::kEpiTanh) v = epi_tanh(v);
      else v = p.res_scale * p.R[(long long)((s2 ? td.res_base2 : td.res_base) + lrow * p.res_row_stride) * p.ldr + col] + v;
    }
    p.C[(long long)(td.out_row0 + lrow) * p.ldc + col] = v;
  }
}

int gcd_i(int a, int b) { a = abs(a); b = abs(b); while (b) { int t = a % b; a = b; b = t; } return a; }
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct DeviceNode {     // model-level (batch independent) device data of one fused node
  float *W = nullptr; int ldw = 0, npad = 0, bn = 128;
  unsigned short *Wp = nullptr; long long plane_stride = 0;      // the weights split into three bf16 planes [3][npad][ldw] (k3_nnet_batch_set_precision(.., 1) makes them)
  float *bias = nullptr;
  float *W_iv = nullptr;                      // [N x ivector_dim] (nodes fed by ReplaceIndex(ivector, t, 0))
  std::vector<float *> op_scale, op_offset;   // per op (null when not scale/offset)
};

// seq_bias[s][n] = bias[n] + sum_k W_iv[n][k] * ivector[iv_row[s]][k]: what the i-vector columns of the first affine add to every row of
// sequence (chunk) s.  One thread per (s, n), ascending k like the reference's sgemm; tiny (sequences x N x ivector_dim).
__global__ void k3_seq_bias_kernel(const float *__restrict__ iv, long long ld_iv, const int *__restrict__ iv_row, int num_seqs, const float *__restrict__ W_iv, int iv_dim,
                                   const float *__restrict__ bias, int N, float *__restrict__ out, long long ld_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (long long)num_seqs * N) return;
  const int s = (int)(i / N), n = (int)(i % N);
  const float *x = iv + (long long)iv_row[s] * ld_iv, *w = W_iv + (long long)n * iv_dim; float a = 0.f;
  for (int k = 0; k < iv_dim; k++) a = fmaf(w[k], x[k], a);
  out[(long long)s * ld_out + n] = (bias ? bias[n] : 0.f) + a;
}

}  // namespace

struct k3_nnet {
  k3::FusedModel fm;
  std::vector<DeviceNode> dev;
  std::vector<void *> allocs;
  bool uploaded = false;
  ~k3_nnet() { for (void *p : allocs) (void)hipFree(p); }
};

namespace {
// LogSoftmaxComponent / SoftmaxComponent::Propagate (nnet-simple-component.cc:3618-3625, :3494-3504; CuMatrixBase::LogSoftMaxPerRow / SoftMaxPerRow): one
// wavefront per row, in place:
// max, sum of exp(x - max) (float, like the CPU's VectorBase::ApplyLogSoftMax / ApplySoftMax), then x - max - log(sum) or exp(x - max) / sum; optional y *
// scale[c] + offset[c] behind it
// (the decodable's prior subtraction and acoustic scale on the output node).
__global__ __launch_bounds__(256) void k3_row_softmax_kernel(float *C, long long ldc, int rows, int cols, int op, const float *scale, const float *offset) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  float *x = C + (long long)r * ldc;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, x[c]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.0f;
  for (int c = lane; c < cols; c += 64) sum += expf(x[c] - mx);
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float lsum = logf(sum), inv = 1.0f / sum;
  for (int c = lane; c < cols; c += 64) {
    float y = op == 1 ? x[c] - mx - lsum : expf(x[c] - mx) * inv;
    if (scale) y = y * scale[c] + offset[c];
    x[c] = y;
  }
}
// NormalizeComponent::Propagate (nnet-normalize-component.cc; cu::NormalizePerRow, cudamatrix/cu-math.cc:280-318) in place: x * (max(|x|^2 / (D target_rms^2), 2^-66))^-1/2
__global__ __launch_bounds__(256) void k3_row_normalize_kernel(float *C, long long ldc, int rows, int cols, float target_rms, const float *scale, const float *offset) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  float *x = C + (long long)r * ldc;
  float ss = 0.0f;
  for (int c = lane; c < cols; c += 64) ss += x[c] * x[c];
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float f = 1.0f / sqrtf(fmaxf(ss * (1.0f / ((float)cols * target_rms * target_rms)), 1.3552527156068805425e-20f));
  for (int c = lane; c < cols; c += 64) { float y = x[c] * f; if (scale) y = y * scale[c] + offset[c]; x[c] = y; }
}

}  // namespace

struct k3_nnet_batch {
  k3_nnet *net = nullptr;
  int num_utts = 0, subsampling = 1, precision = 0;      // precision: 0 = FP32 matrix core (the parity path), 1 = split-bf16, operands split by the loader, 2 = split-bf16, the activations' planes written by the producing epilogue (both exploratory)
  // mode 2: per node the three bf16 planes of its output (null: none; same rows / ld as the fp32 buffer, which shares its slot with other nodes -- so do the planes) and whether a
  // consumer wants them
  std::vector<int> act_slot; std::vector<size_t> slot_bytes; std::vector<unsigned short *> slot_planes; std::vector<char> wants_planes;
  std::vector<int> num_frames;
  std::vector<long long> out_offsets;           // [U+1] rows of the output matrix
  long long total_in_rows = 0, total_out_rows = 0;
  std::vector<long long> node_rows;      // output rows of every fused node (the row-wise softmax kernels run over them)
  double flops = 0.0;
  // per node
  std::vector<GemmParams> params;
  std::vector<int> node_a, node_r, node_g;      // first time, right extension, step
  std::vector<void *> allocs;
  float *out_scale = nullptr, *out_offset = nullptr;
  // i-vector input: sequences (chunks) and the i-vector row each one takes
  int num_seqs = 0; int *d_seq_iv_row = nullptr; long long total_iv_rows = 0;
  std::vector<float *> seq_bias;                // per node (null = none) [num_seqs x ld]
  std::vector<int> seq_bias_ld;
  // K3_NNET_SPLIT=1 (opt-in, see k3_nnet_batch_create): the batch is planned as TWO halves of its utterances (each a batch of its own: tiles, activation buffers), run on two
  // streams: a launch's last, partly filled round of tiles and the ramps at both ends of the forward's ~36 launches then run under the other half's tiles (512 x 10 s: 27.0 ->
  // 26.3 ms per forward back to back; the same tiles and the same arithmetic, so the output is bit-identical).  This object then only holds the totals.
  std::unique_ptr<k3_nnet_batch> half[2];
  long long half_in_rows0 = 0, half_out_rows0 = 0;      // rows of the first half in the feature / output matrices
  hipStream_t side = nullptr; int side_prio = 0; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  ~k3_nnet_batch() {
    for (void *p : allocs) (void)hipFree(p);
    if (side) (void)hipStreamDestroy(side);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
  }
};

namespace {

template <typename T>
int upload(std::vector<void *> *allocs, const std::vector<T> &h, T **d) {
  *d = nullptr;
  if (h.empty()) return K3_OK;
  K3_HIP_CHECK(hipMalloc((void **)d, h.size() * sizeof(T)));
  allocs->push_back(*d);
  K3_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return K3_OK;
}

void free_all(std::vector<void *> *allocs) {
  for (void *p : *allocs) (void)hipFree(p);
  allocs->clear();
}

}  // namespace

// ------------------------------------------------------------------------------ C ABI ----
// weights are uploaded lazily (first batch) so that k3_nnet_load / k3_nnet_get_info work without a GPU
static int ensure_uploaded(k3_nnet *net) {
  k3::FusedModel &fm = net->fm;
  if (net->uploaded) return K3_OK;
  net->dev.resize(fm.nodes.size());
  for (size_t i = 0; i < fm.nodes.size(); i++) {
    const k3::FusedNode &f = fm.nodes[i];
    DeviceNode &d = net->dev[i];
    if (f.has_gemm) {
      const int K = (int)f.offsets.size() * f.in_dim, N = f.out_dim;
      d.bn = (N % 96 == 0 && N % 128 != 0) ? 96 : 128;
      d.npad = (int)align_up(N, d.bn); d.ldw = (int)align_up(K, kBK);
      std::vector<float> wp((size_t)d.npad * d.ldw, 0.0f);
      for (int n = 0; n < N; n++) memcpy(&wp[(size_t)n * d.ldw], &f.W[(size_t)n * K], sizeof(float) * K);
      int rc = upload(&net->allocs, wp, &d.W); if (rc) { return rc; }
      rc = upload(&net->allocs, f.bias, &d.bias); if (rc) { return rc; }
      rc = upload(&net->allocs, f.W_iv, &d.W_iv); if (rc) { return rc; }
    }
    d.op_scale.assign(f.ops.size(), nullptr); d.op_offset.assign(f.ops.size(), nullptr);
    for (size_t o = 0; o < f.ops.size(); o++)
      if (f.ops[o].kind == k3::kEpiScaleOffset) {
        int rc = upload(&net->allocs, f.ops[o].scale, &d.op_scale[o]); if (rc) { return rc; }
        rc = upload(&net->allocs, f.ops[o].offset, &d.op_offset[o]); if (rc) { return rc; }
      }
  }
  net->uploaded = true;
  return K3_OK;
}

extern "C" int k3_nnet_load(const char *path, k3_nnet **out) {
  K3_REQUIRE(path && out, "k3_nnet_load: null argument");
  k3::RawModel raw; std::string err;
  if (!k3::ReadModelFile(path, &raw, &err)) { k3::set_error("k3_nnet_load: %s", err.c_str()); return K3_ERR_ARG; }
  std::unique_ptr<k3_nnet> net(new k3_nnet());
  if (!k3::FuseModel(raw, &net->fm, &err)) { k3::set_error("k3_nnet_load: %s: %s", path, err.c_str()); return K3_ERR_UNSUPPORTED; }
  for (const k3::FusedNode &f : net->fm.nodes) {
    if ((int)f.offsets.size() > kMaxOffsets || (int)f.ops.size() > kMaxOps) {
      k3::set_error("k3_nnet_load: node %s has %zu offsets / %zu epilogue ops (limits %d / %d)", f.name.c_str(), f.offsets.size(), f.ops.size(), kMaxOffsets, kMaxOps);
      return K3_ERR_UNSUPPORTED;
    }
    if (f.in_dim % 4 != 0 && f.has_gemm) {
      k3::set_error("k3_nnet_load: node %s input dim %d is not a multiple of 4 (vector loads)", f.name.c_str(), f.in_dim);
      return K3_ERR_UNSUPPORTED;
    }
  }
  *out = net.release();
  return K3_OK;
}

extern "C" void k3_nnet_destroy(k3_nnet *net) { delete net; }

extern "C" int k3_nnet_get_info(const k3_nnet *net, k3_nnet_info *info) {
  K3_REQUIRE(net && info, "k3_nnet_get_info: null argument");
  info->input_dim = net->fm.input_dim; info->output_dim = net->fm.output_dim;
  info->left_context = net->fm.left_context; info->right_context = net->fm.right_context;
  info->num_components = net->fm.num_components; info->num_fused_nodes = (int)net->fm.nodes.size();
  info->has_priors = net->fm.priors.empty() ? 0 : 1; info->num_params = net->fm.num_params; info->ivector_dim = net->fm.ivector_dim;
  return K3_OK;
}

extern "C" int k3_nnet_get_priors(const k3_nnet *net, float *h_priors) {
  K3_REQUIRE(net && h_priors, "k3_nnet_get_priors: null argument");
  K3_REQUIRE(!net->fm.priors.empty(), "k3_nnet_get_priors: model has no priors");
  memcpy(h_priors, net->fm.priors.data(), sizeof(float) * net->fm.priors.size());
  return K3_OK;
}

extern "C" void k3_nnet_batch_destroy(k3_nnet_batch *b) { delete b; }

// One planner for both entry points.  A SEQUENCE is what the network is evaluated over as one piece: a whole utterance (no i-vector, or one
// i-vector per utterance: the result does not depend on how the reference chunks it) or one chunk of frames_per_chunk frames of an utterance
// (--online-ivectors: every chunk gets its own i-vector, so the activations near a chunk boundary differ between the two chunks that need them,
// exactly as in DecodableNnetSimple::EnsureFrameIsComputed, nnet-am-decodable-simple.cc:93-168).  Input rows outside the UTTERANCE are clamped
// (edge-frame replication, :154-163) whatever the sequence.
namespace { struct Seq { int u, t0, n_out, iv_row; }; }
static int batch_create_impl(k3_nnet *net, int32_t num_utts, const int32_t *h_num_frames, int32_t subsampling, const float *h_log_priors, float acoustic_scale,
                             bool with_ivector, int32_t frames_per_chunk, int32_t online_ivector_period, const int32_t *h_num_ivector_rows, k3_nnet_batch **out) {
  K3_REQUIRE(net && h_num_frames && out && num_utts > 0 && subsampling >= 1, "k3_nnet_batch_create: bad argument");
  { const int rc = ensure_uploaded(net); if (rc) return rc; }
  const k3::FusedModel &fm = net->fm;
  const int nn = (int)fm.nodes.size();
  std::unique_ptr<k3_nnet_batch> b(new k3_nnet_batch());
  b->net = net; b->num_utts = num_utts; b->subsampling = subsampling;
  b->num_frames.assign(h_num_frames, h_num_frames + num_utts);
  for (int u = 0; u < num_utts; u++) K3_REQUIRE(h_num_frames[u] > 0, "k3_nnet_batch_create: utterance with no frames");
  std::vector<Seq> seqs;
  {
    long long iv_base = 0;
    for (int u = 0; u < num_utts; u++) {
      const int n_sub = (h_num_frames[u] + subsampling - 1) / subsampling;
      if (!with_ivector || online_ivector_period <= 0) { seqs.push_back({u, 0, n_sub, with_ivector ? (int)iv_base : -1}); if (with_ivector) iv_base += 1; continue; }
      // the chunk rounded up to a multiple of s (CheckAndFixConfigs, nnet-am-decodable-simple.h:120-134)
      const int per = (frames_per_chunk + subsampling - 1) / subsampling, rows = h_num_ivector_rows[u];
      K3_REQUIRE(rows > 0, "k3_nnet_batch_create_ivector: utterance without i-vector rows");
      for (int c0 = 0; c0 < n_sub; c0 += per) {
        const int n = std::min(per, n_sub - c0), first = c0 * subsampling, last = (c0 + n - 1) * subsampling;
        int f = (first + (last - first) / 2) / online_ivector_period;                                        // GetCurrentIvector :178-213
        if (f >= rows) {
          if ((long long)(f - (rows - 1)) * online_ivector_period > 50) {
            k3::set_error("Could not get iVector for frame %d, only available till frame %d * ivector-period=%d (mismatched --online-ivector-period?)",
                first + (last - first) / 2, rows, online_ivector_period);
            return K3_ERR_ARG;
          }
          f = rows - 1;
        }
        seqs.push_back({u, first, n, (int)(iv_base + f)});
      }
      iv_base += rows;
    }
    b->total_iv_rows = iv_base;
  }
  const int num_seqs = (int)seqs.size(); b->num_seqs = num_seqs;

  // ---- which time steps must each node produce?  t = a + k*g, k >= 0, up to t_last_out(u) + r
  std::vector<int> A(nn, 0), R(nn, 0), G(nn, 0); std::vector<char> used(nn, 0);
  A[fm.output_node] = 0; R[fm.output_node] = 0; G[fm.output_node] = subsampling; used[fm.output_node] = 1;
  {
    // consumers are always later nodes, so one backward sweep suffices; a node may have several consumers
    std::vector<std::vector<std::pair<int, int>>> anchors(nn);   // (anchor time, step) contributed by consumers
    std::vector<int> rmax(nn, -(1 << 30));
    for (int i = nn - 1; i >= 0; i--) {
      if (i != fm.output_node) {
        if (anchors[i].empty()) continue;          // unused node (e.g. feeds only an unused output)
        int a = 1 << 30, g = 0;
        for (auto &an : anchors[i]) a = std::min(a, an.first);
        for (auto &an : anchors[i]) { g = gcd_i(g, an.second); g = gcd_i(g, an.first - a); }
        A[i] = a; G[i] = g; R[i] = rmax[i]; used[i] = 1;
      }
      const k3::FusedNode &f = fm.nodes[i];
      auto contribute = [&](int src, int off) {
        if (src < 0) return;
        anchors[src].push_back({A[i] + off, G[i]});
        rmax[src] = std::max(rmax[src], R[i] + off);
      };
      for (int o : f.offsets) contribute(f.input, o);
      for (const k3::EpiOp &op : f.ops) if (op.kind == k3::kEpiResidual) contribute(op.res_node, 0);
    }
  }
  b->node_a = A; b->node_r = R; b->node_g = G;

  // ---- row bookkeeping
  auto rows_of = [&](int i, int q) { const int tlast = (seqs[q].n_out - 1) * subsampling; return (tlast + R[i] - A[i]) / G[i] + 1; };      // rows of node i in sequence q
  std::vector<std::vector<long long>> rowoff(nn, std::vector<long long>(num_seqs + 1, 0));
  for (int i = 0; i < nn; i++) if (used[i]) for (int q = 0; q < num_seqs; q++) rowoff[i][q + 1] = rowoff[i][q] + rows_of(i, q);
  std::vector<long long> featoff(num_utts + 1, 0);
  for (int u = 0; u < num_utts; u++) featoff[u + 1] = featoff[u] + b->num_frames[u];
  b->total_in_rows = featoff[num_utts];
  b->out_offsets.assign(num_utts + 1, 0);                      // the chunks of an utterance are consecutive sequences, so its output rows are contiguous
  for (int q = 0; q < num_seqs; q++) b->out_offsets[seqs[q].u + 1] += seqs[q].n_out;
  for (int u = 0; u < num_utts; u++) b->out_offsets[u + 1] += b->out_offsets[u];
  b->total_out_rows = b->out_offsets[num_utts];
  b->node_rows.assign(nn, 0); for (int i = 0; i < nn; i++) if (used[i]) b->node_rows[i] = rowoff[i][num_seqs];
  for (int i = 0; i < nn; i++) K3_REQUIRE(rowoff[i][num_seqs] < (1ll << 31), "k3_nnet_batch_create: more than 2^31 rows in one batch");

  // ---- activation buffers with liveness-based reuse
  std::vector<int> last_use(nn, -1);
  for (int i = 0; i < nn; i++) {
    if (!used[i]) continue;
    if (fm.nodes[i].input >= 0) last_use[fm.nodes[i].input] = i;
    for (const k3::EpiOp &op : fm.nodes[i].ops) if (op.kind == k3::kEpiResidual && op.res_node >= 0) last_use[op.res_node] = i;
  }
  struct Slot { size_t bytes; int busy_until; float *ptr; };
  std::vector<Slot> slots; std::vector<int> slot_of(nn, -1); std::vector<int> ld(nn, 0);
  for (int i = 0; i < nn; i++) {
    if (!used[i] || i == fm.output_node) continue;
    ld[i] = (int)align_up(fm.nodes[i].out_dim, 4);
    const size_t need = (size_t)rowoff[i][num_seqs] * ld[i] * sizeof(float);
    int best = -1;
    for (size_t s = 0; s < slots.size(); s++) if (slots[s].busy_until < i && (best < 0 || slots[s].bytes > slots[best].bytes)) best = (int)s;
    if (best < 0) { slots.push_back({need, last_use[i], nullptr}); best = (int)slots.size() - 1; }
    else { slots[best].bytes = std::max(slots[best].bytes, need); slots[best].busy_until = last_use[i]; }
    slot_of[i] = best;
  }
  for (Slot &s : slots) { K3_HIP_CHECK(hipMalloc((void **)&s.ptr, std::max<size_t>(s.bytes, 256))); b->allocs.push_back(s.ptr); }
  b->act_slot = slot_of; b->slot_bytes.clear(); for (const Slot &s : slots) b->slot_bytes.push_back(std::max<size_t>(s.bytes, 256));
  b->slot_planes.assign(slots.size(), nullptr); b->wants_planes.assign(nn, 0);
  for (int i = 0; i < nn; i++) {      // a node's planes are wanted when a consumer can run on the bf16 matrix core from planes (tile-aligned offsets, 16-byte aligned plane rows)
    const int src = used[i] ? fm.nodes[i].input : -1;
    if (src >= 0 && fm.nodes[i].has_gemm && fm.nodes[i].in_dim % kBK == 0 && ld[src] % 8 == 0 && slot_of[src] >= 0) b->wants_planes[src] = 1;
  }

  // ---- output transform: (x - log_prior) * acwt as one more scale/offset op (nnet-am-decodable-simple.cc:268-271)
  const bool out_xform = (h_log_priors != nullptr) || acoustic_scale != 1.0f;
  if (out_xform) {
    std::vector<float> sc(fm.output_dim, acoustic_scale), of(fm.output_dim, 0.0f);
    if (h_log_priors) for (int i = 0; i < fm.output_dim; i++) of[i] = -h_log_priors[i] * acoustic_scale;
    int rc = upload(&b->allocs, sc, &b->out_scale); if (rc) { return rc; }
    rc = upload(&b->allocs, of, &b->out_offset); if (rc) { return rc; }
  }

  // ---- per-node launch parameters + tile tables
  b->params.resize(nn); b->seq_bias.assign(nn, nullptr); b->seq_bias_ld.assign(nn, 0);
  for (int i = 0; i < nn; i++) {
    GemmParams &p = b->params[i]; memset(&p, 0, sizeof(p));
    if (!used[i]) { p.num_m_tiles = 0; continue; }
    const k3::FusedNode &f = fm.nodes[i]; const DeviceNode &d = net->dev[i];
    const int src = f.input;
    const int g_in = src < 0 ? 1 : G[src], a_in = src < 0 ? 0 : A[src];
    p.in_dim = f.in_dim; p.noff = (int)f.offsets.size(); p.row_stride = G[i] / g_in;
    for (int o = 0; o < p.noff; o++) {
      const int num = A[i] + f.offsets[o] - a_in;
      if (num % g_in != 0 || G[i] % g_in != 0) { k3::set_error("k3_nnet_batch_create: internal time-grid error at node %s", f.name.c_str()); return K3_ERR_ARG; }
      p.shifts[o] = num / g_in;
    }
    p.tiles_per_off = (f.in_dim % kBK == 0) ? f.in_dim / kBK : 0; p.tiles_per_seg = 384 / kBK;
    p.A = src < 0 ? nullptr : slots[slot_of[src]].ptr;       // network input pointer is patched in k3_nnet_forward
    p.lda = src < 0 ? 0 : ld[src];
    p.W = d.W; p.ldw = d.ldw; p.Ktot = p.noff * f.in_dim; p.N = f.out_dim; p.bias = d.bias;
    p.C = (i == fm.output_node) ? nullptr : slots[slot_of[i]].ptr; p.ldc = ld[i];
    p.nops = (int)f.ops.size();
    int res = -2;
    for (int o = 0; o < p.nops; o++) {
      p.op_kind[o] = f.ops[o].kind; p.op_scale[o] = d.op_scale[o]; p.op_offset[o] = d.op_offset[o];
      if (f.ops[o].kind == k3::kEpiResidual) { res = f.ops[o].res_node; p.res_scale = f.ops[o].res_scale; }
    }
    if (i == fm.output_node && out_xform && f.row_op == 0) {      // (behind a softmax output layer the transform is applied by the row kernel: it follows the component)
      if (p.nops >= kMaxOps) { k3::set_error("k3_nnet_batch_create: too many epilogue ops on the output node"); return K3_ERR_UNSUPPORTED; }
      p.op_kind[p.nops] = k3::kEpiScaleOffset; p.op_scale[p.nops] = b->out_scale; p.op_offset[p.nops] = b->out_offset; p.nops++;
    }
    int g_res = 1, a_res = 0;
    if (res >= -1) {
      g_res = res < 0 ? 1 : G[res]; a_res = res < 0 ? 0 : A[res];
      p.R = res < 0 ? nullptr : slots[slot_of[res]].ptr; p.ldr = res < 0 ? 0 : ld[res];
      p.res_row_stride = G[i] / g_res;
    }
    std::vector<TileDesc> tiles; TileDesc open; bool has_open = false;      // `open`: a slab whose sequence ended before row 128
    auto segment = [&](int q, int r0, int *in_base, int *in_lo, int *in_hi, int *res_base) {      // rows r0.. of sequence q: input / residual row of its local row 0, clamp bounds
      const int u = seqs[q].u, t0 = seqs[q].t0;
      if (src < 0) { *in_base = (int)featoff[u] + t0 + r0 * p.row_stride; *in_lo = (int)featoff[u]; *in_hi = (int)featoff[u + 1] - 1; }
      else { *in_base = (int)rowoff[src][q] + r0 * p.row_stride; *in_lo = (int)rowoff[src][q]; *in_hi = (int)rowoff[src][q + 1] - 1; }
      *res_base = 0;
      if (res >= -1) *res_base = (int)(res < 0 ? featoff[u] + t0 : rowoff[res][q]) + (A[i] - a_res) / g_res + r0 * p.res_row_stride;
    };
    for (int q = 0; q < num_seqs; q++) {
      const int rows = rows_of(i, q); int r0 = 0;
      if (has_open) {      // the previous sequence's last slab takes this sequence's first rows
        const int take = std::min(kBM - open.nrows, rows);
        segment(q, 0, &open.in_base2, &open.in_lo2, &open.in_hi2, &open.res_base2); open.bias_row2 = q;
        open.split = open.nrows; open.nrows += take;
        open.in_base2 -= open.split * p.row_stride; open.res_base2 -= open.split * p.res_row_stride;      // local row r -> base2 + r * stride
        tiles.push_back(open); has_open = false; r0 = take;
      }
      for (; r0 < rows; r0 += kBM) {
        TileDesc t; memset(&t, 0, sizeof t);
        t.out_row0 = (int)rowoff[i][q] + r0; t.nrows = std::min(kBM, rows - r0); t.split = t.nrows; t.bias_row = q;
        segment(q, r0, &t.in_base, &t.in_lo, &t.in_hi, &t.res_base);
        t.in_base2 = t.in_base; t.in_lo2 = t.in_lo; t.in_hi2 = t.in_hi; t.res_base2 = t.res_base; t.bias_row2 = q;
        if (t.nrows < kBM && q + 1 < num_seqs) { open = t; has_open = true; } else tiles.push_back(t);
      }
      if (f.has_gemm) b->flops += 2.0 * rows * (double)p.Ktot * f.out_dim;
    }
    if (has_open) tiles.push_back(open);
    if (!f.W_iv.empty()) {                                       // this node starts from "bias + W_iv . ivector(sequence)" (k3_seq_bias_kernel, run by k3_nnet_forward_ivector)
      K3_REQUIRE(with_ivector, "k3_nnet_batch_create: the model has an i-vector input: use k3_nnet_batch_create_ivector");
      float *sb = nullptr; const int ldsb = (int)align_up(f.out_dim, 4);
      K3_HIP_CHECK(hipMalloc((void **)&sb, (size_t)num_seqs * ldsb * sizeof(float))); b->allocs.push_back(sb);
      b->seq_bias[i] = sb; b->seq_bias_ld[i] = ldsb; p.seq_bias = sb; p.ld_seq_bias = ldsb;
      b->flops += 2.0 * num_seqs * (double)fm.ivector_dim * f.out_dim;
    }
    TileDesc *dt = nullptr; int rc = upload(&b->allocs, tiles, &dt); if (rc) { return rc; }
    p.tiles = dt; p.num_m_tiles = (int)tiles.size();
    p.num_n_tiles = f.has_gemm ? d.npad / d.bn : 1;
  }
  if (with_ivector) {
    std::vector<int> rows(num_seqs); for (int q = 0; q < num_seqs; q++) rows[q] = seqs[q].iv_row;
    int *d = nullptr; const int rc = upload(&b->allocs, rows, &d); if (rc) return rc;
    b->d_seq_iv_row = d;
  }
  *out = b.release();
  return K3_OK;
}

extern "C" int k3_nnet_batch_create(k3_nnet *net, int32_t num_utts, const int32_t *h_num_frames, int32_t subsampling,
                                    const float *h_log_priors, float acoustic_scale, k3_nnet_batch **out) {
  K3_REQUIRE(net, "k3_nnet_batch_create: bad argument");
  K3_REQUIRE(net->fm.ivector_dim == 0, "k3_nnet_batch_create: the model has an i-vector input (input-node name=ivector): use k3_nnet_batch_create_ivector");
  K3_REQUIRE(h_num_frames && out && num_utts > 0, "k3_nnet_batch_create: bad argument");
  // two halves on two streams: OPT-IN (K3_NNET_SPLIT=1).  Measured (tools/bench_nnet_streams.py, bench.py --decoder-stream-priority): forwards back to back 27.0 -> 26.3 ms, but
  // nothing in the pipelined step -- there the gaps of one forward's launches are already filled by the decoder's lanes and the other batch's kernels (80.0 against 80.1 ms with
  // the decoder streams at high queue priority; 89 ms without: two queues of GEMM workgroups then take the freed slots the next decoder launch is waiting for).
  long long frames = 0; for (int u = 0; u < num_utts; u++) frames += h_num_frames[u] > 0 ? h_num_frames[u] : 0;
  const int force = getenv("K3_NNET_SPLIT") ? atoi(getenv("K3_NNET_SPLIT")) : -1;      // (read per call: tests/test_nnet_gpu.py builds both forms in one process)
  const bool split = num_utts >= 2 && force == 1;
  if (!split) return batch_create_impl(net, num_utts, h_num_frames, subsampling, h_log_priors, acoustic_scale, false, 0, 0, nullptr, out);
  std::unique_ptr<k3_nnet_batch> b(new k3_nnet_batch());
  b->net = net; b->num_utts = num_utts; b->subsampling = subsampling; b->num_frames.assign(h_num_frames, h_num_frames + num_utts);
  int u0 = 1; { long long acc = 0; for (int u = 0; u < num_utts; u++) { acc += h_num_frames[u]; if (2 * acc >= frames) { u0 = std::min(std::max(u + 1, 1), num_utts - 1); break; } } }      // first half: about half of the frames
  for (int h = 0; h < 2; h++) {
    k3_nnet_batch *hb = nullptr;
    const int rc = batch_create_impl(net, h == 0 ? u0 : num_utts - u0, h_num_frames + (h == 0 ? 0 : u0), subsampling, h_log_priors, acoustic_scale, false, 0, 0, nullptr, &hb);
    if (rc) return rc;
    b->half[h].reset(hb);
  }
  const k3_nnet_batch &h0 = *b->half[0], &h1 = *b->half[1];
  b->half_in_rows0 = h0.total_in_rows; b->half_out_rows0 = h0.total_out_rows;
  b->total_in_rows = h0.total_in_rows + h1.total_in_rows; b->total_out_rows = h0.total_out_rows + h1.total_out_rows; b->flops = h0.flops + h1.flops;
  b->out_offsets = h0.out_offsets; for (size_t i = 1; i < h1.out_offsets.size(); i++) b->out_offsets.push_back(h0.total_out_rows + h1.out_offsets[i]);
  b->node_rows = h0.node_rows; for (size_t i = 0; i < h1.node_rows.size(); i++) b->node_rows[i] += h1.node_rows[i];
  b->node_a = h0.node_a; b->node_r = h0.node_r; b->node_g = h0.node_g; b->num_seqs = h0.num_seqs + h1.num_seqs;
  K3_HIP_CHECK(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming)); K3_HIP_CHECK(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
  *out = b.release();
  return K3_OK;
}

extern "C" int k3_nnet_batch_create_ivector(k3_nnet *net, int32_t num_utts, const int32_t *h_num_frames, int32_t subsampling, const float *h_log_priors, float acoustic_scale,
                                            int32_t frames_per_chunk, int32_t online_ivector_period, const int32_t *h_num_ivector_rows, k3_nnet_batch **out) {
  K3_REQUIRE(net, "k3_nnet_batch_create_ivector: bad argument");
  // "Neural net expects 'ivector' features with dimension 0 but you provided N" (:105-107)
  K3_REQUIRE(net->fm.ivector_dim > 0, "k3_nnet_batch_create_ivector: the model has no i-vector input");
  K3_REQUIRE(online_ivector_period == 0 || (online_ivector_period > 0 && frames_per_chunk > 0 && h_num_ivector_rows),
      "k3_nnet_batch_create_ivector: online i-vectors need a period, a chunk size and the row counts");
  return batch_create_impl(net, num_utts, h_num_frames, subsampling, h_log_priors, acoustic_scale, true, frames_per_chunk, online_ivector_period, h_num_ivector_rows, out);
}

extern "C" int64_t k3_nnet_batch_ivector_rows(const k3_nnet_batch *b) { return b ? b->total_iv_rows : -1; }

extern "C" int64_t k3_nnet_batch_output_rows(const k3_nnet_batch *b, int64_t *h_out_offsets) {
  if (!b) return -1;
  if (h_out_offsets) for (size_t i = 0; i < b->out_offsets.size(); i++) h_out_offsets[i] = b->out_offsets[i];
  return b->total_out_rows;
}

extern "C" double k3_nnet_batch_flops(const k3_nnet_batch *b) { return b ? b->flops : -1.0; }

// k3_nnet_batch_set_precision(batch, 1): the affine products of this batch run as six bf16 matrix-core products over three-way split operands (k3_tdnn_gemm_x6_kernel) wherever a
// node's time offsets are tile aligned (every layer of a TDNN-F but the first); 0 = back to the FP32 matrix core.  The first call splits the model's weights (host, once).
extern "C" int k3_nnet_batch_set_precision(k3_nnet_batch *b, int32_t mode) {
  K3_REQUIRE(b && (mode == 0 || mode == 1 || mode == 2), "k3_nnet_batch_set_precision: mode is 0 (FP32 matrix core), 1 (split-bf16, split in the loader) or 2 (split-bf16, planes from the producer)");
  if (mode >= 1) {
    k3_nnet *net = b->net; static std::mutex mu; std::lock_guard<std::mutex> g(mu);
    for (size_t i = 0; i < net->fm.nodes.size(); i++) {
      const k3::FusedNode &f = net->fm.nodes[i]; DeviceNode &d = net->dev[i];
      if (!f.has_gemm || d.Wp || f.in_dim % kBK != 0) continue;
      const int K = (int)f.offsets.size() * f.in_dim, N = f.out_dim; const size_t plane = (size_t)d.npad * d.ldw;
      std::vector<unsigned short> wp(3 * plane, 0);
      for (int n = 0; n < N; n++) for (int k = 0; k < K; k++) {
        const float x = f.W[(size_t)n * K + k]; unsigned xb; memcpy(&xb, &x, 4);
        unsigned hb = xb & 0xFFFF0000u; float hf; memcpy(&hf, &hb, 4); const float r1 = x - hf; unsigned r1b; memcpy(&r1b, &r1, 4);
        unsigned mb = r1b & 0xFFFF0000u; float mf; memcpy(&mf, &mb, 4); const float r2 = r1 - mf; unsigned r2b; memcpy(&r2b, &r2, 4);
        const size_t at = (size_t)n * d.ldw + k;
        wp[at] = (unsigned short)(xb >> 16); wp[plane + at] = (unsigned short)(r1b >> 16); wp[2 * plane + at] = (unsigned short)(r2b >> 16);
      }
      const int rc = upload(&net->allocs, wp, &d.Wp); if (rc) return rc;
      d.plane_stride = (long long)plane;
    }
  }
  auto planes = [](k3_nnet_batch *x) -> int {      // mode 2: three planes per activation slot that a consumer reads from planes (6 bytes per element next to the 4 of the fp32 copy)
    for (size_t i = 0; i < x->act_slot.size(); i++) {
      const int sl = x->act_slot[i];
      if (sl < 0 || !x->wants_planes[i] || x->slot_planes[sl]) continue;
      K3_HIP_CHECK(hipMalloc((void **)&x->slot_planes[sl], 3 * (x->slot_bytes[sl] / 4) * sizeof(unsigned short) + 256)); x->allocs.push_back(x->slot_planes[sl]);
    }
    return K3_OK;
  };
  if (mode == 2) {
    if (b->half[0]) { for (int h = 0; h < 2; h++) { const int rc = planes(b->half[h].get()); if (rc) return rc; } }
    else { const int rc = planes(b); if (rc) return rc; }
  }
  b->precision = mode;
  for (int h = 0; h < 2; h++) if (b->half[h]) b->half[h]->precision = mode;
  return K3_OK;
}

static int forward_impl(k3_nnet_batch *b, const float *d_feats, int64_t ld_feats, float *d_out, int64_t ld_out, void *stream);
extern "C" int k3_nnet_forward(k3_nnet_batch *b, const float *d_feats, int64_t ld_feats, float *d_out, int64_t ld_out, void *stream) {
  K3_REQUIRE(b && b->net->fm.ivector_dim == 0, "k3_nnet_forward: null batch, or the model has an i-vector input (use k3_nnet_forward_ivector)");
  if (!b->half[0]) return forward_impl(b, d_feats, ld_feats, d_out, ld_out, stream);
  K3_REQUIRE(d_feats && d_out, "k3_nnet_forward: null argument");
  // the second half on a side stream of the caller's priority, forked from and joined to the caller's stream by events (also what a stream capture needs)
  hipStream_t st = (hipStream_t)stream; int prio = 0;
  if (st) (void)hipStreamGetPriority(st, &prio);
  if (!b->side || b->side_prio != prio) {
    if (b->side) { K3_HIP_CHECK(hipStreamSynchronize(b->side)); K3_HIP_CHECK(hipStreamDestroy(b->side)); b->side = nullptr; }
    K3_HIP_CHECK(hipStreamCreateWithPriority(&b->side, hipStreamNonBlocking, prio)); b->side_prio = prio;
  }
  K3_HIP_CHECK(hipEventRecord(b->ev_fork, st)); K3_HIP_CHECK(hipStreamWaitEvent(b->side, b->ev_fork, 0));
  { const int rc = forward_impl(b->half[0].get(), d_feats, ld_feats, d_out, ld_out, stream); if (rc) return rc; }
  { const int rc = forward_impl(b->half[1].get(), d_feats + b->half_in_rows0 * ld_feats, ld_feats, d_out + b->half_out_rows0 * ld_out, ld_out, b->side); if (rc) return rc; }
  K3_HIP_CHECK(hipEventRecord(b->ev_join, b->side)); K3_HIP_CHECK(hipStreamWaitEvent(st, b->ev_join, 0));
  return K3_OK;
}
extern "C" int k3_nnet_forward_ivector(k3_nnet_batch *b, const float *d_feats, int64_t ld_feats, const float *d_ivectors, int64_t ld_ivectors, float *d_out,
    int64_t ld_out, void *stream) {
  K3_REQUIRE(b && d_ivectors && b->d_seq_iv_row, "k3_nnet_forward_ivector: null argument, or the batch was not made by k3_nnet_batch_create_ivector");
  const k3::FusedModel &fm = b->net->fm;
  K3_REQUIRE(ld_ivectors >= fm.ivector_dim, "k3_nnet_forward_ivector: ld_ivectors < i-vector dim");
  for (size_t i = 0; i < fm.nodes.size(); i++) {
    if (!b->seq_bias[i]) continue;
    const long long n = (long long)b->num_seqs * fm.nodes[i].out_dim;
    hipLaunchKernelGGL(k3_seq_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_ivectors, (long long)ld_ivectors,
        b->d_seq_iv_row, b->num_seqs, b->net->dev[i].W_iv,
                       fm.ivector_dim, b->net->dev[i].bias, fm.nodes[i].out_dim, b->seq_bias[i], (long long)b->seq_bias_ld[i]);
  }
  return forward_impl(b, d_feats, ld_feats, d_out, ld_out, stream);
}
static int forward_impl(k3_nnet_batch *b, const float *d_feats, int64_t ld_feats, float *d_out, int64_t ld_out, void *stream) {
  K3_REQUIRE(b && d_feats && d_out, "k3_nnet_forward: null argument");
  const k3::FusedModel &fm = b->net->fm;
  K3_REQUIRE(ld_feats >= fm.input_dim && ld_feats % 4 == 0 && ((uintptr_t)d_feats & 15) == 0, "k3_nnet_forward: features must be 16-byte aligned with ld % 4 == 0");
  K3_REQUIRE(ld_out >= fm.output_dim, "k3_nnet_forward: ld_out < output dim");
  static std::once_flag attr_once; int attr_rc = K3_OK;      // per-thread streams may call this concurrently (SURVEY 8b "Threading")
  std::call_once(attr_once, [&]() { attr_rc = [&]() -> int {
#define K3_GEMM_VARIANTS(X) X(128, 64, 64, true, kEpiAny) X(128, 64, 64, false, kEpiAny) X(128, 64, 64, true, kEpiReluScaleRes) X(128, 64, 64, false, kEpiReluScale) \
                            X(128, 64, 64, true, kEpiReluScale) X(96, 32, 96, true, kEpiAny) X(96, 32, 96, false, kEpiAny) X(96, 32, 96, true, kEpiNone) \
                            X(128, 64, 64, true, kEpiAnyMap) X(128, 64, 64, false, kEpiAnyMap) X(96, 32, 96, true, kEpiAnyMap) X(96, 32, 96, false, kEpiAnyMap)
#define K3_SET_ATTR(bn, wm, wn, al, ep) K3_HIP_CHECK(hipFuncSetAttribute((const void *)k3_tdnn_gemm_kernel<bn, wm, wn, al, ep>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    K3_GEMM_VARIANTS(K3_SET_ATTR)
#undef K3_SET_ATTR
#define K3_X6_VARIANTS(X) X(128, 64, 64, kEpiAny) X(128, 64, 64, kEpiReluScaleRes) X(128, 64, 64, kEpiReluScale) X(96, 32, 96, kEpiAny) X(96, 32, 96, kEpiNone)
#define K3_SET_ATTR6(bn, wm, wn, ep) K3_HIP_CHECK(hipFuncSetAttribute((const void *)k3_tdnn_gemm_x6_kernel<bn, wm, wn, ep, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                                     K3_HIP_CHECK(hipFuncSetAttribute((const void *)k3_tdnn_gemm_x6_kernel<bn, wm, wn, ep, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    K3_X6_VARIANTS(K3_SET_ATTR6)
#undef K3_SET_ATTR6
    return K3_OK; }(); });
  if (attr_rc) return attr_rc;
  hipStream_t st = (hipStream_t)stream;
  for (size_t i = 0; i < fm.nodes.size(); i++) {
    GemmParams p = b->params[i];
    if (p.num_m_tiles == 0) continue;
#ifdef K3_GEMM_PROF
    { static const int dbg = getenv("K3_GEMM_DBG") ? atoi(getenv("K3_GEMM_DBG")) : 0; p.dbg = dbg;
      static long long *dbuf = nullptr;
      if (dbg & 8) {
        if (!dbuf) { (void)hipMalloc((void **)&dbuf, 64 * 4 * sizeof(long long)); (void)hipMemset(dbuf, 0, 64 * 4 * sizeof(long long)); }
        p.dbg_buf = dbuf + 4 * (i % 64);
        if (i + 1 == fm.nodes.size()) {    // dump after the last node was launched (previous forward's numbers + this one's so far)
          long long h[64 * 4]; (void)hipDeviceSynchronize(); (void)hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
          for (size_t n = 0; n < fm.nodes.size() && n < 64; n++) if (h[4 * n + 3]) fprintf(stderr,
              "node %zu %-24s blocks %lld  prologue %.0f  mainloop %.0f  epilogue %.0f cycles/block\n", n, fm.nodes[n].name.c_str(), h[4 * n + 3],
                  (double)h[4 * n] / h[4 * n + 3], (double)h[4 * n + 1] / h[4 * n + 3], (double)h[4 * n + 2] / h[4 * n + 3]);
          (void)hipMemset(dbuf, 0, sizeof(h));
        }
      }
    }
#endif
    const k3::FusedNode &f = fm.nodes[i];
    if (f.input < 0) { p.A = d_feats; p.lda = ld_feats; }
    for (const k3::EpiOp &op : f.ops) if (op.kind == k3::kEpiResidual && op.res_node < 0) { p.R = d_feats; p.ldr = ld_feats; }
    if ((int)i == fm.output_node) { p.C = d_out; p.ldc = ld_out; }
    auto planes_of = [&](int node, long long *stride) -> unsigned short * {
      if (b->precision != 2 || node < 0 || node >= (int)b->act_slot.size() || b->act_slot[node] < 0 || !b->wants_planes[node]) return nullptr;
      *stride = (long long)(b->slot_bytes[b->act_slot[node]] / 4); return b->slot_planes[b->act_slot[node]];
    };
    if ((int)i != fm.output_node) p.Cp = planes_of((int)i, &p.cp_stride);
    if (!f.has_gemm) {
      hipLaunchKernelGGL(k3_elementwise_kernel, dim3(p.num_m_tiles), dim3(256), 0, st, p);
    } else {
      const int blocks = p.num_m_tiles * p.num_n_tiles;
      // the epilogue program, matched against the fixed ones
      bool has_map = false; for (int o = 0; o < p.nops; o++) has_map = has_map || p.op_kind[o] == k3::kEpiSigmoid || p.op_kind[o] == k3::kEpiTanh;
      int epi = has_map ? kEpiAnyMap : kEpiAny;
      if (p.nops == 0) epi = kEpiNone;
      else if (p.nops == 3 && p.op_kind[0] == k3::kEpiRelu && p.op_kind[1] == k3::kEpiScaleOffset && p.op_kind[2] == k3::kEpiResidual) epi = kEpiReluScaleRes;
      else if (p.nops == 2 && p.op_kind[0] == k3::kEpiRelu && p.op_kind[1] == k3::kEpiScaleOffset) epi = kEpiReluScale;
      const bool al = p.tiles_per_off > 0, bn96 = b->net->dev[i].bn == 96;
      const size_t lds = 2 * (kBM + (bn96 ? 96 : 128)) * kLdsLd * sizeof(float);
#ifdef K3_GEMM_PROF
      static const size_t lds_pad = getenv("K3_GEMM_LDS_PAD") ? (size_t)atoi(getenv("K3_GEMM_LDS_PAD")) : 0;      // > 6 KB: one workgroup per CU
#else
      const size_t lds_pad = 0;
#endif
      bool launched = false;
      // split-bf16: three planes of 80-byte rows per operand, or the epilogue's staging tile, whichever is larger
      if (b->precision >= 1 && al && b->net->dev[i].Wp && !has_map) {
        const int bn_ = bn96 ? 96 : 128, wm_ = bn96 ? 32 : 64, wn_ = bn96 ? 96 : 64;
        const size_t lds6 = std::max<size_t>((size_t)3 * (kBM + bn_) * 80, (size_t)4 * wm_ * (wn_ + 4) * sizeof(float));
        const int epi6 = (epi == kEpiNone && !bn96) ? (int)kEpiAny : ((epi == kEpiReluScaleRes || epi == kEpiReluScale) && bn96 ? (int)kEpiAny : epi);
        p.Ap = planes_of(f.input, &p.ap_stride);
        const bool pa = p.Ap != nullptr;
#define K3_LAUNCH6(bn, wm, wn, ep) if (!launched && bn96 == (bn == 96) && epi6 == ep) { \
          if (pa) hipLaunchKernelGGL((k3_tdnn_gemm_x6_kernel<bn, wm, wn, ep, true>), dim3(blocks), dim3(256), lds6, st, p, b->net->dev[i].Wp, b->net->dev[i].plane_stride); \
          else hipLaunchKernelGGL((k3_tdnn_gemm_x6_kernel<bn, wm, wn, ep, false>), dim3(blocks), dim3(256), lds6, st, p, b->net->dev[i].Wp, b->net->dev[i].plane_stride); \
          launched = true; }
        K3_X6_VARIANTS(K3_LAUNCH6)
#undef K3_LAUNCH6
        if (launched) p.Cp = nullptr;      // (written by the epilogue)
      }
      // K3_NNET_LDS_FLOOR=<bytes> (environment, experiment of round 6): every GEMM workgroup asks for at least this much LDS.  With 82432 two GEMM workgroups no longer fit on a CU
      // but one fits beside a token-passing lane (81216 B) and two lanes still fit: while GEMM work is queued next to a decoder launch every CU runs one of each.
      static const size_t lds_floor = getenv("K3_NNET_LDS_FLOOR") ? (size_t)atoll(getenv("K3_NNET_LDS_FLOOR")) : 0;
      const size_t lds_launch = std::max(lds + lds_pad, lds_floor);
#define K3_LAUNCH(bn, wm, wn, al_, ep) if (!launched && bn96 == (bn == 96) && al == al_ && epi == ep) { hipLaunchKernelGGL((k3_tdnn_gemm_kernel<bn, wm, wn, al_, ep>), dim3(blocks), dim3((kBM / wm) * (bn / wn) * 64), lds_launch, st, p); launched = true; }
      K3_GEMM_VARIANTS(K3_LAUNCH)
      if (!launched) { epi = has_map ? kEpiAnyMap : kEpiAny; K3_GEMM_VARIANTS(K3_LAUNCH) }      // no fixed-program instantiation for this shape: run-time dispatch
#undef K3_LAUNCH
    }
    if (f.row_op) {      // LogSoftmaxComponent / SoftmaxComponent on the node's rows, in place; on the output node followed by (x - log prior) * acoustic scale
      const bool outn = (int)i == fm.output_node;
      if (f.row_op == 3) hipLaunchKernelGGL(k3_row_normalize_kernel, dim3((unsigned)((b->node_rows[i] + 3) / 4)), dim3(256), 0, st, p.C, (long long)p.ldc,
          (int)b->node_rows[i], f.out_dim, f.row_param,
                                            outn ? b->out_scale : (const float *)nullptr, outn ? b->out_offset : (const float *)nullptr);
      else hipLaunchKernelGGL(k3_row_softmax_kernel, dim3((unsigned)((b->node_rows[i] + 3) / 4)), dim3(256), 0, st, p.C, (long long)p.ldc,
          (int)b->node_rows[i], f.out_dim, f.row_op,
                         outn ? b->out_scale : (const float *)nullptr, outn ? b->out_offset : (const float *)nullptr);
    }
    if (p.Cp && b->precision == 2) {      // a producer that is not the x6 kernel (first layer, element-wise node, row operation): its planes by a pass of their own
      const long long rows = b->node_rows[i]; const int cols4 = (int)(p.ldc / 4);
      hipLaunchKernelGGL(k3_split_planes_kernel, dim3((unsigned)((rows * cols4 + 255) / 256)), dim3(256), 0, st, p.C, (long long)p.ldc, rows, cols4, p.Cp, p.cp_stride);
    }
    K3_HIP_CHECK(hipGetLastError());
  }
  return K3_OK;
}

// ================================================================================================ stateful streaming forward (round 5)
// BatchedStaticNnet3::RunBatch (cudadecoder/batched-static-nnet3.cc:139-233) evaluates every chunk of a stream together with its whole left and right context:
// frames_per_chunk + ~80 input frames for 17 output rows at the benchmark model, and this repository's StaticNnet3 (kaldi_amd/host/k3_online.h) did the same.  The engine
// below evaluates every row of every node exactly ONCE per stream: each node keeps, per channel, the last H_i rows it produced (what its consumers' time offsets still reach
// back to), a pass consumes frames_per_chunk new input frames per channel and produces frames_per_chunk / G_i new rows per node.
//
// Time bookkeeping (channel-local time, t = 0 the stream's first frame).  With R_in = the model's right context and R_i the right extension of node i (the backward sweep of
// the whole-utterance planner), after p passes node i can hold exactly the rows t <= hi_i(p) = C p - 1 - (R_in - R_i) of its grid A_i + k G_i: the new rows of a pass are the
// C / G_i grid points in (hi_i(p - 1), hi_i(p)].  Rows t <= hi_i(0) depend only on input frames at t < 0, which the reference replicates from frame 0 (edge-frame
// replication, nnet-am-decodable-simple.cc:154-163): they all equal the node's response to a constant input, c_i(frame 0).  So a stream starts with its histories SEEDED
// with c_i -- one extra tiny forward over one row per channel whose loads all hit that row -- and its first pass is a pass like any other; the end of a stream is more passes
// whose missing input frames replicate the last one.  Every row is the same arithmetic on the same operands as in the whole-utterance forward (same kernel, same K order; the
// rows of a tile are independent), so the outputs are bit-identical to k3_nnet_forward over the whole utterance (tests/test_nnet_gpu.py).
//
// Layout: node buffers are TIME-major, channel-minor -- row (h, c) = h * num_channels + c, h = 0 .. H_i - 1 the history, then the C / G_i new time steps -- so that a 128-row
// slab of the GEMM is 128 channels of one time step (full slabs whatever C / G_i is: 17 rows per channel and pass would fill 13 % of a per-channel slab), a time offset o is a
// constant row shift (o / G_src) * num_channels, and moving the history is one strided copy per node.  Channels that sit a pass out keep their histories (the move is masked).
struct k3_nnet_stream {
  k3_nnet *net = nullptr; int nch = 0, C = 0, s = 1, in_dim = 0, out_dim = 0, H_in = 0, N_out = 0, first_out = 0, R_in = 0;
  std::unique_ptr<k3_nnet_batch> pass, seed;      // the per-pass plan and the constant-response plan (one row per channel and node), both run by forward_impl
  float *in_buf = nullptr, *const_in = nullptr; long long ld_in = 0;
  struct NodeBuf { float *buf = nullptr, *cst = nullptr; int ld = 0, dim = 0, H = 0, N = 0; };
  std::vector<NodeBuf> nb; NodeBuf *d_nb = nullptr; int n_nb = 0;      // device copy of the nodes that keep a history (incl. the input as entry 0)
  // per-call host arrays (row offsets, active flags, reset lists): page-locked ring + device copies
  static constexpr int kSlots = 4;
  struct Slot { char *h = nullptr, *d = nullptr; hipEvent_t ev = nullptr; bool used = false; } slot[kSlots]; unsigned seq = 0; size_t slot_bytes = 0;
  std::vector<void *> allocs;
  ~k3_nnet_stream() {
    for (void *p : allocs) (void)hipFree(p);
    for (Slot &g : slot) {
      if (g.h) (void)hipHostFree(g.h);
      if (g.d) (void)hipFree(g.d);
      if (g.ev) (void)hipEventDestroy(g.ev);
    }
  }
};

namespace {
// new input rows of a pass into the time-major input buffer: channel c's rows [off[c], off[c + 1]) of `src` (at most C; fewer = the stream's audio has ended: the missing frames
// replicate the last real one -- this pass's, or the newest history row when the channel has none left); inactive channels are skipped
__global__ __launch_bounds__(256) void k3_stream_prep_kernel(const float *src, long long lds, const long long *start, const int *count, float *in_buf,
    long long ld, int nch, int C, int H, int dim4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x; const long long per_t = (long long)nch * dim4;
  if (i >= (long long)C * per_t) return;
  const int t = (int)(i / per_t), c = (int)((i % per_t) / dim4), q = (int)(i % dim4);
  const int n = count[c]; if (n < 0) return;      // (the channel sits this pass out)
  const long long r0 = start[c];
  const float4 *from = n > 0 ? reinterpret_cast<const float4 *>(src + (r0 + (t < n ? t : n - 1)) * lds) :
      reinterpret_cast<const float4 *>(in_buf + ((long long)(H - 1) * nch + c) * ld);
  reinterpret_cast<float4 *>(in_buf + ((long long)(H + t) * nch + c) * ld)[q] = from[q];
}
// behind a pass: every node's newest H rows become its history (active channels only).  One launch for all nodes: blockIdx.y = node, x over (h, channel, float4 column)
__global__ __launch_bounds__(256) void k3_stream_shift_kernel(const k3_nnet_stream::NodeBuf *nodes, const int *count, int nch) {
  const k3_nnet_stream::NodeBuf nb = nodes[blockIdx.y]; const int d4 = nb.dim / 4;
  const long long n = (long long)nb.H * nch * d4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int h = (int)(i / ((long long)nch * d4)), c = (int)((i / d4) % nch), q = (int)(i % d4);
    if (count[c] < 0) continue;
    // (H <= N: source and destination never overlap)
    reinterpret_cast<float4 *>(nb.buf + ((long long)h * nch + c) * nb.ld)[q] =
        reinterpret_cast<const float4 *>(nb.buf + ((long long)(nb.N + h) * nch + c) * nb.ld)[q];
  }
}
// a restarted channel's histories: H copies of the node's constant response (entry 0: the input itself, i.e. frame 0 replicated)
__global__ __launch_bounds__(256) void k3_stream_seed_kernel(const k3_nnet_stream::NodeBuf *nodes, const int *channels, int n_ch, int nch) {
  const k3_nnet_stream::NodeBuf nb = nodes[blockIdx.y]; const int d4 = nb.dim / 4;
  const long long n = (long long)nb.H * n_ch * d4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int h = (int)(i / ((long long)n_ch * d4)), c = channels[(i / d4) % n_ch], q = (int)(i % d4);
    reinterpret_cast<float4 *>(nb.buf + ((long long)h * nch + c) * nb.ld)[q] = reinterpret_cast<const float4 *>(nb.cst + (long long)c * nb.ld)[q];
  }
}
// frame 0 of the restarted channels into their rows of the constant-input matrix
__global__ __launch_bounds__(256) void k3_stream_const_in_kernel(const float *first, long long ld, const int *channels, int n_ch, float *const_in, long long ld_in, int dim4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i >= (long long)n_ch * dim4) return;
  const int k = (int)(i / dim4), q = (int)(i % dim4);
  reinterpret_cast<float4 *>(const_in + (long long)channels[k] * ld_in)[q] = reinterpret_cast<const float4 *>(first + (long long)k * ld)[q];
}
int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
}  // namespace

extern "C" void k3_nnet_stream_destroy(k3_nnet_stream *s) { delete s; }

extern "C" int k3_nnet_stream_create(k3_nnet *net, int32_t num_channels, int32_t frames_per_chunk, int32_t subsampling, const float *h_log_priors,
    float acoustic_scale, k3_nnet_stream **out) {
  K3_REQUIRE(net && out && num_channels > 0 && frames_per_chunk > 0 && subsampling >= 1 && frames_per_chunk % subsampling == 0,
      "k3_nnet_stream_create: bad argument (frames_per_chunk must be a positive multiple of the subsampling factor)");
  K3_REQUIRE(net->fm.input_dim % 4 == 0, "k3_nnet_stream_create: the input dimension must be a multiple of 4 (vector loads)");
  const k3::FusedModel &fm = net->fm; const int nn = (int)fm.nodes.size(), C = frames_per_chunk, NCH = num_channels;
  if (fm.ivector_dim > 0) {
    k3::set_error("k3_nnet_stream_create: models with an i-vector input are evaluated chunk by chunk with their context (k3_nnet_batch_create_ivector)");
    return K3_ERR_UNSUPPORTED;
  }
  // ---- time grids: the whole-utterance planner's backward sweep (A = first time, R = right extension, G = step)
  std::vector<int> A(nn, 0), R(nn, 0), G(nn, 0); std::vector<char> used(nn, 0); int R_in = -(1 << 30), A_in = 1 << 30;
  A[fm.output_node] = 0; R[fm.output_node] = 0; G[fm.output_node] = subsampling; used[fm.output_node] = 1;
  {
    std::vector<std::vector<std::pair<int, int>>> anchors(nn); std::vector<int> rmax(nn, -(1 << 30));
    for (int i = nn - 1; i >= 0; i--) {
      if (i != fm.output_node) {
        if (anchors[i].empty()) continue;
        int a = 1 << 30, g = 0;
        for (auto &an : anchors[i]) a = std::min(a, an.first);
        for (auto &an : anchors[i]) { g = gcd_i(g, an.second); g = gcd_i(g, an.first - a); }
        A[i] = a; G[i] = g; R[i] = rmax[i]; used[i] = 1;
      }
      const k3::FusedNode &f = fm.nodes[i];
      auto contribute = [&](int src, int off) {
        if (src < 0) {
          R_in = std::max(R_in, R[i] + off);
          A_in = std::min(A_in, A[i] + off);
          return;
        }
        anchors[src].push_back({A[i] + off, G[i]});
        rmax[src] = std::max(rmax[src], R[i] + off);
      };
      for (int o : f.offsets) contribute(f.input, o);
      for (const k3::EpiOp &op : f.ops) if (op.kind == k3::kEpiResidual) contribute(op.res_node, 0);
    }
  }
  K3_REQUIRE(R_in > -(1 << 30), "k3_nnet_stream_create: no node reads the network input");
  // node -1 (the input) as index nn in the tables below
  auto Gx = [&](int i) { return i < 0 ? 1 : G[i]; };
  std::vector<int> first(nn + 1, 0), Nn(nn + 1, 0), H(nn + 1, 0);
  auto idx = [&](int i) { return i < 0 ? nn : i; };
  for (int i = -1; i < nn; i++) {
    if (i >= 0 && !used[i]) continue;
    const int g = Gx(i), a = i < 0 ? 0 : A[i], r = i < 0 ? R_in : R[i];
    if (C % g != 0) {
      k3::set_error("k3_nnet_stream_create: frames_per_chunk %d is not a multiple of node %s's time step %d", C, i < 0 ? "input" : fm.nodes[i].name.c_str(), g);
      return K3_ERR_UNSUPPORTED;
    }
    const int lo = -1 - (R_in - r);                                   // hi_i(0): the newest row before the first pass
    first[idx(i)] = a + (floor_div(lo - a, g) + 1) * g;               // smallest grid point > lo
    Nn[idx(i)] = C / g;
  }
  for (int i = 0; i < nn; i++) {
    if (!used[i]) continue;
    const k3::FusedNode &f = fm.nodes[i];
    if (f.row_op && i != fm.output_node) {
      k3::set_error("k3_nnet_stream_create: node %s applies a row operation in place (NormalizeComponent / softmax inside the network): not supported by the stateful engine", f.name.c_str());
      return K3_ERR_UNSUPPORTED;
    }
    auto need = [&](int src, int o_min, int o_max) {
      const int gs = Gx(src), fs = first[idx(src)], fi = first[i];
      if ((fi + o_min - fs) % gs != 0 || G[i] % gs != 0) return false;
      // grid points of the source in [fi + o_min, fs): rows of its history this node still reads
      {
        const int d = fs - (fi + o_min);
        H[idx(src)] = std::max(H[idx(src)], d > 0 ? d / gs : 0);
      }
      return fi + (Nn[i] - 1) * G[i] + o_max <= fs + (Nn[idx(src)] - 1) * gs;      // the newest row it reads exists
    };
    bool ok = true;
    {
      int lo = 1 << 30, hi = -(1 << 30);
      for (int o : f.offsets) {
        lo = std::min(lo, o);
        hi = std::max(hi, o);
        if ((first[i] + o - first[idx(f.input)]) % Gx(f.input) != 0) ok = false;
      }
      ok = ok && need(f.input, lo, hi);
    }
    // (the bypass may come from a finer grid: G_i % G_res == 0 is all it takes)
    for (const k3::EpiOp &op : f.ops) if (op.kind == k3::kEpiResidual) ok = ok && need(op.res_node, 0, 0);
    if (!ok) {
      k3::set_error("k3_nnet_stream_create: the time grids of node %s (first %d, step %d) and its inputs do not line up for incremental evaluation",
          f.name.c_str(), first[i], G[i]);
      return K3_ERR_UNSUPPORTED;
    }
  }
  H[nn] = std::max(H[nn], 1);      // (the input keeps at least its newest frame: what the end of a stream replicates)
  for (int i = -1; i < nn; i++) if ((i < 0 || used[i]) && H[idx(i)] > Nn[idx(i)]) {
    k3::set_error("k3_nnet_stream_create: frames_per_chunk %d is shorter than the history a node keeps (%d rows)", C, H[idx(i)]);
    return K3_ERR_UNSUPPORTED;
  }

  if (getenv("K3_NNET_STREAM_PLAN")) for (int i = -1; i < nn; i++) if (i < 0 || used[i])      // development aid: the plan, before anything is allocated
    fprintf(stderr, "k3_nnet_stream plan: node %-22s A %4d R %3d G %d first %4d new %3d history %d\n", i < 0 ? "input" : fm.nodes[i].name.c_str(),
        i < 0 ? A_in : A[i], i < 0 ? R_in : R[i], Gx(i), first[idx(i)], Nn[idx(i)], H[idx(i)]);
  { const int rc = ensure_uploaded(net); if (rc) return rc; }      // (everything above is host arithmetic: a model the engine cannot run is refused without touching the device)
  std::unique_ptr<k3_nnet_stream> S(new k3_nnet_stream());
  S->net = net;
  S->nch = NCH;
  S->C = C;
  S->s = subsampling;
  S->in_dim = fm.input_dim;
  S->out_dim = fm.output_dim;
  S->H_in = H[nn];
  S->N_out = Nn[fm.output_node];
  S->first_out = first[fm.output_node];
  S->R_in = R_in;
  auto dalloc = [&](size_t bytes, float **p) -> int {
    K3_HIP_CHECK(hipMalloc((void **)p, std::max<size_t>(bytes, 256)));
    S->allocs.push_back(*p);
    K3_HIP_CHECK(hipMemset(*p, 0, std::max<size_t>(bytes, 256)));
    return K3_OK;
  };
  S->nb.assign(nn + 1, k3_nnet_stream::NodeBuf());
  S->ld_in = (long long)align_up(fm.input_dim, 4);
  { int rc = dalloc((size_t)(H[nn] + C) * NCH * S->ld_in * 4, &S->in_buf); if (rc) return rc; rc = dalloc((size_t)NCH * S->ld_in * 4, &S->const_in); if (rc) return rc; }
  S->nb[nn] = {S->in_buf, S->const_in, (int)S->ld_in, (int)S->ld_in, H[nn], C};
  for (int i = 0; i < nn; i++) {
    if (!used[i] || i == fm.output_node) continue;
    k3_nnet_stream::NodeBuf &b = S->nb[i]; b.ld = (int)align_up(fm.nodes[i].out_dim, 4); b.dim = b.ld; b.H = H[i]; b.N = Nn[i];
    int rc = dalloc((size_t)(H[i] + Nn[i]) * NCH * b.ld * 4, &b.buf); if (rc) return rc;
    rc = dalloc((size_t)NCH * b.ld * 4, &b.cst); if (rc) return rc;
  }
  { std::vector<k3_nnet_stream::NodeBuf> keep; keep.push_back(S->nb[nn]); for (int i = 0; i < nn; i++) if (S->nb[i].buf && S->nb[i].H > 0) keep.push_back(S->nb[i]);
    S->n_nb = (int)keep.size(); K3_HIP_CHECK(hipMalloc((void **)&S->d_nb, keep.size() * sizeof(keep[0]))); S->allocs.push_back(S->d_nb);
    K3_HIP_CHECK(hipMemcpy(S->d_nb, keep.data(), keep.size() * sizeof(keep[0]), hipMemcpyHostToDevice)); }
  S->slot_bytes = align_up(sizeof(long long) * (NCH + 1), 16) + align_up(sizeof(int) * NCH, 16) * 2;
  for (auto &g : S->slot) {
    K3_HIP_CHECK(hipHostMalloc((void **)&g.h, S->slot_bytes, hipHostMallocDefault));
    K3_HIP_CHECK(hipMalloc((void **)&g.d, S->slot_bytes));
    K3_HIP_CHECK(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
  }

  // ---- the two plans, as k3_nnet_batch objects that forward_impl runs: `pass` (C / G_i time steps of num_channels rows per node) and `seed` (one row per channel and node,
  // every time offset reading that row)
  float *out_scale = nullptr, *out_offset = nullptr;
  const bool out_xform = (h_log_priors != nullptr) || acoustic_scale != 1.0f;
  const int blocks_per_t = (NCH + kBM - 1) / kBM;
  for (int which = 0; which < 2; which++) {
    std::unique_ptr<k3_nnet_batch> b(new k3_nnet_batch());
    b->net = net; b->num_utts = NCH; b->subsampling = subsampling; b->params.resize(nn); b->seq_bias.assign(nn, nullptr); b->seq_bias_ld.assign(nn, 0); b->node_rows.assign(nn, 0);
    if (which == 0 && out_xform) {
      std::vector<float> sc(fm.output_dim, acoustic_scale), of(fm.output_dim, 0.0f);
      if (h_log_priors) for (int i = 0; i < fm.output_dim; i++) of[i] = -h_log_priors[i] * acoustic_scale;
      int rc = upload(&b->allocs, sc, &out_scale); if (rc) return rc; rc = upload(&b->allocs, of, &out_offset); if (rc) return rc;
      b->out_scale = out_scale; b->out_offset = out_offset;
    }
    for (int i = 0; i < nn; i++) {
      GemmParams &p = b->params[i]; memset(&p, 0, sizeof(p));
      if (!used[i] || (which == 1 && i == fm.output_node)) { p.num_m_tiles = 0; continue; }      // (the output node keeps no history: nothing to seed)
      const k3::FusedNode &f = fm.nodes[i]; const DeviceNode &d = net->dev[i]; const int src = f.input;
      const k3_nnet_stream::NodeBuf &sb = S->nb[idx(src)];
      p.in_dim = f.in_dim; p.noff = (int)f.offsets.size(); p.row_stride = 1;
      for (int o = 0; o < p.noff; o++) p.shifts[o] = which == 1 ? 0 : ((first[i] + f.offsets[o] - first[idx(src)]) / Gx(src)) * NCH;
      p.tiles_per_off = (f.in_dim % kBK == 0) ? f.in_dim / kBK : 0; p.tiles_per_seg = 384 / kBK;
      p.A = src < 0 ? nullptr : (which == 1 ? sb.cst : sb.buf); p.lda = src < 0 ? 0 : sb.ld;      // (the network input is patched in by forward_impl)
      p.W = d.W; p.ldw = d.ldw; p.Ktot = p.noff * f.in_dim; p.N = f.out_dim; p.bias = d.bias;
      p.C = i == fm.output_node ? nullptr : (which == 1 ? S->nb[i].cst : S->nb[i].buf); p.ldc = i == fm.output_node ? 0 : S->nb[i].ld;
      p.nops = (int)f.ops.size(); int res = -2;
      for (int o = 0; o < p.nops; o++) {
        p.op_kind[o] = f.ops[o].kind;
        p.op_scale[o] = d.op_scale[o];
        p.op_offset[o] = d.op_offset[o];
        if (f.ops[o].kind == k3::kEpiResidual) {
          res = f.ops[o].res_node;
          p.res_scale = f.ops[o].res_scale;
        }
      }
      if (i == fm.output_node && out_xform && f.row_op == 0) {
        if (p.nops >= kMaxOps) { k3::set_error("k3_nnet_stream_create: too many epilogue ops on the output node"); return K3_ERR_UNSUPPORTED; }
        p.op_kind[p.nops] = k3::kEpiScaleOffset; p.op_scale[p.nops] = out_scale; p.op_offset[p.nops] = out_offset; p.nops++;
      }
      const k3_nnet_stream::NodeBuf *rb = res >= -1 ? &S->nb[idx(res)] : nullptr;
      if (rb) { p.R = res < 0 ? nullptr : (which == 1 ? rb->cst : rb->buf); p.ldr = res < 0 ? 0 : rb->ld; p.res_row_stride = 1; }
      const int steps = which == 1 ? 1 : Nn[i], Hi = i == fm.output_node ? 0 : H[i];
      const long long in_rows = which == 1 ? NCH : (long long)(sb.H + sb.N) * NCH;
      std::vector<TileDesc> tiles;
      for (int k = 0; k < steps; k++) for (int bl = 0; bl < blocks_per_t; bl++) {
        TileDesc t; memset(&t, 0, sizeof t);
        t.nrows = std::min(kBM, NCH - bl * kBM); t.split = t.nrows; t.bias_row = 0;
        t.out_row0 = which == 1 ? bl * kBM : (Hi + k) * NCH + bl * kBM;
        t.in_base = which == 1 ? bl * kBM : (sb.H + k * (G[i] / Gx(src))) * NCH + bl * kBM; t.in_lo = 0; t.in_hi = (int)in_rows - 1;
        if (rb) t.res_base = which == 1 ? bl * kBM : (rb->H + (first[i] + k * G[i] - first[idx(res)]) / Gx(res)) * NCH + bl * kBM;
        t.in_base2 = t.in_base; t.in_lo2 = t.in_lo; t.in_hi2 = t.in_hi; t.res_base2 = t.res_base; t.bias_row2 = 0;
        tiles.push_back(t);
      }
      if (f.has_gemm) b->flops += 2.0 * steps * NCH * (double)p.Ktot * f.out_dim;
      b->node_rows[i] = (long long)steps * NCH;
      TileDesc *dt = nullptr; int rc = upload(&b->allocs, tiles, &dt); if (rc) return rc;
      p.tiles = dt; p.num_m_tiles = (int)tiles.size(); p.num_n_tiles = f.has_gemm ? d.npad / d.bn : 1;
    }
    (which == 0 ? S->pass : S->seed) = std::move(b);
  }
  *out = S.release();
  return K3_OK;
}

extern "C" int k3_nnet_stream_get_info(const k3_nnet_stream *s, k3_nnet_stream_info *info) {
  K3_REQUIRE(s && info, "k3_nnet_stream_get_info: null argument");
  info->num_channels = s->nch;
  info->frames_per_chunk = s->C;
  info->subsampling = s->s;
  info->output_rows_per_pass = s->N_out;
  info->first_output_time = s->first_out;
  info->right_context = s->R_in;
  info->input_history = s->H_in; info->flops_per_pass = s->pass->flops;
  return K3_OK;
}

namespace {
int stream_slot(k3_nnet_stream *s, k3_nnet_stream::Slot **out) {
  k3_nnet_stream::Slot &g = s->slot[s->seq++ % k3_nnet_stream::kSlots];
  if (g.used) K3_HIP_CHECK(hipEventSynchronize(g.ev));
  *out = &g;
  return K3_OK;
}
}

// The listed channels start new streams: their histories become the response to a constant input -- d_first_frames row i = frame 0 of channel h_channels[i]'s stream (what the
// reference replicates to the left of the utterance).  Queued on `stream`; the next k3_nnet_stream_forward on that stream sees the new state.
extern "C" int k3_nnet_stream_reset(k3_nnet_stream *s, const int32_t *h_channels, int32_t n, const float *d_first_frames, int64_t ld, void *stream) {
  K3_REQUIRE(s && (n == 0 || (h_channels && d_first_frames)) && n >= 0 && n <= s->nch && ld >= s->in_dim, "k3_nnet_stream_reset: bad argument");
  if (n == 0) return K3_OK;
  for (int i = 0; i < n; i++) K3_REQUIRE(h_channels[i] >= 0 && h_channels[i] < s->nch, "k3_nnet_stream_reset: channel out of range");
  hipStream_t st = (hipStream_t)stream; k3_nnet_stream::Slot *g; { const int rc = stream_slot(s, &g); if (rc) return rc; }
  int *hc = reinterpret_cast<int *>(g->h); memcpy(hc, h_channels, sizeof(int) * n);
  K3_HIP_CHECK(hipMemcpyAsync(g->d, g->h, sizeof(int) * n, hipMemcpyHostToDevice, st));
  const int *dc = reinterpret_cast<const int *>(g->d);
  // frame 0 of the restarted channels into their rows of the constant-input matrix (rows of the other channels keep whatever they hold: their results are not used)
  K3_REQUIRE(ld % 4 == 0 && ((uintptr_t)d_first_frames & 15) == 0, "k3_nnet_stream_reset: d_first_frames must be 16-byte aligned with ld % 4 == 0");
  {
    const int dim4 = s->in_dim / 4;
    const long long m = (long long)n * dim4;
    hipLaunchKernelGGL(k3_stream_const_in_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d_first_frames, (long long)ld, dc, n, s->const_in,
        s->ld_in, dim4);
  }
  { const int rc = forward_impl(s->seed.get(), s->const_in, s->ld_in, s->const_in /* unused: the output node is not part of the seed plan */, s->out_dim > 0 ? (int64_t)align_up(s->out_dim, 4) : 4, stream); if (rc) return rc; }
  hipLaunchKernelGGL(k3_stream_seed_kernel, dim3(64, (unsigned)s->n_nb), dim3(256), 0, st, s->d_nb, dc, n, s->nch);
  K3_HIP_CHECK(hipGetLastError());
  K3_HIP_CHECK(hipEventRecord(g->ev, st)); g->used = true;
  return K3_OK;
}

// One pass.  Channel c with h_row_count[c] >= 0 takes part and consumes rows h_row_start[c] .. + h_row_count[c] of d_new as its next frames: exactly frames_per_chunk of them while
// its stream goes on, fewer (or none) once its audio has ended -- the missing frames replicate the last real one; h_row_count[c] < 0: the channel sits the pass out and keeps its
// state.  d_out [output_rows_per_pass * num_channels x ld_out], time-major: row k * num_channels + c is channel c's output at time first_output_time + (passes of c before this
// one) * frames_per_chunk + k * subsampling; rows of channels that sat out are undefined.
extern "C" int k3_nnet_stream_forward(k3_nnet_stream *s, const float *d_new, int64_t ld_new, const int64_t *h_row_start, const int32_t *h_row_count,
    float *d_out, int64_t ld_out, void *stream) {
  K3_REQUIRE(s && h_row_start && h_row_count && d_out && ld_out >= s->out_dim && ld_new >= s->in_dim && ld_new % 4 == 0, "k3_nnet_stream_forward: bad argument");
  long long total = 0;
  for (int c = 0; c < s->nch; c++) {
    K3_REQUIRE(h_row_count[c] <= s->C && (h_row_count[c] <= 0 || h_row_start[c] >= 0),
        "k3_nnet_stream_forward: a channel takes at most frames_per_chunk rows per pass");
    total += std::max(0, h_row_count[c]);
  }
  K3_REQUIRE(total == 0 || (d_new && ((uintptr_t)d_new & 15) == 0), "k3_nnet_stream_forward: d_new must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream; k3_nnet_stream::Slot *g; { const int rc = stream_slot(s, &g); if (rc) return rc; }
  const size_t o_cnt = align_up(sizeof(long long) * (s->nch + 1), 16);
  memcpy(g->h, h_row_start, sizeof(long long) * s->nch); memcpy(g->h + o_cnt, h_row_count, sizeof(int) * s->nch);
  K3_HIP_CHECK(hipMemcpyAsync(g->d, g->h, s->slot_bytes, hipMemcpyHostToDevice, st));
  const long long *d_start = reinterpret_cast<const long long *>(g->d); const int *d_cnt = reinterpret_cast<const int *>(g->d + o_cnt);
  const int dim4 = s->in_dim / 4; const long long n = (long long)s->C * s->nch * dim4;
  hipLaunchKernelGGL(k3_stream_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_new ? d_new : s->in_buf, (long long)ld_new, d_start, d_cnt,
      s->in_buf, s->ld_in, s->nch, s->C, s->H_in, dim4);
  { const int rc = forward_impl(s->pass.get(), s->in_buf, s->ld_in, d_out, ld_out, stream); if (rc) return rc; }
  hipLaunchKernelGGL(k3_stream_shift_kernel, dim3(128, (unsigned)s->n_nb), dim3(256), 0, st, s->d_nb, d_cnt, s->nch);
  K3_HIP_CHECK(hipGetLastError());
  K3_HIP_CHECK(hipEventRecord(g->ev, st)); g->used = true;
  return K3_OK;
}
