// k3_matrix.hip -- the CuMatrixBase<float> operations the nnet3 forward pass of a TDNN / TDNN-F model executes when it runs
// through Kaldi's generic NnetComputer (AddMatMat, CopyRowsFromVec, CopyFromMat, ApplyFloor, MulColsVec,
// AddVecToRows, Scale, AddMat, CopyRows, AddRows, SetZero/Set), as gfx950 kernels behind a C ABI.  They let a Kaldi build keep its
// own graph compiler and executor (nnet3/nnet-compute.cc:236-459) and only swap the device kernels under CuMatrix
// (cudamatrix/cu-matrix.cc, cu-kernels.cu); the fused path in k3_nnet.hip is the fast path for the same models.
// All matrices are row-major float32 with a leading dimension (CuMatrixBase::Stride()), device pointers.
#include "k3_common.h"
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// C = alpha * op(A) * op(B) + beta * C   (CuMatrixBase::AddMatMat, cu-matrix.cc:1329-1375 -> cublas_gemm).
// 64 x 64 x 16 tiles, 4 wavefronts of one 32x32 FP32-MFMA tile each; operands go through LDS element-wise so that any
// transpose / stride / ragged edge is handled by the index function.  k is consumed in ascending order.
__global__ __launch_bounds__(256) void k3_gemm_generic_kernel(int M, int N, int K, float alpha, const float *A, long long lda, int ta,
                                                              const float *B, long long ldb, int tb, float beta, float *C, long long ldc, float *W, int Kc) {
  __shared__ float As[64][17], Bs[16][65];
  // split-K (W != null; long-K products with few output tiles, e.g. a weight gradient): block z multiplies k in [z Kc, (z + 1) Kc) into its own M x N plane of W (alpha 1,
  // beta 0; Kc is a multiple of the 384-wide accumulation blocks); k3_gemm_splitk_reduce_kernel adds the planes in ascending z
  int kb_ = 0, ke_ = K;
  if (W) { kb_ = (int)blockIdx.z * Kc; ke_ = min(K, kb_ + Kc); C = W + (long long)blockIdx.z * M * N; ldc = N; alpha = 1.0f; beta = 0.0f; }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  // The accumulation order is the contract: like the blocked CPU sgemm the reference calls, k runs in ascending order
  // inside blocks of 384 and every block's sum is added to the running result, which starts from beta * C.
  f32x16 acc, total;
  const int col = n0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    acc[r] = 0.0f; total[r] = 0.0f;
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (beta != 0.0f && row < M && col < N) total[r] = beta * C[(long long)row * ldc + col];
  }
  for (int k0 = kb_; k0 < ke_; k0 += 16) {
    if (k0 != kb_ && k0 % 384 == 0) {
#pragma unroll
      for (int r = 0; r < 16; r++) { total[r] += alpha * acc[r]; acc[r] = 0.0f; }
    }
    for (int e = tid; e < 64 * 16; e += 256) {
      // consecutive threads walk the operand's CONTIGUOUS dimension (k for A, n for B; the other one when the operand is transposed): the LDS strides 17 / 65 are odd, so
      // either order is conflict-free there
      const int r = ta ? (e & 63) : (e >> 4), k = ta ? (e >> 6) : (e & 15), gm = m0 + r, gk = k0 + k;
      As[r][k] = (gm < M && gk < ke_) ? (ta ? A[(long long)gk * lda + gm] : A[(long long)gm * lda + gk]) : 0.0f;
      const int kb = tb ? (e & 15) : (e >> 6), c = tb ? (e >> 4) : (e & 63), gn = n0 + c, gk2 = k0 + kb;
      Bs[kb][c] = (gn < N && gk2 < ke_) ? (tb ? B[(long long)gn * ldb + gk2] : B[(long long)gk2 * ldb + gn]) : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      const int k = 2 * kk + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[wm * 32 + (lane & 31)][k], Bs[k][wn * 32 + (lane & 31)], acc, 0, 0, 0);
    }
    __syncthreads();
  }
  if (col < N) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M) C[(long long)row * ldc + col] = total[r] + alpha * acc[r];
    }
  }
}

// The same product for operands that allow 16-byte loads (16-byte aligned base pointers, leading dimensions a multiple of 4 floats -- whole CuMatrix objects, not arbitrary column
// ranges): 128 x 128 x 16 tiles, 4 wavefronts of 2 x 2 32x32 MFMA tiles, every operand fetched with dwordx4 loads along its CONTIGUOUS dimension (k when the operand is used as
// stored for A / transposed for B, the m / n dimension otherwise), the next k-tile in registers while the present one is multiplied (one barrier per k-tile, two LDS buffers).
// Same accumulation order as k3_gemm_generic_kernel: k ascending, 384-wide blocks added to a total that starts at beta C.  The training pass's GEMMs (activations x weights,
// output derivative x weights, output derivative^T x activations over a minibatch) all qualify; the generic kernel stays the fall-back.
template <int T, int TA, int TB, int BK>
__global__ __launch_bounds__(256) void k3_gemm_tile_kernel(int M, int N, int K, float alpha, const float *__restrict__ A, long long lda,
    const float *__restrict__ B, long long ldb, float beta,
                                                           float *C, long long ldc, float *W, int Kc, int kblk) {
  // a (64 T) x (64 T) output tile per workgroup, T = 1 or 2: four wavefronts as 2 x 2, each T x T MFMA 32x32 blocks; k-tiles of BK (16 / 32 / 64) double-buffered through LDS.
  // The sum over k is formed in blocks of kblk (block sums added in ascending order), so a split of K at multiples of kblk gives the bits of the unsplit product.
  constexpr int TS = 64 * T;
  constexpr int U = T * BK / 16, Q = BK / 4;      // dwordx4 per thread and operand per k-tile; dwordx4 per row of BK k's
  __shared__ float As[2][TS][BK + 1], Bs[2][BK][TS + 4];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  int kb_ = 0, ke_ = K;
  if (W) { kb_ = (int)blockIdx.z * Kc; ke_ = min(K, kb_ + Kc); C = W + (long long)blockIdx.z * M * N; ldc = N; alpha = 1.0f; beta = 0.0f; }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
  f32x16 acc[T][T], total[T][T], cin[T][T];
#pragma unroll
  for (int i = 0; i < T; i++)
#pragma unroll
    for (int j = 0; j < T; j++) {
      const int col = n0 + wn * 32 * T + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[i][j][r] = 0.0f; total[i][j][r] = 0.0f; cin[i][j][r] = 0.0f; }
      (void)col;
    }
  // beta C is requested under the MFMAs of the last k-tile (at the top of the kernel the operand loads queued behind these 16 T^2 row pieces: +10 us on a [4736 x 768] product)
  auto load_c = [&]() {
#pragma unroll
    for (int i = 0; i < T; i++)
#pragma unroll
      for (int j = 0; j < T; j++) {
        const int col = n0 + wn * 32 * T + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = m0 + wm * 32 * T + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < M && col < N) cin[i][j][r] = C[(long long)row * ldc + col];
        }
      }
  };
  // the running sum starts from beta C (the reference-like order of DESIGN.md 2.1): with one accumulation block that is (beta C) + alpha acc whenever C arrives; with several,
  // C is requested under the first k-tile's MFMAs and seeds `total` at the first block boundary
  bool have_c = beta == 0.0f, seeded = false;
  const bool several = (ke_ - 1) / kblk > kb_ / kblk;
  f32x4 ra[U], rb[U];
  // one k-tile of both operands into registers: T dwordx4 per thread and operand; elements outside the matrices / the k range are zero
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int f = tid + 256 * u;
      {
        const int r = TA ? (f & (16 * T - 1)) * 4 : f / Q, k = TA ? f / (16 * T) : (f % Q) * 4;      // TA: k-major rows of TS m's; else m-major rows of BK k's
        const int gm = m0 + r, gk = k0 + k; f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (TA) {
          if (gk < ke_) {
            const float *src = A + (long long)gk * lda + gm;
            if (gm + 3 < M) v = *reinterpret_cast<const f32x4 *>(src);
            else {
              for (int e = 0; e < 4; e++) if (gm + e < M) v[e] = src[e];
            }
          }
        } else {
          if (gm < M) {
            const float *src = A + (long long)gm * lda + gk;
            if (gk + 3 < ke_) v = *reinterpret_cast<const f32x4 *>(src);
            else {
              for (int e = 0; e < 4; e++) if (gk + e < ke_) v[e] = src[e];
            }
          }
        }
        ra[u] = v;
      }
      {
        const int c = TB ? f / Q : (f & (16 * T - 1)) * 4, k = TB ? (f % Q) * 4 : f / (16 * T);      // TB: n-major rows of BK k's; else k-major rows of TS n's
        const int gn = n0 + c, gk = k0 + k; f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (TB) {
          if (gn < N) {
            const float *src = B + (long long)gn * ldb + gk;
            if (gk + 3 < ke_) v = *reinterpret_cast<const f32x4 *>(src);
            else {
              for (int e = 0; e < 4; e++) if (gk + e < ke_) v[e] = src[e];
            }
          }
        } else {
          if (gk < ke_) {
            const float *src = B + (long long)gk * ldb + gn;
            if (gn + 3 < N) v = *reinterpret_cast<const f32x4 *>(src);
            else {
              for (int e = 0; e < 4; e++) if (gn + e < N) v[e] = src[e];
            }
          }
        }
        rb[u] = v;
      }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int f = tid + 256 * u;
      if (TA) { const int r = (f & (16 * T - 1)) * 4, k = f / (16 * T); for (int e = 0; e < 4; e++) As[buf][r + e][k] = ra[u][e]; }
      else { const int r = f / Q, k = (f % Q) * 4; for (int e = 0; e < 4; e++) As[buf][r][k + e] = ra[u][e]; }
      if (TB) { const int c = f / Q, k = (f % Q) * 4; for (int e = 0; e < 4; e++) Bs[buf][k + e][c] = rb[u][e]; }
      else { const int c = (f & (16 * T - 1)) * 4, k = f / (16 * T); for (int e = 0; e < 4; e++) Bs[buf][k][c + e] = rb[u][e]; }
    }
  };
  fetch(kb_); stage(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = kb_; k0 < ke_; k0 += BK, buf ^= 1) {
    if (k0 != kb_ && k0 % kblk == 0) {
#pragma unroll
      for (int i = 0; i < T; i++)
#pragma unroll
        for (int j = 0; j < T; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) { total[i][j][r] = (seeded ? total[i][j][r] : beta == 0.0f ? 0.0f : beta * cin[i][j][r]) + alpha * acc[i][j][r]; acc[i][j][r] = 0.0f; }
      seeded = true;
    }
    const bool more = k0 + BK < ke_;
    if (more) fetch(k0 + BK);      // in flight while this tile is multiplied
    if (!have_c && (several || !more)) { load_c(); have_c = true; }
#pragma unroll
    for (int kk = 0; kk < BK / 2; kk++) {
      const int k = 2 * kk + (lane >> 5);
      float a[T], b[T];
#pragma unroll
      for (int i = 0; i < T; i++) { a[i] = As[buf][wm * 32 * T + i * 32 + (lane & 31)][k]; b[i] = Bs[buf][k][wn * 32 * T + i * 32 + (lane & 31)]; }
#pragma unroll
      for (int i = 0; i < T; i++)
#pragma unroll
        for (int j = 0; j < T; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stage(buf ^ 1);      // (the other buffer was last read before the previous barrier)
    __syncthreads();
  }
  if (!have_c) load_c();      // (K = 0)
#pragma unroll
  for (int i = 0; i < T; i++)
#pragma unroll
    for (int j = 0; j < T; j++) {
      const int col = n0 + wn * 32 * T + j * 32 + (lane & 31);
      if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = m0 + wm * 32 * T + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < M) C[(long long)row * ldc + col] = (seeded ? total[i][j][r] : beta == 0.0f ? 0.0f : beta * cin[i][j][r]) + alpha * acc[i][j][r];
        }
      }
    }
}

__global__ void k3_gemm_splitk_reduce_kernel(const float *W, int S, long long MN, int N, float alpha, float beta, float *C, long long ldc) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i >= MN) return;
  float acc = 0.0f;
  for (int z = 0; z < S; z++) acc += W[(long long)z * MN + i];
  float *c = C + (i / N) * ldc + (i % N);
  *c = (beta != 0.0f ? beta * *c : 0.0f) + alpha * acc;
}

enum {
  kOpSet, kOpScale, kOpFloor, kOpCeil, kOpAddConst, kOpCopyRowsFromVec, kOpMulColsVec, kOpMulRowsVec, kOpAddVecToRows, kOpAddVecToCols, kOpCopy, kOpCopyT,
      kOpAddMat, kOpAddMatT,
       kOpCopyRows, kOpAddRows, kOpMulElements, kOpHeaviside, kOpAddMatDiagVec, kOpAddMatDiagVecT, kOpAddRowRanges, kOpCopyLowerToUpper, kOpAddToDiag,
           kOpAddVecVecOuter, kOpDivElements, kOpAddDiagVecMat, kOpAddDiagVecMatT,
       kOpSigmoid, kOpTanh, kOpDiffSigmoid, kOpDiffTanh, kOpMax, kOpLog, kOpPow, kOpPowAbs, kOpDivRowsVec, kOpCopyCols, kOpAddCols, kOpCopyColsFromVec,
           kOpMulRows, kOpSetMatMatDivMat, kOpAddMatMatElements
       };
struct EwParams {
  int op, rows, cols;
  float *C;
  long long ldc;
  const float *S;
  long long lds;
  const float *v;
  const int *idx;
  float a, b;
  int src_rows;
  const float *S2;
  long long lds2;
  const float *S3;
  long long lds3;
};
// SigmoidComponent / TanhComponent (matrix/kaldi-vector.cc:900-960, the overflow-safe forms of the build without MKL's vector math)
__device__ __forceinline__ float ew_sigmoid(float x) { if (x > 0.0f) return 1.0f / (1.0f + expf(-x)); const float e = expf(x); return e / (e + 1.0f); }
__device__ __forceinline__ float ew_tanh(float x) {
  if (x > 0.0f) {
    const float e = expf(-x);
    return -1.0f + 2.0f / (1.0f + e * e);
  }
  const float e = expf(x);
  return 1.0f - 2.0f / (1.0f + e * e);
}

// the new value of element (r, c); dv = its old value (loaded only for the operations that read it)
__device__ __forceinline__ float ew_elem(const EwParams &p, int r, int c, float dv) {
  float x = 0.0f;
  switch (p.op) {
      case kOpSet: x = p.a; break;
      case kOpScale: x = dv * p.a; break;
      case kOpFloor: x = fmaxf(dv, p.a); break;
      case kOpCeil: x = fminf(dv, p.a); break;
      case kOpAddConst: x = dv + p.a; break;
      case kOpCopyRowsFromVec: x = p.v[c]; break;
      case kOpMulColsVec: x = dv * p.v[c]; break;
      case kOpMulRowsVec: x = dv * p.v[r]; break;
      case kOpAddVecToRows: x = p.a * p.v[c] + p.b * dv; break;                   // cu-matrix.cc AddVecToRows: beta * this + alpha * row
      case kOpAddVecToCols: x = p.a * p.v[r] + p.b * dv; break;
      case kOpCopy: x = p.S[(long long)r * p.lds + c]; break;
      case kOpCopyT: x = p.S[(long long)c * p.lds + r]; break;
      case kOpAddMat: x = dv + p.a * p.S[(long long)r * p.lds + c]; break;
      case kOpAddMatT: x = dv + p.a * p.S[(long long)c * p.lds + r]; break;
      case kOpCopyRows: { const int s = p.idx[r]; x = s < 0 ? 0.0f : p.S[(long long)s * p.lds + c]; break; }          // index -1 = zero row
      case kOpAddRows: { const int s = p.idx[r]; x = s < 0 ? dv : dv + p.a * p.S[(long long)s * p.lds + c]; break; }
      case kOpMulElements: x = dv * p.S[(long long)r * p.lds + c]; break;
      case kOpHeaviside: x = p.S[(long long)r * p.lds + c] > 0.0f ? 1.0f : 0.0f; break;
      case kOpAddMatDiagVec: x = p.b * dv + p.a * p.S[(long long)r * p.lds + c] * p.v[c]; break;               // this = beta this + alpha M diag(v)
      case kOpAddMatDiagVecT: x = p.b * dv + p.a * p.S[(long long)c * p.lds + r] * p.v[c]; break;
      case kOpCopyLowerToUpper: x = c > r ? p.C[(long long)c * p.ldc + r] : dv; break;                        // (reads only below the diagonal, writes only above it)
      case kOpAddToDiag: x = r == c ? dv + p.a : dv; break;
      case kOpAddVecVecOuter: x = dv + p.a * p.v[r] * p.S[c]; break;                                           // this += alpha x y^T
      case kOpDivElements: x = dv / p.S[(long long)r * p.lds + c]; break;
      case kOpAddDiagVecMat: x = p.b * dv + p.a * p.v[r] * p.S[(long long)r * p.lds + c]; break;                 // this = beta this + alpha diag(v) M
      case kOpAddDiagVecMatT: x = p.b * dv + p.a * p.v[r] * p.S[(long long)c * p.lds + r]; break;
      case kOpSigmoid: x = ew_sigmoid(p.S[(long long)r * p.lds + c]); break;
      case kOpTanh: x = ew_tanh(p.S[(long long)r * p.lds + c]); break;
      // kaldi-matrix.cc:3004-3018: diff .* value .* (1 - value)
      case kOpDiffSigmoid: {
        const float y = p.S[(long long)r * p.lds + c];
        x = p.S2[(long long)r * p.lds2 + c] * y * (1.0f - y);
        break;
      }
      case kOpDiffTanh: { const float y = p.S[(long long)r * p.lds + c]; x = p.S2[(long long)r * p.lds2 + c] * (1.0f - y * y); break; }
      case kOpMax: x = fmaxf(dv, p.S[(long long)r * p.lds + c]); break;
      case kOpLog: x = logf(p.S[(long long)r * p.lds + c]); break;
      case kOpPow: x = powf(p.S[(long long)r * p.lds + c], p.a); break;
      // kaldi-matrix.cc:2145-2160
      case kOpPowAbs: {
        const float sv = p.S[(long long)r * p.lds + c], y = powf(fabsf(sv), p.a);
        x = (p.b != 0.0f && sv < 0.0f) ? -y : y;
        break;
      }
      case kOpDivRowsVec: x = dv / p.v[r]; break;
      // kaldi-matrix.cc:2836-2858 (index -1 = zero column)
      case kOpCopyCols: {
        const int s = p.idx[c];
        x = s < 0 ? 0.0f : p.S[(long long)r * p.lds + s];
        break;
      }
      case kOpAddCols: { const int s = p.idx[c]; x = s < 0 ? dv : dv + p.S[(long long)r * p.lds + s]; break; }
      case kOpCopyColsFromVec: x = p.v[r]; break;
      // kaldi-matrix.cc:189-208
      case kOpSetMatMatDivMat: {
        const float i = p.S3[(long long)r * p.lds3 + c], o = p.S2[(long long)r * p.lds2 + c], od = p.S[(long long)r * p.lds + c];
        x = i != 0.0f ? od * (o / i) : od;
        break;
      }
      // kaldi-matrix.cc:636-650
      case kOpAddMatMatElements: x = p.b * dv + p.a * p.S[(long long)r * p.lds + c] * p.S2[(long long)r * p.lds2 + c];
      break;
      // cu-matrix.cc:2813-2843 (a negative index leaves the row alone)
      case kOpMulRows: {
        const int s = p.idx[r];
        x = s < 0 ? dv : dv * p.S[(long long)s * p.lds + c];
        break;
      }
      // cu-kernels.cu _add_row_ranges
      case kOpAddRowRanges: {
        const int b0 = p.idx[2 * r], b1 = p.idx[2 * r + 1];
        x = dv;
        for (int k = b0; k < b1; k++) x += p.S[(long long)k * p.lds + c];
        break;
      }
    }
  return x;
}
// The same operations four columns at a time (dwordx4) for the layouts the network's big matrices have: cols, strides and pointers multiples of 4 floats / 16 bytes, no transposed
// source.  A minibatch's activations are 4 - 14 k rows x 768: ~475 of these per training iteration, each a 30 - 90 MB stream.
__device__ __forceinline__ bool ew_vec4_op(int op) {
  return op == kOpSet || op == kOpScale || op == kOpFloor || op == kOpCeil || op == kOpAddConst || op == kOpCopyRowsFromVec || op == kOpMulColsVec ||
      op == kOpMulRowsVec || op == kOpAddVecToRows ||
         op == kOpAddVecToCols || op == kOpCopy || op == kOpAddMat || op == kOpCopyRows || op == kOpAddRows || op == kOpMulElements || op == kOpHeaviside ||
             op == kOpAddMatDiagVec || op == kOpDivElements ||
         op == kOpAddDiagVecMat || op == kOpSigmoid || op == kOpTanh || op == kOpDiffSigmoid || op == kOpDiffTanh || op == kOpMax || op == kOpMulRows;
}
// element value from the old value d, the source element s_, the vector element v (column- or row-indexed by op)
__device__ __forceinline__ float ew_f(int op, float d, float s_, float v, float a, float b, float s2 = 0.0f) {
  switch (op) {
    case kOpSet: return a;
    case kOpScale: return d * a;
    case kOpFloor: return fmaxf(d, a);
    case kOpCeil: return fminf(d, a);
    case kOpAddConst: return d + a;
    case kOpCopyRowsFromVec: return v;
    case kOpMulColsVec: case kOpMulRowsVec: return d * v;
    case kOpAddVecToRows: case kOpAddVecToCols: return a * v + b * d;
    case kOpCopy: case kOpCopyRows: return s_;
    case kOpAddMat: case kOpAddRows: return d + a * s_;
    case kOpMulElements: return d * s_;
    case kOpHeaviside: return s_ > 0.0f ? 1.0f : 0.0f;
    case kOpAddMatDiagVec: return b * d + a * s_ * v;
    case kOpDivElements: return d / s_;
    case kOpAddDiagVecMat: return b * d + a * v * s_;
    case kOpSigmoid: return ew_sigmoid(s_);
    case kOpTanh: return ew_tanh(s_);
    case kOpDiffSigmoid: return s2 * s_ * (1.0f - s_);
    case kOpDiffTanh: return s2 * (1.0f - s_ * s_);
    case kOpMax: return fmaxf(d, s_);
    case kOpMulRows: return d * s_;
  }
  return d;
}
__global__ __launch_bounds__(256) void k3_ew4_kernel(EwParams p) {      // 256 columns x 8 rows per workgroup and step; two rows per thread
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  if (c >= p.cols) return;
  const int op = p.op;
  const bool unary = op == kOpSigmoid || op == kOpTanh, diff2 = op == kOpDiffSigmoid || op == kOpDiffTanh;
  const bool reads_d = !(op == kOpSet || op == kOpCopyRowsFromVec || op == kOpCopy || op == kOpCopyRows || op == kOpHeaviside || unary || diff2);
  const bool has_s = op == kOpCopy || op == kOpAddMat || op == kOpCopyRows || op == kOpAddRows || op == kOpMulElements || op == kOpHeaviside ||
      op == kOpAddMatDiagVec || op == kOpDivElements || op == kOpAddDiagVecMat || unary || diff2 || op == kOpMax || op == kOpMulRows;
  const bool col_v = op == kOpCopyRowsFromVec || op == kOpMulColsVec || op == kOpAddVecToRows || op == kOpAddMatDiagVec,
      row_v = op == kOpMulRowsVec || op == kOpAddVecToCols || op == kOpAddDiagVecMat;
  const bool indexed = op == kOpCopyRows || op == kOpAddRows || op == kOpMulRows;
  f32x4 vc = {0.0f, 0.0f, 0.0f, 0.0f};
  if (col_v) vc = *reinterpret_cast<const f32x4 *>(p.v + c);
  for (int r0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2; r0 < p.rows; r0 += gridDim.y * 8) {
    f32x4 x[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int r = r0 + e; if (r >= p.rows) break;
      f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f}, sv = d, s2 = d;
      if (diff2) s2 = *reinterpret_cast<const f32x4 *>(p.S2 + (long long)r * p.lds2 + c);
      if (reads_d) d = *reinterpret_cast<const f32x4 *>(p.C + (long long)r * p.ldc + c);
      bool skip = false;
      if (has_s) {
        const int sr = indexed ? p.idx[r] : r;
        if (sr >= 0) sv = *reinterpret_cast<const f32x4 *>(p.S + (long long)sr * p.lds + c);
        else skip = op == kOpAddRows || op == kOpMulRows;
      }
      const float vr = row_v ? p.v[r] : 0.0f;
#pragma unroll
      for (int k = 0; k < 4; k++) x[e][k] = skip ? d[k] : ew_f(op, d[k], sv[k], col_v ? vc[k] : vr, p.a, p.b, s2[k]);
    }
#pragma unroll
    for (int e = 0; e < 2; e++) if (r0 + e < p.rows) *reinterpret_cast<f32x4 *>(p.C + (long long)(r0 + e) * p.ldc + c) = x[e];
  }
}
__global__ __launch_bounds__(256) void k3_ew_kernel(EwParams p) {      // 64 columns x 16 rows per workgroup and step: four rows per thread, their loads issued together
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if (c >= p.cols) return;
  const bool reads_d = !(p.op == kOpSet || p.op == kOpCopyRowsFromVec || p.op == kOpCopy || p.op == kOpCopyT || p.op == kOpCopyRows || p.op == kOpHeaviside ||
      p.op == kOpSigmoid || p.op == kOpTanh ||
                         p.op == kOpDiffSigmoid || p.op == kOpDiffTanh || p.op == kOpLog || p.op == kOpPow || p.op == kOpPowAbs || p.op == kOpCopyCols ||
                             p.op == kOpCopyColsFromVec || p.op == kOpSetMatMatDivMat);
  for (int r0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4; r0 < p.rows; r0 += gridDim.y * 16) {
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; e++) if (r0 + e < p.rows) { const float dv = reads_d ? p.C[(long long)(r0 + e) * p.ldc + c] : 0.0f; x[e] = ew_elem(p, r0 + e, c, dv); }
#pragma unroll
    for (int e = 0; e < 4; e++) if (r0 + e < p.rows) p.C[(long long)(r0 + e) * p.ldc + c] = x[e];
  }
}

// element-wise vector kernels the CuVector side of the adapter needs (model preparation: BatchNormComponent::ComputeDerived, nnet-normalize-component.cc:205-247)
template <typename TS, typename TD> __global__ void k3_vec_convert_kernel(const TS *s, TD *d, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) d[i] = (TD)s[i]; }
__global__ void k3_vec_unary_kernel(int op, float *d, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float x = d[i];
    d[i] = op == 0 ? logf(x) : op == 1 ? expf(x) : 1.0f / x;
  }
}
__global__ void k3_vec_pow_kernel(const float *s, float *d, int n, float power) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) d[i] = powf(s[i], power); }
__global__ void k3_vec_add_vec_vec_kernel(float alpha, const float *a, const float *b, float beta, float *d, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = alpha * a[i] * b[i] + beta * d[i];
}

// float64 vectors: accumulated statistics of a model (BatchNorm sums, NonlinearComponent value / derivative sums) are CuVector<double>
__global__ void k3_vec64_kernel(int op, double alpha, const double *a, const double *b, double beta, double *d, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  switch (op) {
    case 0: d[i] = d[i] * alpha; break;                                  // Scale
    case 1: d[i] = pow(a[i], alpha); break;                              // Pow
    case 2: d[i] = alpha * a[i] + beta * d[i]; break;                    // AddVec
    case 3: d[i] = alpha * a[i] * b[i] + beta * d[i]; break;             // AddVecVec
  }
}


// v[c] = beta v[c] + alpha sum_r f(r, c): AddRowSumMat (f = M), AddDiagMat2 with kTrans (f = M^2), AddDiagMatMat(M, kTrans, N, kNoTrans) (f = M N); op 3 / 4: the same
// over the columns of a row (v[r]: AddDiagMat2 kNoTrans, AddColSumMat).  One wavefront-wide column strip per workgroup row block; partial sums in double.
struct RedParams { int op, rows, cols; const float *M; long long ldm; const float *N; long long ldn; float *v; float alpha, beta; double *part; int rows_per_chunk; };
// blockIdx.y = row chunk: partial sums (double) to p.part[chunk][col], or the result itself when there is one chunk
__global__ __launch_bounds__(256) void k3_colred_kernel(RedParams p) {
  __shared__ double part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const int r0 = blockIdx.y * p.rows_per_chunk, r1 = min(p.rows, r0 + p.rows_per_chunk);
  double acc = 0.0;
  // (op 5: a weight per row -- A^T x)
  if (c < p.cols) for (int r = r0 + w; r < r1; r += 4) {
    const float m = p.M[(long long)r * p.ldm + c];
    acc += p.op == 0 ? (double)m : p.op == 1 ? (double)m * m : p.op == 5 ? (double)m * p.N[(long long)r * p.ldn] : (double)m * p.N[(long long)r * p.ldn + c];
  }
  part[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && c < p.cols) {
    const double s_ = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    if (p.part) p.part[(long long)blockIdx.y * p.cols + c] = s_;
    else p.v[c] = (p.beta == 0.0f ? 0.0f : p.beta * p.v[c]) + p.alpha * (float)s_;
  }
}
// 64 columns per workgroup; wavefront w adds chunks w, w + 4, ... in ascending order, the four sums in wavefront order (deterministic)
__global__ __launch_bounds__(256) void k3_colred_fold_kernel(RedParams p, int chunks) {
  __shared__ double part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  double s_ = 0.0;
  if (c < p.cols) for (int y = w; y < chunks; y += 4) s_ += p.part[(long long)y * p.cols + c];
  part[w][threadIdx.x & 63] = s_;
  __syncthreads();
  if (w == 0 && c < p.cols) {
    s_ = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    p.v[c] = (p.beta == 0.0f ? 0.0f : p.beta * p.v[c]) + p.alpha * (float)s_;
  }
}
__global__ __launch_bounds__(64) void k3_rowred_kernel(RedParams p) {      // one wavefront per row
  const int r = blockIdx.x, lane = threadIdx.x; double acc = 0.0;
  for (int c = lane; c < p.cols; c += 64) { const float m = p.M[(long long)r * p.ldm + c]; acc += p.op == 3 ? (double)m * m : (double)m; }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) p.v[r] = (p.beta == 0.0f ? 0.0f : p.beta * p.v[r]) + p.alpha * (float)acc;
}


// Scalar reductions (TraceMatMat, VecVec, Trace, Sum, Max, Min of cudamatrix/cu-matrix.h, cu-vector.h): per-workgroup partial results in double, folded on the host in
// workgroup order (deterministic).  op 0: sum A(i,j) B(i,j); 1: sum A(i,j) B(j,i); 2: sum A(i,i); 3: sum A(i,j); 4: max; 5: min.
struct ScalParams {
  int op, rows, cols;
  const float *A;
  long long lda;
  const float *B;
  long long ldb;
  double *part;
  unsigned *ticket;
  double *h_out;
  unsigned long long *h_seq;
  unsigned long long seq;
};
__global__ __launch_bounds__(256) void k3_scalar_reduce_kernel(ScalParams p) {
  __shared__ double sh[256];
  const long long n = p.op == 2 ? (long long)p.rows : (long long)p.rows * p.cols;
  double acc = p.op == 4 ? -INFINITY : p.op == 5 ? INFINITY : 0.0;
  if (p.op == 2) { for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += p.A[i * p.lda + i]; }
  else {
    // element i = (r, c) with r, c advanced incrementally (a 64-bit division per element was most of this kernel's time)
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
    int r = (int)(i0 / p.cols), c = (int)(i0 % p.cols); const int dr = (int)(step / p.cols), dc = (int)(step % p.cols);
    for (long long i = i0; i < n; i += step) {
      const float a = p.A[(long long)r * p.lda + c];
      switch (p.op) {
        case 0: acc += (double)a * p.B[(long long)r * p.ldb + c]; break;
        case 1: acc += (double)a * p.B[(long long)c * p.ldb + r]; break;
        case 3: acc += a; break;
        case 4: acc = fmax(acc, (double)a); break;
        case 5: acc = fmin(acc, (double)a); break;
      }
      r += dr; c += dc; if (c >= p.cols) { c -= p.cols; r++; }
    }
  }
  auto fold = [&](double v) {      // the workgroup's 256 values by a fixed tree (deterministic)
    sh[threadIdx.x] = v; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        const double x = sh[threadIdx.x], y = sh[threadIdx.x + o];
        sh[threadIdx.x] = p.op == 4 ? fmax(x, y) : p.op == 5 ? fmin(x, y) : x + y;
      }
      __syncthreads();
    }
    return sh[0];
  };
  const double mine = fold(acc);
  // the last workgroup to finish folds the partial results (thread t: partials t, t + 256, ... in ascending order, then the same tree) and hands the scalar to the host through
  // page-locked memory, followed by this call's sequence number: the host polls that word instead of waiting for the stream (an interrupt-driven wait costs ~20 us per reduction,
  // and natural-gradient training makes ~230 of them per minibatch)
  __shared__ bool last;
  if (threadIdx.x == 0) {
    // No fences: an agent-scope release writes the XCD's L2 back (~20 us per call, measured).  The partial result goes out as a returning agent-scope atomic (complete when it
    // returns), then the tickets -- two levels, groups of 32 workgroups, so that no word takes more than 32 serialised atomics.
    (void)__hip_atomic_exchange(&p.part[blockIdx.x], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned g = blockIdx.x >> 5, ng = (gridDim.x + 31) >> 5, gsize = min(32u, gridDim.x - g * 32u);
    last = __hip_atomic_fetch_add(&p.ticket[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1 &&
        __hip_atomic_fetch_add(&p.ticket[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1;
  }
  __syncthreads();
  if (!last) return;
  double v = p.op == 4 ? -INFINITY : p.op == 5 ? INFINITY : 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {
    const double y = __hip_atomic_load(&p.part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = p.op == 4 ? fmax(v, y) : p.op == 5 ? fmin(v, y) : v + y;
  }
  const double r = fold(v);
  if (threadIdx.x == 0) {
    for (int g = 0; g <= 32; g++) __hip_atomic_store(&p.ticket[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // result, check word (result bits ^ sequence number), sequence number: relaxed system-scope stores; the host accepts the result when all three agree
    const unsigned long long bits = (unsigned long long)__double_as_longlong(r);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p.h_out), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p.h_seq + 1, bits ^ (p.seq * 0x9E3779B97F4A7C15ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p.h_seq, p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// SoftMaxPerRow / LogSoftMaxPerRow and their derivatives (cu-matrix.h:328-334, :403-411): one wavefront per row.  op 0: dst = softmax(src); 1: dst = log-softmax(src);
// 2 (DiffSoftmaxPerRow): dst = diff .* value - value * <diff, value>; 3 (DiffLogSoftmaxPerRow): dst = out_deriv - exp(out_value) * sum(out_deriv).  For op 2 / 3: A = value / out_value, B = diff / out_deriv.
__global__ __launch_bounds__(256) void k3_row_softmax_kernel(int op, float *D, long long ldd, const float *A, long long lda, const float *B, long long ldb, int rows, int cols) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  float *d = D + (long long)r * ldd; const float *a = A + (long long)r * lda; const float *b = B ? B + (long long)r * ldb : nullptr;
  if (op <= 1) {
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, a[c]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.0f;
    for (int c = lane; c < cols; c += 64) sum += expf(a[c] - mx);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float lsum = logf(sum), inv = 1.0f / sum;
    for (int c = lane; c < cols; c += 64) d[c] = op == 1 ? a[c] - mx - lsum : expf(a[c] - mx) * inv;
  } else {
    float sum = 0.0f;
    for (int c = lane; c < cols; c += 64) sum += op == 2 ? a[c] * b[c] : b[c];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    for (int c = lane; c < cols; c += 64) d[c] = op == 2 ? b[c] * a[c] - a[c] * sum : b[c] - expf(a[c]) * sum;
  }
}

// cu::NormalizePerRow / cu::DiffNormalizePerRow (cudamatrix/cu-math.cc:280-318, :349-409; NormalizeComponent, nnet-normalize-component.cc): one wavefront per row.
// forward: y = x * (max(|x|^2 / (D target_rms^2), 2^-66))^-1/2, and with add_log_stddev one more column log(target_rms) - log(that factor).
// backward (the order of the CPU branch): in_deriv (+)= [log-stddev term] + f * out_deriv - (1 / (D target_rms^2)) * <out_deriv, x> * f^3 * x; f^3 := 0 where the floor applied;
// in_deriv aliasing out_deriv (the component's in-place backprop) is overwritten instead of added to.
__global__ __launch_bounds__(256) void k3_row_normalize_kernel(int op, float *D, long long ldd, const float *X, long long ldx, const float *G, long long ldg,
    int rows, int cols, float target_rms, int add_log) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float kFloor = 1.3552527156068805425e-20f;      // 2^-66
  float *d = D + (long long)r * ldd; const float *x = X + (long long)r * ldx; const float *g = G ? G + (long long)r * ldg : nullptr;
  const float d_scaled = (float)cols * target_rms * target_rms;
  float ss = 0.0f, dot = 0.0f;
  for (int c = lane; c < cols; c += 64) { const float v = x[c]; ss += v * v; if (op == 1) dot += g[c] * v; }
  for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); dot += __shfl_xor(dot, o); }
  if (op == 0) {
    const float f = 1.0f / sqrtf(fmaxf(ss * (1.0f / d_scaled), kFloor));
    for (int c = lane; c < cols; c += 64) d[c] = x[c] * f;
    if (add_log && lane == 0) d[cols] = -logf(f) + logf(target_rms);
    return;
  }
  const float scaled = ss * (1.0f / d_scaled), f = 1.0f / sqrtf(fmaxf(scaled, kFloor)), f3 = scaled <= kFloor ? 0.0f : f * f * f;
  const float lsd = add_log ? g[cols] / fmaxf(ss, (float)cols * kFloor) : 0.0f, w = (-1.0f / d_scaled) * (dot * f3);
  const bool in_place = D == G;
  for (int c = lane; c < cols; c += 64) {
    float t;
    if (in_place) t = g[c] * f;
    else { t = d[c]; if (add_log) t += lsd * x[c]; t += f * g[c]; }
    d[c] = t + w * x[c];
  }
}

// CuRand (cudamatrix/cu-rand.h:31-64): counter-based generator (Philox-4x32-10, Salmon et al. 2011: the published round constants and key schedule) -- element
// (r, c) of a call draws
// from counter {offset + (r * cols + c) / 4, stream} under key = seed, so a fill is reproducible for a given (seed, offset) whatever the launch shape or stride, and calls with
// disjoint offset ranges are independent.  kind 0: uniform in [0, 1) with 24 random bits (never 1.0); kind 1: standard normal (Box-Muller over two of the four words).
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int i = 0; i < 10; i++) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ __launch_bounds__(256) void k3_rand_kernel(int kind, float *C, long long ldc, int rows, int cols, unsigned long long seed, unsigned long long offset) {
  const long long n = (long long)rows * cols, q = (long long)blockIdx.x * 256 + threadIdx.x;      // one counter = four consecutive elements (row-major over the logical matrix)
  if (q * 4 >= n) return;
  const unsigned long long ctr = offset + (unsigned long long)q;
  unsigned w[4]; philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
  float v[4];
  if (kind == 0) { for (int e = 0; e < 4; e++) v[e] = (float)(w[e] >> 8) * (1.0f / 16777216.0f); }
  else {
    for (int e = 0; e < 4; e += 2) {
      const float u1 = ((float)(w[e] >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(w[e + 1] >> 8) * (1.0f / 16777216.0f);      // u1 in (0, 1]
      const float rad = sqrtf(-2.0f * logf(u1)), ang = 6.283185307179586f * u2;
      v[e] = rad * cosf(ang); v[e + 1] = rad * sinf(ang);
    }
  }
  for (int e = 0; e < 4; e++) { const long long i = q * 4 + e; if (i < n) C[(i / cols) * ldc + (i % cols)] = v[e]; }
}

int launch_ew(const EwParams &p, void *stream) {
  if (p.rows <= 0 || p.cols <= 0) return K3_OK;
  static const int traced = [] { const char *e = getenv("K3_GEMM_TRACE"); return e && atoi(e) >= 2 ? 1 : 0; }();
  if (traced) fprintf(stderr, "k3 ew op %d rows %d cols %d\n", p.op, p.rows, p.cols);      // developer aid
  auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool has_s = p.op == kOpCopy || p.op == kOpAddMat || p.op == kOpCopyRows || p.op == kOpAddRows || p.op == kOpMulElements || p.op == kOpHeaviside ||
      p.op == kOpAddMatDiagVec || p.op == kOpDivElements || p.op == kOpAddDiagVecMat ||
                     p.op == kOpSigmoid || p.op == kOpTanh || p.op == kOpDiffSigmoid || p.op == kOpDiffTanh || p.op == kOpMax || p.op == kOpMulRows;
  const bool has_s2 = p.op == kOpDiffSigmoid || p.op == kOpDiffTanh;
  const bool col_v = p.op == kOpCopyRowsFromVec || p.op == kOpMulColsVec || p.op == kOpAddVecToRows || p.op == kOpAddMatDiagVec;
  const bool v4 = (p.op == kOpSet || p.op == kOpScale || p.op == kOpFloor || p.op == kOpCeil || p.op == kOpAddConst || p.op == kOpMulRowsVec ||
      p.op == kOpAddVecToCols || has_s || col_v) &&
                  p.cols % 4 == 0 && p.cols >= 64 && p.ldc % 4 == 0 && al16(p.C) && (!has_s || (p.lds % 4 == 0 && al16(p.S))) &&
                      (!has_s2 || (p.lds2 % 4 == 0 && al16(p.S2))) && (!col_v || al16(p.v)) && !getenv("K3_EW_SCALAR");
  if (v4) {
    hipLaunchKernelGGL(k3_ew4_kernel, dim3((p.cols + 255) / 256, (unsigned)std::min(65535, (p.rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, p);
    K3_HIP_CHECK(hipGetLastError());
    return K3_OK;
  }
  dim3 grid((p.cols + 63) / 64, (unsigned)std::min(65535, (p.rows + 15) / 16));
  hipLaunchKernelGGL(k3_ew_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
EwParams mk(int op, float *C, long long ldc, int rows, int cols) { EwParams p{}; p.op = op; p.C = C; p.ldc = ldc; p.rows = rows; p.cols = cols; return p; }

}  // namespace

#define K3_MAT_REQUIRE(C, ldc, rows, cols) K3_REQUIRE((C) && (rows) >= 0 && (cols) >= 0 && (ldc) >= (cols), "k3_mat: bad matrix argument")

// scratch that a two-kernel operation hands from its first kernel to its second (split-K planes, column-reduction partials): one buffer per (device, stream), so that
// operations queued on different streams never share one; it only grows (after a device synchronise: an earlier operation of the stream may still read the old one).
// `hold` keeps the table's lock until the caller has queued BOTH kernels of its operation: two host threads that issue such operations on the same stream (the null stream, say)
// must not interleave as producer A, producer B, consumer A -- consumer A would read B's planes -- and the buffer must not be regrown under a queued-but-unlaunched pair.
namespace {
int workspace(hipStream_t st, size_t bytes, void **out, std::unique_lock<std::mutex> &hold) {
  struct Ws { void *p = nullptr; size_t cap = 0; };
  static std::mutex mu; static std::map<std::pair<int, hipStream_t>, Ws> tab;
  int dev = 0; K3_HIP_CHECK(hipGetDevice(&dev));
  hold = std::unique_lock<std::mutex>(mu);
  Ws &w = tab[{dev, st}];
  if (bytes > w.cap) {
    if (w.p) { K3_HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(w.p); w.p = nullptr; w.cap = 0; }
    const size_t cap = std::max(bytes + bytes / 2, (size_t)1 << 20);
    K3_HIP_CHECK(hipMalloc(&w.p, cap)); w.cap = cap;
  }
  *out = w.p; return K3_OK;
}
int col_reduce(RedParams p, hipStream_t st) {      // ops 0 - 2 and 5 of k3_colred_kernel
  if (p.cols <= 0) return K3_OK;
  // rows split into chunks so that the launch fills the chip (a [9152 x 768] sum was 12 workgroups walking 9152 rows each); partials in double, folded in chunk order
  const int xb = (p.cols + 63) / 64; int chunks = std::max(1, std::min(64, std::min((p.rows + 63) / 64, (1024 + xb - 1) / xb)));
  if (chunks > 1) {
    p.rows_per_chunk = (p.rows + chunks - 1) / chunks; chunks = (p.rows + p.rows_per_chunk - 1) / p.rows_per_chunk;
    // (held until both kernels are queued)
    std::unique_lock<std::mutex> hold;
    void *wsv = nullptr;
    {
      const int rc = workspace(st, (size_t)chunks * p.cols * sizeof(double), &wsv, hold);
      if (rc) return rc;
    }
    p.part = static_cast<double *>(wsv);
    hipLaunchKernelGGL(k3_colred_kernel, dim3(xb, chunks), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k3_colred_fold_kernel, dim3(xb), dim3(256), 0, st, p, chunks);
  } else { p.rows_per_chunk = p.rows; hipLaunchKernelGGL(k3_colred_kernel, dim3(xb, 1), dim3(256), 0, st, p); }
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
std::atomic<long long> g_gemm_flops{0};
template <int T, int BK> void launch_tile(int ta, int tb, dim3 grid, hipStream_t st, int M, int N, int K, float alpha, const float *A, long long lda,
    const float *B, long long ldb, float beta, float *C, long long ldc, float *W, int Kc, int kblk) {
  if (ta) {
    if (tb) hipLaunchKernelGGL((k3_gemm_tile_kernel<T, 1, 1, BK>), grid, dim3(256), 0, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, W, Kc, kblk);
    else hipLaunchKernelGGL((k3_gemm_tile_kernel<T, 1, 0, BK>), grid, dim3(256), 0, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, W, Kc, kblk);
  }
  else {
    if (tb) hipLaunchKernelGGL((k3_gemm_tile_kernel<T, 0, 1, BK>), grid, dim3(256), 0, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, W, Kc, kblk);
    else hipLaunchKernelGGL((k3_gemm_tile_kernel<T, 0, 0, BK>), grid, dim3(256), 0, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, W, Kc, kblk);
  }
}
// T: 0 = the generic kernel (any alignment), 1 / 2 = the 64 / 128 tile kernel
void launch_gemm(int ta, int tb, int T, dim3 grid, hipStream_t st, int M, int N, int K, float alpha, const float *A, long long lda, const float *B,
    long long ldb, float beta, float *C, long long ldc, float *W, int Kc, int kblk) {
  if (T == 0) { hipLaunchKernelGGL(k3_gemm_generic_kernel, grid, dim3(256), 0, st, M, N, K, alpha, A, lda, ta, B, ldb, tb, beta, C, ldc, W, Kc); return; }
  static const int bk = [] { const char *e = getenv("K3_GEMM_BK"); return e ? atoi(e) : 32; }();      // (developer aid: the 64-tile kernel's k-tile)
  if (T == 2) launch_tile<2, 16>(ta, tb, grid, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, W, Kc, kblk);
  else if (bk == 16) launch_tile<1, 16>(ta, tb, grid, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, W, Kc, kblk);
  else if (bk == 64) launch_tile<1, 64>(ta, tb, grid, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, W, Kc, kblk);
  else launch_tile<1, 32>(ta, tb, grid, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, W, Kc, kblk);
}
}  // namespace

// 2 M N K summed over the k3_mat_add_mat_mat calls of this process
extern "C" int64_t k3_mat_gemm_flops(int32_t reset) {
  return reset ? g_gemm_flops.exchange(0) : g_gemm_flops.load();
}

static int add_mat_mat(float alpha, const float *d_A, int64_t lda, int32_t trans_a, const float *d_B, int64_t ldb, int32_t trans_b, float beta, float *d_C,
    int64_t ldc, int32_t M, int32_t N, int32_t K, void *stream);
extern "C" int k3_mat_add_mat_mat(float alpha, const float *d_A, int64_t lda, int32_t trans_a, const float *d_B, int64_t ldb, int32_t trans_b, float beta,
                                  float *d_C, int64_t ldc, int32_t M, int32_t N, int32_t K, void *stream) {
  static const int timed = [] { const char *e = getenv("K3_GEMM_TRACE"); return e && atoi(e) >= 2 ? 1 : 0; }();
  if (!timed) return add_mat_mat(alpha, d_A, lda, trans_a, d_B, ldb, trans_b, beta, d_C, ldc, M, N, K, stream);
  // developer aid (K3_GEMM_TRACE=2): every product alone on the device, its shape and its time
  (void)hipDeviceSynchronize(); const auto t0 = std::chrono::steady_clock::now();
  const int rc = add_mat_mat(alpha, d_A, lda, trans_a, d_B, ldb, trans_b, beta, d_C, ldc, M, N, K, stream);
  (void)hipDeviceSynchronize(); const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  fprintf(stderr, "k3 gemm M %d N %d K %d ta %d tb %d lda %lld ldb %lld beta %g us %.1f\n", M, N, K, trans_a ? 1 : 0, trans_b ? 1 : 0, (long long)lda,
      (long long)ldb, (double)beta, us);
  return rc;
}
static int add_mat_mat(float alpha, const float *d_A, int64_t lda, int32_t trans_a, const float *d_B, int64_t ldb, int32_t trans_b, float beta, float *d_C,
    int64_t ldc, int32_t M, int32_t N, int32_t K, void *stream) {
  K3_REQUIRE(d_A && d_B && d_C && M >= 0 && N >= 0 && K >= 0 && ldc >= N, "k3_mat_add_mat_mat: bad argument");
  K3_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N), "k3_mat_add_mat_mat: leading dimension smaller than the row length");
  if (M == 0 || N == 0) return K3_OK;
  g_gemm_flops += 2ll * M * N * K;
  const int ta = trans_a ? 1 : 0, tb = trans_b ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  // a product with one output column over a transposed A (the bias gradient y = A^T x of a minibatch) is a weighted column sum, not a tile problem
  if (N == 1 && ta && ldc == 1 && M >= 64 &&
      !getenv("K3_GEMM_GENERIC")) return col_reduce(RedParams{5, K, M, d_A, (long long)lda, d_B, tb ? 1ll : (long long)ldb, d_C, alpha, beta, nullptr, K}, st);
  // the tile kernels want dwordx4 loads: aligned operands.  Everything else (and K < 16) stays on the generic kernel.
  const bool aligned = ((reinterpret_cast<uintptr_t>(d_A) | reinterpret_cast<uintptr_t>(d_B)) & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0 && K >= 16 && !getenv("K3_GEMM_GENERIC");
  const long long tiles128 = (long long)((N + 127) / 128) * ((M + 127) / 128), tiles64 = (long long)((N + 63) / 64) * ((M + 63) / 64);
  // 128-tiles when they fill the chip by themselves (or with the coarse split of a long K: a layer's weight gradient), else 64-tiles: four times the workgroups, and K split
  // in steps of 64 until there are two workgroups per CU (a minibatch of 64 sequences is 4 - 14 k rows: [4736 x 96 x 768] is 37 big tiles)
  // Products that accumulate into C with A as it lies (beta != 0: every forward product and input derivative of the network) keep the reference-like order: blocks of 384,
  // K split only there.  The others (beta == 0 or A transposed: weight gradients, the preconditioner's factors) split K in steps of 64 until the chip is full.
  const bool fine = beta == 0.0f || ta;
  const int T = !aligned ? 0 : (tiles128 >= 224 || (tiles128 >= 24 && K >= 3072)) ? 2 : 1, TS = T == 2 ? 128 : 64, kblk = (T == 1 && fine) ? 64 : 384;
  // developer aid: which products miss the tile kernels
  if (T == 0 && getenv("K3_GEMM_TRACE")) fprintf(stderr, "k3 generic gemm M %d N %d K %d ta %d tb %d lda %lld ldb %lld A&15 %d B&15 %d\n", M, N, K, ta, tb,
      (long long)lda, (long long)ldb, (int)(reinterpret_cast<uintptr_t>(d_A) & 15), (int)(reinterpret_cast<uintptr_t>(d_B) & 15));
  const long long tiles = T == 2 ? tiles128 : tiles64;
  const dim3 grid((N + TS - 1) / TS, (M + TS - 1) / TS);
  int S = 1, Kc = 0;
  if (T == 2) { if (tiles < 192 && K >= 768) { S = (int)std::min<long long>(64, std::max<long long>(2, 512 / tiles)); Kc = ((K + S - 1) / S + 383) / 384 * 384; } }
  else if (T == 1) { if (tiles < 384 && K >= 2 * kblk) { S = (int)std::min<long long>(64, (512 + tiles - 1) / tiles); Kc = ((K + S - 1) / S + kblk - 1) / kblk * kblk; } }
  else if (tiles < 384 && K >= (fine ? 512 : 3072)) {
    S = (int)std::min<long long>(16, std::max<long long>(2, 1024 / tiles));
    Kc = fine ? ((K + S - 1) / S + 127) / 128 * 128 : ((K + S - 1) / S + 383) / 384 * 384;
  }
  if (S > 1) S = (K + Kc - 1) / Kc;
  if (S > 1) {      // planes are whole accumulation blocks, reduced in ascending order: deterministic, and for the tile kernels the same sums as without the split
    // (held until both kernels are queued)
    std::unique_lock<std::mutex> hold;
    void *wsv = nullptr;
    {
      const int rc = workspace(st, (size_t)S * M * N * sizeof(float), &wsv, hold);
      if (rc) return rc;
    }
    float *ws = static_cast<float *>(wsv);
    launch_gemm(ta, tb, T, dim3(grid.x, grid.y, S), st, M, N, K, alpha, d_A, (long long)lda, d_B, (long long)ldb, beta, d_C, (long long)ldc, ws, Kc, kblk);
    const long long MN = (long long)M * N;
    hipLaunchKernelGGL(k3_gemm_splitk_reduce_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, st, ws, S, MN, N, alpha, beta, d_C, ldc);
    K3_HIP_CHECK(hipGetLastError());
    return K3_OK;
  }
  launch_gemm(ta, tb, T, grid, st, M, N, K, alpha, d_A, (long long)lda, d_B, (long long)ldb, beta, d_C, (long long)ldc, (float *)nullptr, 0, kblk);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
extern "C" int k3_mat_set(float *C, int64_t ldc, int32_t rows, int32_t cols, float value, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  EwParams p = mk(kOpSet, C, ldc, rows, cols);
  p.a = value;
  return launch_ew(p, st);
}
extern "C" int k3_mat_scale(float *C, int64_t ldc, int32_t rows, int32_t cols, float value, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  EwParams p = mk(kOpScale, C, ldc, rows, cols);
  p.a = value;
  return launch_ew(p, st);
}
extern "C" int k3_mat_add(float *C, int64_t ldc, int32_t rows, int32_t cols, float value, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  EwParams p = mk(kOpAddConst, C, ldc, rows, cols);
  p.a = value;
  return launch_ew(p, st);
}
extern "C" int k3_mat_apply_floor(float *C, int64_t ldc, int32_t rows, int32_t cols, float floor_val, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  EwParams p = mk(kOpFloor, C, ldc, rows, cols);
  p.a = floor_val;
  return launch_ew(p, st);
}
extern "C" int k3_mat_apply_ceiling(float *C, int64_t ldc, int32_t rows, int32_t cols, float ceil_val, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  EwParams p = mk(kOpCeil, C, ldc, rows, cols);
  p.a = ceil_val;
  return launch_ew(p, st);
}
extern "C" int k3_mat_copy_rows_from_vec(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_v, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_v, "k3_mat_copy_rows_from_vec: null vector");
  EwParams p = mk(kOpCopyRowsFromVec, C, ldc, rows, cols);
  p.v = d_v;
  return launch_ew(p, st);
}
extern "C" int k3_mat_mul_cols_vec(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_scale, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_scale, "k3_mat_mul_cols_vec: null vector");
  EwParams p = mk(kOpMulColsVec, C, ldc, rows, cols);
  p.v = d_scale;
  return launch_ew(p, st);
}
extern "C" int k3_mat_mul_rows_vec(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_scale, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_scale, "k3_mat_mul_rows_vec: null vector");
  EwParams p = mk(kOpMulRowsVec, C, ldc, rows, cols);
  p.v = d_scale;
  return launch_ew(p, st);
}
extern "C" int k3_mat_add_vec_to_rows(float alpha, const float *d_row, float beta, float *C, int64_t ldc, int32_t rows, int32_t cols, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_row, "k3_mat_add_vec_to_rows: null vector");
  EwParams p = mk(kOpAddVecToRows, C, ldc, rows, cols);
  p.v = d_row;
  p.a = alpha;
  p.b = beta;
  return launch_ew(p, st);
}
extern "C" int k3_mat_add_vec_to_cols(float alpha, const float *d_col, float beta, float *C, int64_t ldc, int32_t rows, int32_t cols, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_col, "k3_mat_add_vec_to_cols: null vector");
  EwParams p = mk(kOpAddVecToCols, C, ldc, rows, cols);
  p.v = d_col;
  p.a = alpha;
  p.b = beta;
  return launch_ew(p, st);
}
extern "C" int k3_mat_copy_from_mat(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, int32_t trans, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_src && lds >= (trans ? rows : cols), "k3_mat_copy_from_mat: bad source");
  EwParams p = mk(trans ? kOpCopyT : kOpCopy, C, ldc, rows, cols);
  p.S = d_src;
  p.lds = lds;
  return launch_ew(p, st);
}
extern "C" int k3_mat_add_mat(float alpha, const float *d_A, int64_t lda, int32_t trans_a, float *C, int64_t ldc, int32_t rows, int32_t cols, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_A && lda >= (trans_a ? rows : cols), "k3_mat_add_mat: bad source");
  EwParams p = mk(trans_a ? kOpAddMatT : kOpAddMat, C, ldc, rows, cols);
  p.S = d_A;
  p.lds = lda;
  p.a = alpha;
  return launch_ew(p, st);
}
extern "C" int k3_mat_copy_rows(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, const int32_t *d_indexes, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_src && d_indexes && lds >= cols, "k3_mat_copy_rows: bad source");
  EwParams p = mk(kOpCopyRows, C, ldc, rows, cols);
  p.S = d_src;
  p.lds = lds;
  p.idx = d_indexes;
  return launch_ew(p, st);
}
extern "C" int k3_mat_add_rows(float alpha, const float *d_src, int64_t lds, const int32_t *d_indexes, float *C, int64_t ldc, int32_t rows, int32_t cols,
    void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_src && d_indexes && lds >= cols, "k3_mat_add_rows: bad source");
  EwParams p = mk(kOpAddRows, C, ldc, rows, cols);
  p.S = d_src;
  p.lds = lds;
  p.idx = d_indexes;
  p.a = alpha;
  return launch_ew(p, st);
}

// ---- vectors (CuVectorBase): everything else a vector needs is the matrix entry points on a [1 x dim] matrix
extern "C" int k3_mat_mul_elements(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_A, int64_t lda, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_A && lda >= cols, "k3_mat_mul_elements: bad source");
  EwParams p = mk(kOpMulElements, C, ldc, rows, cols);
  p.S = d_A;
  p.lds = lda;
  return launch_ew(p, st);
}
extern "C" int k3_mat_heaviside(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_src && lds >= cols, "k3_mat_heaviside: bad source");
  EwParams p = mk(kOpHeaviside, C, ldc, rows, cols);
  p.S = d_src;
  p.lds = lds;
  return launch_ew(p, st);
}
extern "C" int k3_mat_add_mat_diag_vec(float alpha, const float *d_M, int64_t ldm, int32_t trans_m, const float *d_v, float beta, float *C, int64_t ldc,
    int32_t rows, int32_t cols, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols); K3_REQUIRE(d_M && d_v && ldm >= (trans_m ? rows : cols), "k3_mat_add_mat_diag_vec: bad source");
  EwParams p = mk(trans_m ? kOpAddMatDiagVecT : kOpAddMatDiagVec, C, ldc, rows, cols); p.S = d_M; p.lds = ldm; p.v = d_v; p.a = alpha; p.b = beta; return launch_ew(p, st);
}
extern "C" int k3_mat_add_row_ranges(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, int32_t src_rows, const int32_t *d_ranges, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols); K3_REQUIRE(d_src && d_ranges && lds >= cols && src_rows >= 0, "k3_mat_add_row_ranges: bad source");
  EwParams p = mk(kOpAddRowRanges, C, ldc, rows, cols); p.S = d_src; p.lds = lds; p.idx = d_ranges; p.src_rows = src_rows; return launch_ew(p, st);
}
extern "C" int k3_mat_copy_lower_to_upper(float *C, int64_t ldc, int32_t n, void *st) {                                  // CuMatrixBase::CopyLowerToUpper
  K3_MAT_REQUIRE(C, ldc, n, n); return launch_ew(mk(kOpCopyLowerToUpper, C, ldc, n, n), st);
}
extern "C" int k3_mat_add_to_diag(float *C, int64_t ldc, int32_t rows, int32_t cols, float value, void *st) {                // CuMatrixBase::AddToDiag
  K3_MAT_REQUIRE(C, ldc, rows, cols); EwParams p = mk(kOpAddToDiag, C, ldc, rows, cols); p.a = value; return launch_ew(p, st);
}
// CuMatrixBase::AddVecVec: this += alpha x y^T
extern "C" int k3_mat_add_vec_vec(float alpha, const float *d_x, const float *d_y, float *C, int64_t ldc, int32_t rows, int32_t cols, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols); K3_REQUIRE(d_x && d_y, "k3_mat_add_vec_vec: null vector");
  EwParams p = mk(kOpAddVecVecOuter, C, ldc, rows, cols); p.a = alpha; p.v = d_x; p.S = d_y; return launch_ew(p, st);
}
extern "C" int k3_mat_div_elements(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_A, int64_t lda, void *st) {                       // CuMatrixBase::DivElements
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_A && lda >= cols, "k3_mat_div_elements: bad source");
  EwParams p = mk(kOpDivElements, C, ldc, rows, cols);
  p.S = d_A;
  p.lds = lda;
  return launch_ew(p, st);
}
// CuMatrixBase::AddDiagVecMat
extern "C" int k3_mat_add_diag_vec_mat(float alpha, const float *d_v, const float *d_M, int64_t ldm, int32_t trans_m, float beta, float *C, int64_t ldc,
    int32_t rows, int32_t cols, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols); K3_REQUIRE(d_v && d_M && ldm >= (trans_m ? rows : cols), "k3_mat_add_diag_vec_mat: bad source");
  EwParams p = mk(trans_m ? kOpAddDiagVecMatT : kOpAddDiagVecMat, C, ldc, rows, cols); p.a = alpha; p.b = beta; p.v = d_v; p.S = d_M; p.lds = ldm; return launch_ew(p, st);
}
extern "C" int k3_mat_reduce_scalar(int32_t op, const float *d_A, int64_t lda, const float *d_B, int64_t ldb, int32_t rows, int32_t cols, double *h_result, void *st) {
  K3_REQUIRE(h_result && op >= 0 && op <= 5 && rows >= 0 && cols >= 0, "k3_mat_reduce_scalar: bad argument");
  *h_result = op == 4 ? -INFINITY : op == 5 ? INFINITY : 0.0;
  if (rows == 0 || cols == 0) return K3_OK;
  K3_REQUIRE(d_A && lda >= cols && (op > 1 || (d_B && ldb >= (op == 0 ? cols : rows))) && (op != 2 || rows == cols), "k3_mat_reduce_scalar: bad matrix");
  constexpr int kMaxWgs = 1024;
  // per host thread (the call does not return before its result is there): partial results + the ticket on the device, result + sequence word in page-locked host memory
  struct Scratch { double *d_part = nullptr; unsigned *d_ticket = nullptr; double *h = nullptr; int dev = -1; unsigned long long seq = 0; };
  static thread_local Scratch sc;
  { int dev = 0; K3_HIP_CHECK(hipGetDevice(&dev));
    if (sc.dev != dev) {      // (a thread that moves to another device leaves a few KB behind on the old one)
      K3_HIP_CHECK(hipMalloc((void **)&sc.d_part, kMaxWgs * sizeof(double) + 256));
      sc.d_ticket = reinterpret_cast<unsigned *>(sc.d_part + kMaxWgs);
      K3_HIP_CHECK(hipMemset(sc.d_ticket, 0, 256));
      K3_HIP_CHECK(hipHostMalloc((void **)&sc.h, 128, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent)); memset(sc.h, 0, 128); sc.dev = dev; sc.seq = 0;
    } }
  const long long n = op == 2 ? (long long)rows : (long long)rows * cols;
  const int wgs = (int)std::min<long long>(kMaxWgs, (n + 255) / 256);
  unsigned long long *h_seq = reinterpret_cast<unsigned long long *>(sc.h + 8); const unsigned long long seq = ++sc.seq;
  ScalParams p{op, rows, cols, d_A, lda, d_B, ldb, sc.d_part, sc.d_ticket, sc.h, h_seq, seq};
  hipLaunchKernelGGL(k3_scalar_reduce_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)st, p);
  K3_HIP_CHECK(hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long long bits = 0;
  auto arrived = [&]() {
    if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != seq) return false;
    bits = __atomic_load_n(reinterpret_cast<unsigned long long *>(sc.h), __ATOMIC_ACQUIRE);
    return __atomic_load_n(h_seq + 1, __ATOMIC_ACQUIRE) == (bits ^ (seq * 0x9E3779B97F4A7C15ull));
  };
  for (unsigned spins = 1; !arrived(); spins++) {
    if ((spins & 4095) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {      // a long queue in front of the kernel: wait the ordinary way
      K3_HIP_CHECK(hipStreamSynchronize((hipStream_t)st));
      K3_REQUIRE(arrived(), "k3_mat_reduce_scalar: the reduction kernel finished without delivering its result");
      break;
    }
  }
  memcpy(h_result, &bits, sizeof(double));
  return K3_OK;
}
extern "C" int k3_mat_softmax_rows(int32_t op, float *d_dst, int64_t ldd, const float *d_a, int64_t lda, const float *d_b, int64_t ldb, int32_t rows, int32_t cols, void *st) {
  K3_REQUIRE(d_dst && d_a && op >= 0 && op <= 3 && rows >= 0 && cols >= 0 && ldd >= cols && lda >= cols && (op <= 1 || (d_b && ldb >= cols)), "k3_mat_softmax_rows: bad argument");
  if (rows == 0 || cols == 0) return K3_OK;
  hipLaunchKernelGGL(k3_row_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)st, op, d_dst, (long long)ldd, d_a, (long long)lda,
      d_b, (long long)ldb, rows, cols);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
extern "C" int k3_mat_normalize_rows(int32_t op, float *d_dst, int64_t ldd, const float *d_in, int64_t ldi, const float *d_out_deriv, int64_t ldo,
    int32_t rows, int32_t cols, float target_rms, int32_t add_log_stddev, void *st) {
  K3_REQUIRE(d_dst && d_in && (op == 0 || op == 1) && rows >= 0 && cols >= 0 && ldi >= cols && ldd >= cols + (op == 0 && add_log_stddev ? 1 : 0) && target_rms > 0.0f &&
             (op == 0 || (d_out_deriv && ldo >= cols + (add_log_stddev ? 1 : 0))), "k3_mat_normalize_rows: bad argument");
  // in-place backward (in_deriv aliasing out_deriv) exists for the plain form only: NormalizeComponent advertises kBackpropInPlace only without add_log_stddev
  // (nnet3/nnet-normalize-component.h Properties()), and the reference's CPU branch for the combination (cu-math.cc DiffNormalizePerRow: the log-stddev term is added
  // into the aliased matrix BEFORE the products that read it) is not what the kernel's in-place branch computes
  K3_REQUIRE(!(op == 1 && add_log_stddev && d_dst == d_out_deriv), "k3_mat_normalize_rows: the in-place backward pass is not defined with add_log_stddev");
  if (rows == 0 || cols == 0) return K3_OK;
  hipLaunchKernelGGL(k3_row_normalize_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)st, op, d_dst, (long long)ldd, d_in, (long long)ldi,
      d_out_deriv, (long long)ldo, rows, cols, target_rms, add_log_stddev ? 1 : 0);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
// element maps of the remaining nonlinearities (CuMatrixBase::Sigmoid / Tanh / Log / Pow / PowAbs / Max, cu-matrix.h): op 0 sigmoid(src), 1 tanh(src), 2 log(src), 3 pow(src, a),
// 4 pow(|src|, a) (flag: with src's sign), 5 max(dst, src)
extern "C" int k3_mat_apply_map(int32_t op, float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, float a, int32_t flag, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols); K3_REQUIRE(d_src && lds >= cols && op >= 0 && op <= 5, "k3_mat_apply_map: bad argument");
  static const int ops[6] = {kOpSigmoid, kOpTanh, kOpLog, kOpPow, kOpPowAbs, kOpMax};
  EwParams p = mk(ops[op], C, ldc, rows, cols); p.S = d_src; p.lds = lds; p.a = a; p.b = flag ? 1.0f : 0.0f; return launch_ew(p, st);
}
// CuMatrixBase::DiffSigmoid / DiffTanh (cu-matrix.h:390-396): dst = diff .* value .* (1 - value) (op 0) or diff .* (1 - value^2) (op 1); dst may be diff or value
extern "C" int k3_mat_diff_activation(int32_t op, float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_value, int64_t ldv, const float *d_diff,
    int64_t ldf, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols); K3_REQUIRE(d_value && d_diff && ldv >= cols && ldf >= cols && (op == 0 || op == 1), "k3_mat_diff_activation: bad argument");
  EwParams p = mk(op == 0 ? kOpDiffSigmoid : kOpDiffTanh, C, ldc, rows, cols); p.S = d_value; p.lds = ldv; p.S2 = d_diff; p.lds2 = ldf; return launch_ew(p, st);
}
// CuMatrixBase::MulRows (cu-matrix.h: row r *= src row indexes[r]; -1 = unchanged)
extern "C" int k3_mat_mul_rows(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, const int32_t *d_indexes, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols); K3_REQUIRE(d_src && d_indexes && lds >= cols, "k3_mat_mul_rows: bad source");
  EwParams p = mk(kOpMulRows, C, ldc, rows, cols); p.S = d_src; p.lds = lds; p.idx = d_indexes; return launch_ew(p, st);
}
// CuMatrixBase::SetMatMatDivMat (op 0: dst = A .* (B ./ C), A where C is 0 -- DropoutComponent::Backprop) / AddMatMatElements (op 1: dst = beta dst + alpha A
// .* B) (cu-matrix.h:580,:608)
extern "C" int k3_mat_elements3(int32_t op, float *C, int64_t ldc, int32_t rows, int32_t cols, float alpha, const float *d_A, int64_t lda, const float *d_B,
    int64_t ldb, const float *d_C3, int64_t ldc3, float beta, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols); K3_REQUIRE(d_A && d_B && lda >= cols && ldb >= cols && (op == 1 || (op == 0 && d_C3 && ldc3 >= cols)), "k3_mat_elements3: bad argument");
  EwParams p = mk(op == 0 ? kOpSetMatMatDivMat : kOpAddMatMatElements, C, ldc, rows, cols);
  p.S = d_A;
  p.lds = lda;
  p.S2 = d_B;
  p.lds2 = ldb;
  p.S3 = d_C3;
  p.lds3 = ldc3;
  p.a = alpha;
  p.b = beta;
  return launch_ew(p, st);
}
extern "C" int k3_mat_div_rows_vec(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_div, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_div, "k3_mat_div_rows_vec: null vector");
  EwParams p = mk(kOpDivRowsVec, C, ldc, rows, cols);
  p.v = d_div;
  return launch_ew(p, st);
}
extern "C" int k3_mat_copy_cols_from_vec(float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_col, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols);
  K3_REQUIRE(d_col, "k3_mat_copy_cols_from_vec: null vector");
  EwParams p = mk(kOpCopyColsFromVec, C, ldc, rows, cols);
  p.v = d_col;
  return launch_ew(p, st);
}
// CuMatrixBase::CopyCols / AddCols (cu-matrix.h:102-111): dst(r, c) (+)= src(r, indexes[c]); index -1 = zero / nothing added
extern "C" int k3_mat_copy_cols(int32_t add, float *C, int64_t ldc, int32_t rows, int32_t cols, const float *d_src, int64_t lds, const int32_t *d_indexes, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols); K3_REQUIRE(d_src && d_indexes && lds > 0, "k3_mat_copy_cols: bad source");
  EwParams p = mk(add ? kOpAddCols : kOpCopyCols, C, ldc, rows, cols); p.S = d_src; p.lds = lds; p.idx = d_indexes; return launch_ew(p, st);
}
// CuRand<float>::RandUniform (kind 0) / RandGaussian (kind 1) (cudamatrix/cu-rand.h:50-56).  The caller owns the stream position: `offset` counts groups of four elements, a fill
// of rows x cols consumes ceil(rows * cols / 4) of them.
extern "C" int k3_mat_set_rand(int32_t kind, float *C, int64_t ldc, int32_t rows, int32_t cols, uint64_t seed, uint64_t offset, void *st) {
  K3_MAT_REQUIRE(C, ldc, rows, cols); K3_REQUIRE(kind == 0 || kind == 1, "k3_mat_set_rand: kind must be 0 (uniform) or 1 (gaussian)");
  const long long n = (long long)rows * cols; if (n == 0) return K3_OK;
  hipLaunchKernelGGL(k3_rand_kernel, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)st, kind, C, (long long)ldc, rows, cols,
      (unsigned long long)seed, (unsigned long long)offset);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
extern "C" int k3_vec_col_reduce(int32_t op, float alpha, const float *d_M, int64_t ldm, const float *d_N, int64_t ldn, int32_t rows, int32_t cols, float beta,
    float *d_v, void *st) {
  K3_REQUIRE(d_M && d_v && rows >= 0 && cols >= 0 && ldm >= cols && op >= 0 && op <= 4 && (op != 2 || (d_N && ldn >= cols)), "k3_vec_col_reduce: bad argument");
  RedParams p{op, rows, cols, d_M, ldm, d_N, ldn, d_v, alpha, beta, nullptr, rows};
  if (op <= 2) return col_reduce(p, (hipStream_t)st);
  if (rows > 0) hipLaunchKernelGGL(k3_rowred_kernel, dim3(rows), dim3(64), 0, (hipStream_t)st, p);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}

// CuVectorBase<Real>::CopyFromVec(const CuVectorBase<OtherReal>&)
extern "C" int k3_vec_convert(const void *d_src, int32_t src_is_f64, void *d_dst, int32_t dst_is_f64, int32_t n, void *st) {
  if (n == 0) return K3_OK;
  K3_REQUIRE(d_src && d_dst && n > 0, "k3_vec_convert: bad argument");
  const dim3 g((n + 255) / 256), b(256);
  if (src_is_f64 && !dst_is_f64) hipLaunchKernelGGL((k3_vec_convert_kernel<double, float>), g, b, 0, (hipStream_t)st, (const double *)d_src, (float *)d_dst, n);
  else if (!src_is_f64 && dst_is_f64) hipLaunchKernelGGL((k3_vec_convert_kernel<float, double>), g, b, 0, (hipStream_t)st, (const float *)d_src, (double *)d_dst, n);
  else if (src_is_f64) hipLaunchKernelGGL((k3_vec_convert_kernel<double, double>), g, b, 0, (hipStream_t)st, (const double *)d_src, (double *)d_dst, n);
  else hipLaunchKernelGGL((k3_vec_convert_kernel<float, float>), g, b, 0, (hipStream_t)st, (const float *)d_src, (float *)d_dst, n);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
extern "C" int k3_vec_pow(const float *d_src, float *d_dst, int32_t n, float power, void *st) {                                         // CuVectorBase::Pow / ApplyPow
  if (n == 0) return K3_OK;
  K3_REQUIRE(d_src && d_dst && n > 0, "k3_vec_pow: bad argument");
  hipLaunchKernelGGL(k3_vec_pow_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, d_src, d_dst, n, power);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
// CuVectorBase::AddVecVec: v = alpha a .* b + beta v
extern "C" int k3_vec_add_vec_vec(float alpha, const float *d_a, const float *d_b, float beta, float *d_v, int32_t n, void *st) {
  if (n == 0) return K3_OK;
  K3_REQUIRE(d_a && d_b && d_v && n > 0, "k3_vec_add_vec_vec: bad argument");
  hipLaunchKernelGGL(k3_vec_add_vec_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, alpha, d_a, d_b, beta, d_v, n);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}

// float64 vectors (cudamatrix/cu-vector.h): op 0 Scale(alpha), 1 Pow(a, alpha) -> v, 2 AddVec: v = alpha a + beta v, 3 AddVecVec: v = alpha a .* b + beta v
extern "C" int k3_vec_f64(int32_t op, double alpha, const double *d_a, const double *d_b, double beta, double *d_v, int32_t n, void *st) {
  if (n == 0) return K3_OK;
  K3_REQUIRE(d_v && n > 0 && op >= 0 && op <= 3 && (op == 0 || d_a) && (op != 3 || d_b), "k3_vec_f64: bad argument");
  hipLaunchKernelGGL(k3_vec64_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, op, alpha, d_a, d_b, beta, d_v, n);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}

extern "C" int k3_vec_unary(int32_t op, float *d_v, int32_t n, void *st) {      // op 0 ApplyLog, 1 ApplyExp, 2 InvertElements
  if (n == 0) return K3_OK;
  K3_REQUIRE(d_v && n > 0 && op >= 0 && op <= 2, "k3_vec_unary: bad argument");
  hipLaunchKernelGGL(k3_vec_unary_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, op, d_v, n);
  K3_HIP_CHECK(hipGetLastError());
  return K3_OK;
}
