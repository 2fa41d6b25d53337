// Sustained FP32 MFMA rate of the device (v_mfma_f32_32x32x2_f32, the instruction the TDNN-F GEMM uses): wavefronts that do nothing but issue
// independent MFMAs from registers.  What the GEMM's roofline fraction should be read against when the clock under sustained matrix load is below
// the 2.4 GHz the data-sheet peak (157.3 TFLOP/s) assumes.   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0) {
  const long long t_begin = (long long)__builtin_readcyclecounter();
  f32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 12; u++) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[2] = (long long)__builtin_readcyclecounter() - t_begin;
  float s = 0.f;
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  float *out; hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nacc = 1; nacc <= 4; nacc++)
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; wgs_per_cu *= 2) {
    const int grid = prop.multiProcessorCount * wgs_per_cu;
    for (int rep = 0; rep < 2; rep++) {
      const int iters = 20000;      // x 12 MFMAs x 64 cycles = 41 M cycles ~ 17 ms at 2.4 GHz with one wavefront per SIMD
      hipEventRecord(e0); if (nacc == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f); else if (nacc == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f); else if (nacc == 3) hipLaunchKernelGGL(mfma_loop<3>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f); else hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long ticks = 0; hipMemcpy(&ticks, (char *)out + 16, 8, hipMemcpyDeviceToHost);
      const double flops = (double)grid * 4 * iters * 12 * 4096.0;
      printf("%d independent accumulators, %d CUs, %d wavefronts/SIMD: %.2f ms, %.1f TFLOP/s  (%.3f of 157.3; implied clock %.2f GHz; s_memtime of workgroup 0: %.3f G ticks/s of launch time)\n", nacc, prop.multiProcessorCount, wgs_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3,
             flops / ms / 1e6 / (prop.multiProcessorCount * 4 * 64.0), ticks / ms / 1e6);
    }
  }
  return 0;
}
