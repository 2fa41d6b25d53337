// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters for the access pattern of the token-passing kernels: small (4 B / 16 B) reads and writes at random addresses
// of a footprint far beyond the L2 (4 MB per XCD) and the Infinity Cache (256 MB), next to a streaming read / write of the same buffer.  Every kernel has its own name, so the
// counter passes of tools/pmc_calib.sh can be matched to the bytes each one asked for (printed below): counter bytes / requested bytes = what a byte of this pattern costs
// at the HBM side (a 4-byte random read fetches at least one 32 B sector, usually a 64 B or 128 B line).
//   hipcc --offload-arch=gfx950 -O3 -o hbm_random_access hbm_random_access.hip && ./hbm_random_access
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e__)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__global__ __launch_bounds__(256) void calib_random_read_4B(const uint32_t *buf, uint64_t n_words, uint64_t per_thread, uint32_t *out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; uint32_t s = 0;
  for (uint64_t k = 0; k < per_thread; k++) s += buf[mix(t * per_thread + k) % n_words];
  if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void calib_random_read_16B(const uint4 *buf, uint64_t n_vec, uint64_t per_thread, uint32_t *out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; uint32_t s = 0;
  for (uint64_t k = 0; k < per_thread; k++) { const uint4 v = buf[mix(t * per_thread + k) % n_vec]; s += v.x ^ v.y ^ v.z ^ v.w; }
  if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void calib_random_write_4B(uint32_t *buf, uint64_t n_words, uint64_t per_thread) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t k = 0; k < per_thread; k++) buf[mix(t * per_thread + k) % n_words] = (uint32_t)k;
}
__global__ __launch_bounds__(256) void calib_random_write_16B(uint4 *buf, uint64_t n_vec, uint64_t per_thread) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t k = 0; k < per_thread; k++) buf[mix(t * per_thread + k) % n_vec] = make_uint4((uint32_t)k, 1, 2, 3);
}
__global__ __launch_bounds__(256) void calib_random_atomic_4B(uint32_t *buf, uint64_t n_words, uint64_t per_thread) {      // the decoder's per-state minimum: a returning L2 atomic on a random word
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; uint32_t s = 0;
  for (uint64_t k = 0; k < per_thread; k++) s += __hip_atomic_fetch_min(&buf[mix(t * per_thread + k) % n_words], (uint32_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (s == 0x12345678u) buf[0] = s;
}
__global__ __launch_bounds__(256) void calib_stream_read_16B(const uint4 *buf, uint64_t n_vec, uint32_t *out) {
  uint32_t s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = buf[i]; s += v.x ^ v.w; }
  if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void calib_stream_write_16B(uint4 *buf, uint64_t n_vec) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (uint64_t)gridDim.x * blockDim.x) buf[i] = make_uint4((uint32_t)i, 0, 0, 0);
}
int main() {
  const uint64_t bytes = 8ull << 30, n_words = bytes / 4, n_vec = bytes / 16;      // 8 GiB: 32x the Infinity Cache
  void *buf; uint32_t *out; CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc((void **)&out, 64)); CHECK(hipMemset(buf, 1, bytes));
  const int grid = 256 * 16, block = 256; const uint64_t per_thread = 256, n_acc = (uint64_t)grid * block * per_thread;      // 2^28 accesses per kernel
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto timed = [&](const char *name, double req_bytes, auto launch) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s requested_bytes %.0f  ms %.3f  requested GB/s %.1f\n", name, req_bytes, ms, req_bytes / ms / 1e6);
  };
  timed("calib_random_read_4B", 4.0 * n_acc, [&] { hipLaunchKernelGGL(calib_random_read_4B, dim3(grid), dim3(block), 0, 0, (const uint32_t *)buf, n_words, per_thread, out); });
  timed("calib_random_read_16B", 16.0 * n_acc, [&] { hipLaunchKernelGGL(calib_random_read_16B, dim3(grid), dim3(block), 0, 0, (const uint4 *)buf, n_vec, per_thread, out); });
  timed("calib_random_write_4B", 4.0 * n_acc, [&] { hipLaunchKernelGGL(calib_random_write_4B, dim3(grid), dim3(block), 0, 0, (uint32_t *)buf, n_words, per_thread); });
  timed("calib_random_write_16B", 16.0 * n_acc, [&] { hipLaunchKernelGGL(calib_random_write_16B, dim3(grid), dim3(block), 0, 0, (uint4 *)buf, n_vec, per_thread); });
  timed("calib_random_atomic_4B", 4.0 * n_acc, [&] { hipLaunchKernelGGL(calib_random_atomic_4B, dim3(grid), dim3(block), 0, 0, (uint32_t *)buf, n_words, per_thread); });
  timed("calib_stream_read_16B", (double)bytes, [&] { hipLaunchKernelGGL(calib_stream_read_16B, dim3(grid), dim3(block), 0, 0, (const uint4 *)buf, n_vec, out); });
  timed("calib_stream_write_16B", (double)bytes, [&] { hipLaunchKernelGGL(calib_stream_write_16B, dim3(grid), dim3(block), 0, 0, (uint4 *)buf, n_vec); });
  CHECK(hipDeviceSynchronize());
  return 0;
}
