// k3_common.h -- error plumbing shared by the libk3hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/k3hip.h"

namespace k3 {
void set_error(const char *fmt, ...);
inline int fail(int code, const char *what, const char *file, int line) {
  set_error("%s (%s:%d)", what, file, line);
  return code;
}
// k3_cmvn_online_batch_resume without the wait for its error flag (k3_feat.hip): the streaming i-vector extractor queues it between its own kernels.  The flag can only be
// raised by global statistics without frames, which k3_ivector_create refuses
// one stream of a batched resume: rows [0, t_begin) of `in` are history, rows [t_begin, rows) are normalised into out[t] (both `dim` wide, dense), carry [dim][3]
struct CmvnSeg { const float *in; float *out; long long rows, t_begin; double *carry; };
int cmvn_online_resume_segs_async(const CmvnSeg *d_segs, int num_segs, int dim, const void *opts, const double *d_global_stats, void *stream);
int cmvn_online_resume_async(const float *d_in, long long ld_in, float *d_out, long long ld_out, int dim, const long long *d_frame_offsets, int num_utts, const void *opts,
                             const double *d_global_stats, const long long *d_t_begin, double *d_carry, void *stream);
}  // namespace k3

#define K3_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) {                                                            \
      k3::set_error("HIP error %s at %s:%d: %s", hipGetErrorName(e__), __FILE__, __LINE__, #expr); \
      return K3_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)
#define K3_REQUIRE(cond, msg)                                   \
  do {                                                          \
    if (!(cond)) return k3::fail(K3_ERR_ARG, msg, __FILE__, __LINE__); \
  } while (0)
