// k3_decoder_lit_cap.hip -- the CAPTURE build of the literal_order token-passing kernel (k3_decoder_literal.h with K3_LIT_CAPTURE): the same kernel, with the replay's records
// kept in the lane's HBM scratch, the frame's creation ranks left in place and a handful of scalars written out, so that k3_decoder_create can read the structure of an
// utterance's FIRST FRAME out of one lane after decoding one frame with it (the first-frame template, DecParams::t0_*).  Never on the decoding path: one launch of one
// workgroup per decoder object.
#define K3_LIT_CAPTURE 1
#define K3_LIT_QUEUE 0
#define K3_LIT_CSH 0
#define K3_LIT_PREFETCH_ROW 0
#define K3_LIT_PREFETCH_ARCS 0
#define K3_DEC_BLOCK 512
#define K3_LIT_WPE 4
#include "k3_decoder_dev.h"

namespace {
#define K3_LIT_FORWARD
#include "k3_decoder_literal.h"
}  // namespace

extern "C" int k3_lit_capture_launch(const void *params, size_t params_bytes, int nworkgroups, hipStream_t stream) {
  DecParams p;
  if (params_bytes != sizeof(DecParams)) return -1;
  memcpy(&p, params, sizeof(p));
  if (hipFuncSetAttribute((const void *)k3_decode_forward_literal_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLitArena) != hipSuccess) return -1;
  if (p.tpl_n > 0) hipLaunchKernelGGL(k3_decode_init_from_template_kernel, dim3(nworkgroups), dim3(256), 0, stream, p, const_cast<int *>(p.fresh));
  hipLaunchKernelGGL(k3_decode_forward_literal_kernel, dim3(nworkgroups), dim3(kBlock), kLitArena, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
