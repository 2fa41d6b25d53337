// k3_decoder_lit.hip -- the literal_order token-passing kernel (k3_decoder_literal.h) as its own translation unit: it runs with its own number of
// threads per lane (K3_LIT_BLOCK) while the two-pass kernel, the pruning and the output kernels of k3_decoder.hip keep theirs.  The two files share
// k3_decoder_dev.h (DecParams, the state table, the epsilon closure, the block primitives), each compiled for its block size.
#ifndef K3_LIT_BLOCK
#define K3_LIT_BLOCK 512
#endif
#define K3_DEC_BLOCK K3_LIT_BLOCK
#ifndef K3_LIT_WPE
#define K3_LIT_WPE 4      // two 512-thread workgroups per CU (each within 80 KB of LDS): 4 waves per SIMD = 128 VGPRs per lane (overridable for register-pressure experiments)
#endif
#ifndef K3_LIT_QUEUE
#define K3_LIT_QUEUE 0     // 1: a workgroup decodes lane after lane from the call's work-queue (k3_decoder_config.resident_lanes); see k3_decode_forward_literal_kernel
#endif
#ifndef K3_LIT_PREFETCH_ROW
#define K3_LIT_PREFETCH_ROW 0     // 1: LDS-resident frames request their whole log-likelihood row at the start of the frame (see lit_frame_fast)
#endif
#ifndef K3_LIT_PREFETCH_ARCS
#define K3_LIT_PREFETCH_ARCS 0     // 1: the last phase of an LDS-resident frame requests the first emitting arc of every token of the frame it hands over
#endif
#ifndef K3_LIT_PF_SEQ
#define K3_LIT_PF_SEQ 0     // 1: passes A / B of the general path request arc records two groups and log-likelihoods one group ahead (wave_expand_seq_pf)
#endif
#ifndef K3_LIT_CAPTURE
#define K3_LIT_CAPTURE 0     // 1 (k3_decoder_lit_cap.hip only): the build that decodes the one frame the first-frame template is captured from
#endif
#ifndef K3_LIT_CSH
#define K3_LIT_CSH 0     // 1: passes A / B cut a frame of few tokens into chunks of fewer than 64 tokens (more, shorter chunks per wavefront); measured: see DESIGN.md 4
#endif
#include "k3_decoder_dev.h"

namespace {
#define K3_LIT_FORWARD
#include "k3_decoder_literal.h"
}  // namespace

extern "C" int k3_lit_fast_tokens() { return kFT; }
extern "C" int k3_lit_has_queue() { return K3_LIT_QUEUE; }      // capacity of the LDS-resident frame path (tokens of a frame)
// (exclusive launches ask for a little more than half of a CU's 160 KB: no second lane fits beside the workgroup, another kernel's workgroups do)
constexpr size_t kLitExclusiveLds = 82 * 1024;
extern "C" int k3_lit_forward_prepare() {
  if (hipFuncSetAttribute((const void *)k3_decode_frame0_from_template_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLitArena) != hipSuccess) return -1;
  return hipFuncSetAttribute((const void *)k3_decode_forward_literal_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLitArena > kLitExclusiveLds ? kLitArena : kLitExclusiveLds)) == hipSuccess ? 0 : -1;
}
extern "C" void k3_lit_forward_launch(const void *params, size_t params_bytes, int nworkgroups, int flags, hipStream_t stream) {      // flags: 1 = exclusive LDS, 2 = no lane of this call can use the template kernels
  const int exclusive = flags & 1; const bool templates = (flags & 2) == 0;
  DecParams p; static_assert(sizeof(DecParams) % 8 == 0, "DecParams is copied between translation units");
  if (params_bytes != sizeof(DecParams)) { fprintf(stderr, "k3_lit_forward_launch: DecParams size mismatch\n"); abort(); }
  memcpy(&p, params, sizeof(p));
  const unsigned nl_ = p.q_lanes ? (unsigned)p.q_n : (unsigned)nworkgroups;
  if (templates && p.tpl_n > 0) hipLaunchKernelGGL(k3_decode_init_from_template_kernel, dim3(nl_), dim3(256), 0, stream, p, const_cast<int *>(p.fresh));
  if (templates && p.tpl_n > 0 && p.t0_n > 0) hipLaunchKernelGGL(k3_decode_frame0_from_template_kernel, dim3(nl_), dim3(kBlock), kLitArena, stream, p);
  hipLaunchKernelGGL(k3_decode_forward_literal_kernel, dim3(nworkgroups), dim3(kBlock), exclusive && kLitExclusiveLds > kLitArena ? kLitExclusiveLds : kLitArena, stream, p);
}
