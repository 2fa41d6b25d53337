// k3_online.h -- host-side streaming drivers above the C ABI, the C++ twins of kaldi_amd/online.py / nnet3.BatchedStaticNnet3:
//   OnlineFeatures     cudafeat/online-batched-feature-pipeline-cuda.h:83-92 (ComputeFeaturesBatched): per-channel sample stash
//   StaticNnet3        cudadecoder/batched-static-nnet3.{h,cc} (RunBatch): the network planned once for max_batch_size slots of
//                      frames_per_chunk + context frames, per-channel input context stashed on the GPU between calls
// Only input-level state is kept, so chunked outputs are bit-identical to the whole-utterance batch path.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include "k3_host.h"
#include "../../include/k3hip.h"
namespace k3host {
#define K3O_HIP(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)

template <class T> struct DevBuf {          // grow-only device array
  T *p = nullptr; size_t cap = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  T *need(size_t n) { if (n > cap) { if (p) (void)hipFree(p); cap = n + n / 2 + 64; K3O_HIP(hipMalloc((void **)&p, cap * sizeof(T))); } return p; }
  void upload(const std::vector<T> &h) { need(std::max<size_t>(h.size(), 1)); if (!h.empty()) K3O_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
};

class OnlineFeatures {
 public:
  OnlineFeatures(k3_feat_plan *plan, const k3_feat_opts &o, int num_channels) : plan_(plan), dim_(k3_feat_dim(plan)), stash_(num_channels) {
    if (!o.snip_edges) K3H_ERR << "streaming features need --snip-edges=true";
    shift_ = (int)(o.samp_freq * 0.001 * o.frame_shift_ms);
  }
  int Dim() const { return dim_; }
  // chunks[i]: new samples of channels[i]; returns the number of new frames per slot, rows back to back in d_feats (device, dim wide)
  std::vector<int> ComputeFeaturesBatched(const std::vector<int> &channels, const std::vector<std::vector<float>> &chunks, const std::vector<char> &first, float **d_feats) {
    std::vector<float> all; std::vector<int64_t> woff(1, 0), foff(1, 0); std::vector<int> nf(channels.size());
    for (size_t i = 0; i < channels.size(); i++) {
      std::vector<float> &st = stash_[channels[i]];
      if (first[i]) st.clear();
      st.insert(st.end(), chunks[i].begin(), chunks[i].end());
      nf[i] = k3_feat_num_frames(plan_, (int64_t)st.size());
      all.insert(all.end(), st.begin(), st.end()); woff.push_back((int64_t)all.size()); foff.push_back(foff.back() + nf[i]);
    }
    const int64_t tot = foff.back();
    *d_feats = feats_.need((size_t)std::max<int64_t>(tot, 1) * dim_);
    if (tot > 0) {
      waves_.upload(all); woff_.upload(woff); foff_.upload(foff);
      K3H_CHECK_K3(k3_feat_compute_batch(plan_, waves_.p, woff_.p, foff_.p, (int32_t)channels.size(), tot, feats_.p, dim_, nullptr));
    }
    for (size_t i = 0; i < channels.size(); i++) { std::vector<float> &st = stash_[channels[i]]; st.erase(st.begin(), st.begin() + std::min<size_t>(st.size(), (size_t)nf[i] * shift_)); }
    return nf;
  }
 private:
  k3_feat_plan *plan_; int dim_, shift_; std::vector<std::vector<float>> stash_;
  DevBuf<float> waves_, feats_; DevBuf<int64_t> woff_, foff_;
};

class StaticNnet3 {
 public:
  StaticNnet3(k3_nnet *nnet, int max_batch, int nchannels, int frames_per_chunk, int subsampling, const float *log_priors, float acoustic_scale)
      : B_(max_batch), nch_(nchannels), C_(frames_per_chunk), s_(subsampling) {
    if (C_ <= 0 || C_ % s_) K3H_ERR << "--frames-per-chunk must be a positive multiple of --frame-subsampling-factor";
    k3_nnet_info ni; K3H_CHECK_K3(k3_nnet_get_info(nnet, &ni));
    dim_ = ni.input_dim; odim_ = ni.output_dim; Lc_ = (ni.left_context + s_ - 1) / s_ * s_; Rc_ = ni.right_context; P_ = Lc_ + C_ + Rc_; rps_ = (P_ + s_ - 1) / s_; S_ = Lc_ + Rc_ + C_ + 2 * s_;
    std::vector<int32_t> nf(B_, P_);
    K3H_CHECK_K3(k3_nnet_batch_create(nnet, B_, nf.data(), s_, log_priors, acoustic_scale, &batch_));
    for (int k = 0; k < 2; k++) { stash_[k].need((size_t)nch_ * S_ * dim_); K3O_HIP(hipMemset(stash_[k].p, 0, (size_t)nch_ * S_ * dim_ * 4)); }
    inp_.need((size_t)B_ * P_ * dim_); out_.need((size_t)B_ * rps_ * odim_);
    t_next_.assign(nch_, 0); n_seen_.assign(nch_, 0); lo_.assign(nch_, 0);
  }
  ~StaticNnet3() { if (batch_) k3_nnet_batch_destroy(batch_); }
  int OutputDim() const { return odim_; }
  int FramesPerChunk() const { return C_; }
  void Reset(int ch) { t_next_[ch] = n_seen_[ch] = lo_[ch] = 0; }
  // One planned forward.  d_new: the new frames of the slots back to back (n_new[i] rows each, may be null when all are 0).  Returns per
  // slot (first row, count) of its valid output rows in Out().
  std::vector<std::pair<int, int>> Pass(const std::vector<int> &channels, const float *d_new, const std::vector<int> &n_new, const std::vector<char> &last) {
    std::vector<int32_t> ist((size_t)B_ * P_, -1), inw((size_t)B_ * P_, -1), ust((size_t)nch_ * S_, -1), unw((size_t)nch_ * S_, -1);
    std::vector<char> touched(nch_, 0); for (int ch : channels) touched[ch] = 1;
    for (int ch = 0; ch < nch_; ch++) if (!touched[ch]) for (int64_t k = 0; k < n_seen_[ch] - lo_[ch]; k++) ust[(size_t)ch * S_ + k] = (int32_t)(ch * S_ + k);
    std::vector<std::pair<int, int>> res; int64_t noff = 0, total_new = 0; for (int n : n_new) total_new += n;
    for (size_t i = 0; i < channels.size(); i++) {
      const int ch = channels[i]; const int64_t avail = n_seen_[ch] + n_new[i], tn = t_next_[ch];
      int64_t count = 0;
      if (last[i]) count = avail > tn ? (avail - tn + s_ - 1) / s_ : 0;
      else if (avail - 1 - Rc_ - tn >= 0) count = (avail - 1 - Rc_ - tn) / s_ + 1;
      count = std::min<int64_t>(count, C_ / s_);
      auto source = [&](int64_t tau, int32_t *st, int32_t *nw) { if (tau >= n_seen_[ch]) *nw = (int32_t)(noff + tau - n_seen_[ch]); else *st = (int32_t)(ch * S_ + tau - lo_[ch]); };
      if (avail > 0) for (int k = 0; k < P_; k++) { const int64_t tau = std::min<int64_t>(std::max<int64_t>(tn - Lc_ + k, 0), avail - 1); source(tau, &ist[i * P_ + k], &inw[i * P_ + k]); }
      res.push_back({(int)(i * rps_ + Lc_ / s_), (int)count});
      const int64_t tn2 = tn + count * s_, lo2 = std::max<int64_t>(0, tn2 - Lc_);
      if (avail - lo2 > S_) K3H_ERR << "internal: context stash capacity";
      for (int64_t tau = lo2; tau < avail; tau++) source(tau, &ust[(size_t)ch * S_ + (tau - lo2)], &unw[(size_t)ch * S_ + (tau - lo2)]);
      t_next_[ch] = tn2; n_seen_[ch] = avail; lo_[ch] = lo2; noff += n_new[i];
    }
    float *A = stash_[cur_].p, *Bn = stash_[cur_ ^ 1].p;
    i0_.upload(ist); i1_.upload(inw); i2_.upload(ust); i3_.upload(unw);
    K3H_CHECK_K3(k3_mat_copy_rows(inp_.p, dim_, B_ * P_, dim_, A, dim_, i0_.p, nullptr));
    if (total_new > 0) K3H_CHECK_K3(k3_mat_add_rows(1.0f, d_new, dim_, i1_.p, inp_.p, dim_, B_ * P_, dim_, nullptr));
    K3H_CHECK_K3(k3_mat_copy_rows(Bn, dim_, nch_ * S_, dim_, A, dim_, i2_.p, nullptr));
    if (total_new > 0) K3H_CHECK_K3(k3_mat_add_rows(1.0f, d_new, dim_, i3_.p, Bn, dim_, nch_ * S_, dim_, nullptr));
    cur_ ^= 1;
    K3H_CHECK_K3(k3_nnet_forward(batch_, inp_.p, dim_, out_.p, odim_, nullptr));
    return res;
  }
  const float *Out() const { return out_.p; }
  bool Pending(int ch) const { return t_next_[ch] < n_seen_[ch]; }
 private:
  int B_, nch_, C_, s_, dim_ = 0, odim_ = 0, Lc_ = 0, Rc_ = 0, P_ = 0, rps_ = 0, S_ = 0, cur_ = 0;
  k3_nnet_batch *batch_ = nullptr;
  DevBuf<float> stash_[2], inp_, out_; DevBuf<int32_t> i0_, i1_, i2_, i3_;
  std::vector<int64_t> t_next_, n_seen_, lo_;
};
}  // namespace k3host
