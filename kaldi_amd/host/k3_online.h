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
  ~DevBuf() { if (p) (void)hipFree(p); for (Stage &g : stage_) { if (g.p) (void)hipHostFree(g.p); if (g.ev) (void)hipEventDestroy(g.ev); } }
  // (hipFree waits for the device: no kernel still reads the old block)
  T *need(size_t n) {
    if (n > cap) {
      if (p) (void)hipFree(p);
      cap = n + n / 2 + 64;
      K3O_HIP(hipMalloc((void **)&p, cap * sizeof(T)));
    }
    return p;
  }
  void upload(const std::vector<T> &h) { need(std::max<size_t>(h.size(), 1)); if (!h.empty()) K3O_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
  // The streaming path's form: the host never waits for the device.  begin_upload(n) hands out a page-locked staging block to fill (a ring of kStages blocks, each guarded by an
  // event recorded behind its copy, so a block is only waited for when the host is kStages uploads ahead of the device); end_upload queues the copy on `st`,
  // behind whatever on that
  // stream still reads the device block (every chunk round of the streaming programs paid ~10 blocking pageable copies on the null stream).
  T *begin_upload(size_t n) {
    Stage &g = stage_[seq_ % kStages];
    if (g.used) K3O_HIP(hipEventSynchronize(g.ev));
    if (n > g.cap) { if (g.p) (void)hipHostFree(g.p); g.cap = n + n / 2 + 64; K3O_HIP(hipHostMalloc((void **)&g.p, g.cap * sizeof(T), hipHostMallocDefault)); }
    return g.p;
  }
  void end_upload(size_t n, hipStream_t st) {
    Stage &g = stage_[seq_++ % kStages]; need(std::max<size_t>(n, 1));
    if (n == 0) return;
    K3O_HIP(hipMemcpyAsync(p, g.p, n * sizeof(T), hipMemcpyHostToDevice, st));
    if (!g.ev) K3O_HIP(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
    K3O_HIP(hipEventRecord(g.ev, st)); g.used = true;
  }
  void upload_async(const std::vector<T> &h, hipStream_t st) {
    T *dst = begin_upload(h.size());
    if (!h.empty()) memcpy(dst, h.data(), h.size() * sizeof(T));
    end_upload(h.size(), st);
  }
 private:
  static constexpr int kStages = 4;
  struct Stage { T *p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
  Stage stage_[kStages]; unsigned seq_ = 0;
};

template <class T> struct PinnedBuf {       // grow-only page-locked host array (staging for asynchronous copies)
  T *p = nullptr; size_t cap = 0;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  T *need(size_t n) { if (n > cap) { if (p) (void)hipHostFree(p); cap = n + n / 2 + 64; K3O_HIP(hipHostMalloc((void **)&p, cap * sizeof(T), hipHostMallocDefault)); } return p; }
};

class OnlineFeatures {
 public:
  OnlineFeatures(k3_feat_plan *plan, const k3_feat_opts &o, int num_channels, hipStream_t stream = nullptr) : plan_(plan), dim_(k3_feat_dim(plan)),
      stash_(num_channels), stream_(stream) {
    if (!o.snip_edges) K3H_ERR << "streaming features need --snip-edges=true";
    shift_ = (int)(o.samp_freq * 0.001 * o.frame_shift_ms);
  }
  int Dim() const { return dim_; }
  // chunks[i]: new samples of channels[i]; returns the number of new frames per slot, rows back to back in d_feats (device, dim wide)
  std::vector<int> ComputeFeaturesBatched(const std::vector<int> &channels, const std::vector<std::vector<float>> &chunks, const std::vector<char> &first, float **d_feats) {
    std::vector<const float *> ptr(chunks.size()); std::vector<size_t> len(chunks.size());
    for (size_t i = 0; i < chunks.size(); i++) { ptr[i] = chunks[i].data(); len[i] = chunks[i].size(); }
    return ComputeFeaturesBatched(channels, ptr.data(), len.data(), first, d_feats);
  }
  // the same over (pointer, length) pairs: a caller that holds the whole waveform does not copy its chunks out first.  The samples of a slot -- what the channel stashed for the
  // frame overlap, then the chunk -- go straight into the page-locked staging block of the waveform buffer and travel asynchronously on the work stream.
  std::vector<int> ComputeFeaturesBatched(const std::vector<int> &channels, const float *const *chunk, const size_t *chunk_len, const std::vector<char> &first, float **d_feats) {
    std::vector<int64_t> woff(1, 0), foff(1, 0); std::vector<int> nf(channels.size());
    for (size_t i = 0; i < channels.size(); i++) {
      std::vector<float> &st = stash_[channels[i]]; if (first[i]) st.clear();
      const int64_t n = (int64_t)(st.size() + chunk_len[i]); nf[i] = k3_feat_num_frames(plan_, n);
      woff.push_back(woff.back() + n); foff.push_back(foff.back() + nf[i]);
    }
    const int64_t tot = foff.back();
    *d_feats = feats_.need((size_t)std::max<int64_t>(tot, 1) * dim_);
    float *all = waves_.begin_upload((size_t)woff.back());
    for (size_t i = 0; i < channels.size(); i++) {
      std::vector<float> &st = stash_[channels[i]]; float *dst = all + woff[i];
      if (!st.empty()) memcpy(dst, st.data(), st.size() * sizeof(float));
      if (chunk_len[i]) memcpy(dst + st.size(), chunk[i], chunk_len[i] * sizeof(float));
      // what the next call still needs: the samples behind the last frame's shift
      const size_t n = st.size() + chunk_len[i], used = std::min<size_t>(n, (size_t)nf[i] * shift_);
      std::vector<float> rest(dst + used, dst + n); st.swap(rest);
    }
    if (tot > 0) {
      waves_.end_upload((size_t)woff.back(), stream_); woff_.upload_async(woff, stream_); foff_.upload_async(foff, stream_);
      K3H_CHECK_K3(k3_feat_compute_batch(plan_, waves_.p, woff_.p, foff_.p, (int32_t)channels.size(), tot, feats_.p, dim_, stream_));
    } else waves_.end_upload(0, stream_);
    return nf;
  }
 private:
  k3_feat_plan *plan_; int dim_, shift_; std::vector<std::vector<float>> stash_; hipStream_t stream_ = nullptr;
  DevBuf<float> waves_, feats_; DevBuf<int64_t> woff_, foff_;
};

// Two engines behind one interface.  (i) The stateful engine of the C ABI (k3_nnet_stream_*, round 5): every node keeps its last rows per channel and a pass evaluates the
// frames_per_chunk NEW frames of every listed channel, nothing else -- used whenever the model allows it.  (ii) The reference's scheme (BatchedStaticNnet3::RunBatch): the network
// planned once for max_batch slots of left context + chunk + right context, the context frames stashed per channel and re-evaluated with every chunk -- for models with an
// i-vector input or row operations inside the network.  Either way the rows are those of the whole-utterance forward, bit for bit.
class StaticNnet3 {
 public:
  struct Rows { int first, count, stride; };      // a slot's valid output rows in Out(): first, first + stride, ...
  StaticNnet3(k3_nnet *nnet, int max_batch, int nchannels, int frames_per_chunk, int subsampling, const float *log_priors, float acoustic_scale, hipStream_t stream = nullptr)
      : stream_(stream), B_(max_batch), nch_(nchannels), C_(frames_per_chunk), s_(subsampling) {
    if (C_ <= 0 || C_ % s_) K3H_ERR << "--frames-per-chunk must be a positive multiple of --frame-subsampling-factor";
    k3_nnet_info ni; K3H_CHECK_K3(k3_nnet_get_info(nnet, &ni));
    // (only K3_ERR_UNSUPPORTED -- a model the stateful engine cannot run -- selects the chunk + context scheme, and says so once; a device or allocation failure is an error,
    // not a silent change to the several-times slower engine: ADVICE r5)
    int rc_stream = K3_ERR_UNSUPPORTED;
    if (ni.ivector_dim == 0 && !getenv("K3_ONLINE_RECOMPUTE_CONTEXT")) {
      rc_stream = k3_nnet_stream_create(nnet, nch_, C_, s_, log_priors, acoustic_scale, &inc_);
      if (rc_stream == K3_ERR_UNSUPPORTED) K3H_LOG << "StaticNnet3: the stateful streaming engine does not take this model (" << k3_last_error() << "); chunks are evaluated with their context";
      else if (rc_stream != K3_OK) K3H_CHECK_K3(rc_stream);
    }
    if (rc_stream == K3_OK) {
      k3_nnet_stream_info si;
      K3H_CHECK_K3(k3_nnet_stream_get_info(inc_, &si));
      dim_ = ni.input_dim;
      odim_ = ni.output_dim;
      Rc_ = si.right_context;
      n_out_ = si.output_rows_per_pass;
      first_out_ = si.first_output_time;
      out_.need((size_t)n_out_ * nch_ * odim_); out2_.need((size_t)n_out_ * nch_ * odim_); passes_.assign(nch_, 0); total_.assign(nch_, 0); ended_.assign(nch_, 0);
      return;
    }
    inc_ = nullptr;      // (K3_ERR_UNSUPPORTED: the model needs the chunk + context scheme)
    dim_ = ni.input_dim;
    odim_ = ni.output_dim;
    Lc_ = (ni.left_context + s_ - 1) / s_ * s_;
    Rc_ = ni.right_context;
    P_ = Lc_ + C_ + Rc_;
    rps_ = (P_ + s_ - 1) / s_;
    S_ = Lc_ + Rc_ + C_ + 2 * s_;
    std::vector<int32_t> nf(B_, P_); ivdim_ = ni.ivector_dim;
    // models with the recipes' i-vector input: every slot is an "utterance" with ONE i-vector (the --ivectors form of the planner), handed in per pass
    if (ivdim_ > 0) { K3H_CHECK_K3(k3_nnet_batch_create_ivector(nnet, B_, nf.data(), s_, log_priors, acoustic_scale, C_, 0, nullptr, &batch_)); iv_.need((size_t)B_ * ivdim_); }
    else K3H_CHECK_K3(k3_nnet_batch_create(nnet, B_, nf.data(), s_, log_priors, acoustic_scale, &batch_));
    for (int k = 0; k < 2; k++) { stash_[k].need((size_t)nch_ * S_ * dim_); K3O_HIP(hipMemsetAsync(stash_[k].p, 0, (size_t)nch_ * S_ * dim_ * 4, stream_)); }
    inp_.need((size_t)B_ * P_ * dim_); out_.need((size_t)B_ * rps_ * odim_); out2_.need((size_t)B_ * rps_ * odim_);
    t_next_.assign(nch_, 0); n_seen_.assign(nch_, 0); lo_.assign(nch_, 0);
  }
  ~StaticNnet3() { if (batch_) k3_nnet_batch_destroy(batch_); if (inc_) k3_nnet_stream_destroy(inc_); }
  bool Stateful() const { return inc_ != nullptr; }
  int OutputDim() const { return odim_; }
  int FramesPerChunk() const { return C_; }
  void Reset(int ch) { if (inc_) { passes_[ch] = 0; total_[ch] = 0; ended_[ch] = 0; return; } t_next_[ch] = n_seen_[ch] = lo_[ch] = 0; }
  // One planned forward.  d_new: the new frames of the slots back to back (n_new[i] rows each, may be null when all are 0).  Returns per
  // slot (first row, count) of its valid output rows in Out().
  // d_iv (models with an i-vector input): [channels.size() x IvectorDim()] on the device, the i-vector each slot's chunk is evaluated with
  // (decodable-online-looped.cc:166-205: one per chunk)
  std::vector<Rows> Pass(const std::vector<int> &channels, const float *d_new, const std::vector<int> &n_new, const std::vector<char> &last, const float *d_iv = nullptr) {
    if ((ivdim_ > 0) != (d_iv != nullptr)) K3H_ERR << "StaticNnet3::Pass: the model " << (ivdim_ > 0 ? "has" : "has no") << " i-vector input";
    if (inc_) return PassStateful(channels, d_new, n_new, last);
    std::vector<int32_t> ist((size_t)B_ * P_, -1), inw((size_t)B_ * P_, -1), ust((size_t)nch_ * S_, -1), unw((size_t)nch_ * S_, -1);
    std::vector<char> touched(nch_, 0); for (int ch : channels) touched[ch] = 1;
    for (int ch = 0; ch < nch_; ch++) if (!touched[ch]) for (int64_t k = 0; k < n_seen_[ch] - lo_[ch]; k++) ust[(size_t)ch * S_ + k] = (int32_t)(ch * S_ + k);
    std::vector<Rows> res; int64_t noff = 0, total_new = 0; for (int n : n_new) total_new += n;
    for (size_t i = 0; i < channels.size(); i++) {
      const int ch = channels[i]; const int64_t avail = n_seen_[ch] + n_new[i], tn = t_next_[ch];
      int64_t count = 0;
      if (last[i]) count = avail > tn ? (avail - tn + s_ - 1) / s_ : 0;
      else if (avail - 1 - Rc_ - tn >= 0) count = (avail - 1 - Rc_ - tn) / s_ + 1;
      count = std::min<int64_t>(count, C_ / s_);
      auto source = [&](int64_t tau, int32_t *st, int32_t *nw) {
        if (tau >= n_seen_[ch]) *nw = (int32_t)(noff + tau - n_seen_[ch]);
        else *st = (int32_t)(ch * S_ + tau - lo_[ch]);
      };
      if (avail > 0) for (int k = 0; k < P_; k++) {
        const int64_t tau = std::min<int64_t>(std::max<int64_t>(tn - Lc_ + k, 0), avail - 1);
        source(tau, &ist[i * P_ + k], &inw[i * P_ + k]);
      }
      res.push_back({(int)(i * rps_ + Lc_ / s_), (int)count, 1});
      const int64_t tn2 = tn + count * s_, lo2 = std::max<int64_t>(0, tn2 - Lc_);
      if (avail - lo2 > S_) K3H_ERR << "internal: context stash capacity";
      for (int64_t tau = lo2; tau < avail; tau++) source(tau, &ust[(size_t)ch * S_ + (tau - lo2)], &unw[(size_t)ch * S_ + (tau - lo2)]);
      t_next_[ch] = tn2; n_seen_[ch] = avail; lo_[ch] = lo2; noff += n_new[i];
    }
    float *A = stash_[cur_].p, *Bn = stash_[cur_ ^ 1].p;
    // (page-locked rings: the host does not wait)
    i0_.upload_async(ist, stream_);
    i1_.upload_async(inw, stream_);
    i2_.upload_async(ust, stream_);
    i3_.upload_async(unw, stream_);
    K3H_CHECK_K3(k3_mat_copy_rows(inp_.p, dim_, B_ * P_, dim_, A, dim_, i0_.p, stream_));
    if (total_new > 0) K3H_CHECK_K3(k3_mat_add_rows(1.0f, d_new, dim_, i1_.p, inp_.p, dim_, B_ * P_, dim_, stream_));
    K3H_CHECK_K3(k3_mat_copy_rows(Bn, dim_, nch_ * S_, dim_, A, dim_, i2_.p, stream_));
    if (total_new > 0) K3H_CHECK_K3(k3_mat_add_rows(1.0f, d_new, dim_, i3_.p, Bn, dim_, nch_ * S_, dim_, stream_));
    cur_ ^= 1;
    if (ivdim_ > 0) {
      K3O_HIP(hipMemsetAsync(iv_.p, 0, (size_t)B_ * ivdim_ * 4, stream_));
      K3O_HIP(hipMemcpyAsync(iv_.p, d_iv, channels.size() * (size_t)ivdim_ * 4, hipMemcpyDeviceToDevice, stream_));
      K3H_CHECK_K3(k3_nnet_forward_ivector(batch_, inp_.p, dim_, iv_.p, ivdim_, (which_ ? out2_ : out_).p, odim_, stream_));
    } else K3H_CHECK_K3(k3_nnet_forward(batch_, inp_.p, dim_, (which_ ? out2_ : out_).p, odim_, stream_));
    return res;
  }
  // the rows of the latest pass.  Two buffers alternate (SelectOut before a pass): a consumer on another stream -- token passing -- may still read pass k - 1's rows while pass k is
  // evaluated; the caller orders pass k + 1 behind the consumer of pass k - 1 (an event), as it did for its own copy of the rows
  const float *Out() const { return (which_ ? out2_ : out_).p; }
  void SelectOut(int which) { which_ = which & 1; }
  int OutputStride() const { return odim_; }
  bool Pending(int ch) const {
    if (!inc_) return t_next_[ch] < n_seen_[ch];
    // stateful engine: an ended stream needs more passes (its last frame replicated) until the newest output row reaches its last output time
    return ended_[ch] && total_[ch] > 0 && passes_[ch] * (int64_t)C_ - 1 - Rc_ < (total_[ch] - 1) / s_ * s_;
  }
  int IvectorDim() const { return ivdim_; }
 private:
  // One pass of the stateful engine: slot i's channel consumes its n_new[i] rows of d_new (frames_per_chunk of them, fewer / none only with last[i]: the end of its stream).
  // A channel's FIRST pass seeds its histories with the response to its first frame replicated (k3_nnet_stream_reset).  Output row k of the pass belongs to channel-local time
  // first_out_ + passes * C + k * s: the valid ones -- t >= 0, and below the stream's length once that is known -- are what the caller hands to the decoder.
  std::vector<Rows> PassStateful(const std::vector<int> &channels, const float *d_new, const std::vector<int> &n_new, const std::vector<char> &last) {
    std::vector<int64_t> start(nch_, 0); std::vector<int32_t> count(nch_, -1), fresh, fresh_row; int64_t noff = 0;
    for (size_t i = 0; i < channels.size(); i++) {
      const int ch = channels[i];
      if (n_new[i] != C_ && !last[i]) K3H_ERR << "StaticNnet3::Pass: the stateful engine takes whole chunks of " << C_ << " frames (got " << n_new[i] <<
          " before the end of the stream)";
      if (ended_[ch] && n_new[i] > 0) K3H_ERR << "StaticNnet3::Pass: frames for a stream that has ended";
      // (an empty stream: nothing to evaluate)
      if (passes_[ch] == 0 && total_[ch] == 0) {
        if (n_new[i] == 0) {
          count[ch] = -1;
          continue;
        }
        fresh.push_back(ch);
        fresh_row.push_back((int32_t)noff);
      }
      start[ch] = noff; count[ch] = n_new[i]; noff += n_new[i];
    }
    if (!fresh.empty()) {
      float *f0 = first_.need(fresh.size() * (size_t)dim_); fidx_.upload_async(fresh_row, stream_);
      K3H_CHECK_K3(k3_mat_copy_rows(f0, dim_, (int32_t)fresh.size(), dim_, d_new, dim_, fidx_.p, stream_));
      K3H_CHECK_K3(k3_nnet_stream_reset(inc_, fresh.data(), (int32_t)fresh.size(), f0, dim_, stream_));
    }
    K3H_CHECK_K3(k3_nnet_stream_forward(inc_, d_new, dim_, start.data(), count.data(), (which_ ? out2_ : out_).p, odim_, stream_));
    std::vector<Rows> res;
    for (size_t i = 0; i < channels.size(); i++) {
      const int ch = channels[i];
      if (count[ch] < 0) { res.push_back({0, 0, nch_}); if (last[i]) ended_[ch] = 1; continue; }
      total_[ch] += n_new[i]; if (last[i]) ended_[ch] = 1;
      const int64_t t0 = first_out_ + passes_[ch] * (int64_t)C_; passes_[ch]++;
      int k0 = 0; while (k0 < n_out_ && t0 + (int64_t)k0 * s_ < 0) k0++;
      int k1 = n_out_; if (ended_[ch]) while (k1 > k0 && t0 + (int64_t)(k1 - 1) * s_ >= total_[ch]) k1--;
      res.push_back({k0 * nch_ + ch, k1 - k0, nch_});
    }
    return res;
  }
  k3_nnet_stream *inc_ = nullptr; int n_out_ = 0, first_out_ = 0; std::vector<int64_t> passes_, total_; std::vector<char> ended_; DevBuf<float> first_; DevBuf<int32_t> fidx_;
  hipStream_t stream_ = nullptr; int ivdim_ = 0; DevBuf<float> iv_;
  int B_, nch_, C_, s_, dim_ = 0, odim_ = 0, Lc_ = 0, Rc_ = 0, P_ = 0, rps_ = 0, S_ = 0, cur_ = 0;
  k3_nnet_batch *batch_ = nullptr;
  DevBuf<float> stash_[2], inp_, out_, out2_; int which_ = 0; DevBuf<int32_t> i0_, i1_, i2_, i3_;
  std::vector<int64_t> t_next_, n_seen_, lo_;
};
// The i-vector extractor of an --ivector-extraction-config (OnlineNnet2FeaturePipelineInfo's ivector_extractor_info, online2/online-nnet2-feature-pipeline.cc:70-80) on the GPU
inline k3_ivector *CreateIvectorExtractor(const IvectorExtractionInfo &iv_info, int fdim) {
  k3_ivector_model m; memset(&m, 0, sizeof m);
  m.feat_dim = iv_info.global_cmvn_stats.cols - 1;
  m.lda_rows = iv_info.lda_rows;
  m.lda_cols = iv_info.lda_cols;
  m.num_gauss = iv_info.ubm.num_gauss;
  m.ivector_dim = iv_info.ie.ivector_dim;
  m.lda = iv_info.lda.data();
  m.global_cmvn_stats = iv_info.global_cmvn_stats.data.data();
  m.gconsts = iv_info.ubm.gconsts.data();
  m.means_invvars = iv_info.ubm.means_invvars.data();
  m.inv_vars = iv_info.ubm.inv_vars.data();
  m.M = iv_info.ie.M.data(); m.sigma_inv = iv_info.ie.sigma_inv.data(); m.prior_offset = iv_info.ie.prior_offset;
  k3_ivector_opts o; k3_ivector_opts_default(&o);
  o.left_context = iv_info.left_context;
  o.right_context = iv_info.right_context;
  o.num_gselect = iv_info.num_gselect;
  o.min_post = iv_info.min_post;
  o.posterior_scale = iv_info.posterior_scale;
  o.max_count = iv_info.max_count;
  o.ivector_period = iv_info.ivector_period; o.num_cg_iters = iv_info.num_cg_iters; o.online_cmvn_iextractor = iv_info.online_cmvn_iextractor;
  o.cmvn.cmn_window = iv_info.cmn_window;
  o.cmvn.speaker_frames = iv_info.speaker_frames;
  o.cmvn.global_frames = iv_info.global_frames;
  o.cmvn.normalize_mean = iv_info.normalize_mean;
  o.cmvn.normalize_variance = iv_info.normalize_variance;
  if (m.feat_dim != fdim) K3H_ERR << "The i-vector extractor expects features of dimension " << m.feat_dim << " but the feature config gives " << fdim;
  k3_ivector *ivx = nullptr; K3H_CHECK_K3(k3_ivector_create(&m, &o, &ivx));
  return ivx;
}

// Per-channel i-vectors of a stream, the C++ twin of kaldi_amd/online.py: the extractor sees every feature frame as soon as it exists; Latest() is the i-vector
// the reference's online
// decodable hands the network for a chunk (nnet3/decodable-online-looped.cc:182-197 over OnlineIvectorFeature with use_most_recent_ivector): the estimate made at the last multiple
// of --ivector-period among the frames ready (all frames so far minus the LDA splice's right context while the stream goes on), zero before the first.  One k3_ivector_stream per
// channel carries the CMVN window, the splice context, the statistics and the solver's start between chunks: the estimates are, bit for bit, rows of the whole-utterance extraction
// (k3_ivector_extract_batch: row k = statistics of frames 0 .. k * period), every frame processed once.
class OnlineIvectors {
 public:
  OnlineIvectors(k3_ivector *iv, int right_context, int nch, hipStream_t stream = nullptr) : iv_(iv), st_(nch, nullptr), stream_(stream) {
    (void)right_context;      // (the extractor's own option; kept in the signature for its callers)
    k3_ivector_info i;
    K3H_CHECK_K3(k3_ivector_get_info(iv, &i));
    F_ = i.feat_dim;
    R_ = i.ivector_dim;
    latest_.need((size_t)nch * R_);
    K3O_HIP(hipMemsetAsync(latest_.p, 0, (size_t)nch * R_ * 4, stream_));
    for (auto &s : st_) K3H_CHECK_K3(k3_ivector_stream_create(iv_, &s));
  }
  int Dim() const { return R_; }
  void Reset(int ch) { K3H_CHECK_K3(k3_ivector_stream_reset(st_[ch], stream_)); K3O_HIP(hipMemsetAsync(latest_.p + (size_t)ch * R_, 0, (size_t)R_ * 4, stream_)); }
  // n new feature rows of the channel (device, F_ wide, contiguous); finished: the stream's audio has ended.  Updates Row(ch).
  void Accept(int ch, const float *d_rows, int n, bool finished) {
    // queued on the work stream, like its consumers
    K3H_CHECK_K3(k3_ivector_stream_accept(st_[ch], d_rows, F_, n, finished ? 1 : 0, nullptr, 0, 0, nullptr, latest_.p + (size_t)ch * R_, stream_));
  }
  // the same for the channels of one batch, one launch per stage (k3_ivector_stream_accept_batch): channel channels[i] takes nf[i] rows of d_rows (back to
  // back, F_ wide); first[i]: a
  // new stream takes the channel
  template <class B1, class B2> void AcceptBatch(const std::vector<int> &channels, const float *d_rows, const std::vector<int> &nf, const B1 &first, const B2 &last) {
    const size_t n = channels.size(); if (n == 0) return;
    std::vector<k3_ivector_stream *> st(n); std::vector<int64_t> off(n + 1, 0); std::vector<int32_t> fin(n);
    for (size_t i = 0; i < n; i++) { if (first[i]) Reset(channels[i]); st[i] = st_[channels[i]]; off[i + 1] = off[i] + nf[i]; fin[i] = last[i] ? 1 : 0; }
    float *tmp = batch_.need(n * R_);
    K3H_CHECK_K3(k3_ivector_stream_accept_batch(st.data(), (int32_t)n, d_rows, F_, off.data(), fin.data(), tmp, R_, stream_));
    for (size_t i = 0; i < n; i++) K3O_HIP(hipMemcpyAsync(latest_.p + (size_t)channels[i] * R_, tmp + i * R_, (size_t)R_ * 4, hipMemcpyDeviceToDevice, stream_));
  }
  const float *Row(int ch) const { return latest_.p + (size_t)ch * R_; }
  // the rows of the listed channels back to back (what StaticNnet3::Pass takes)
  const float *Gather(const std::vector<int> &channels) {
    float *g = gather_.need(std::max<size_t>(channels.size(), 1) * R_);
    for (size_t i = 0; i < channels.size(); i++) K3O_HIP(hipMemcpyAsync(g + i * R_, Row(channels[i]), (size_t)R_ * 4, hipMemcpyDeviceToDevice, stream_));
    return g;
  }
  ~OnlineIvectors() { for (auto *s : st_) if (s) k3_ivector_stream_destroy(s); }
 private:
  k3_ivector *iv_; int F_ = 0, R_ = 0; std::vector<k3_ivector_stream *> st_; hipStream_t stream_ = nullptr; DevBuf<float> latest_, gather_, batch_;
};
}  // namespace k3host
