// k3_online_pipeline.h -- the class surface of the reference's streaming CUDA pipeline over the C ABI:
//   kaldi::cuda_decoder::BatchedThreadedNnet3CudaOnlinePipeline (cudadecoder/batched-threaded-nnet3-cuda-online-pipeline.h:119-330): correlation ids that
//   claim channels (TryInitCorrID), DecodeBatch(corr_ids, wave_samples, is_first_chunk, is_last_chunk[, partial_hypotheses, end_point]),
//   SetLatticeCallback / SetBestPathCallback per correlation id, WaitForLatticeCallbacks, GetNSampsPerChunk / GetModelFrequency.
// Same host-layer types as k3_pipeline.h (OpenFst / Kaldi's libraries cannot be linked here).  Behind it: the chunked drivers of k3_online.h (per-channel
// sample stash, per-channel input context of the network planned once, AdvanceDecoding per chunk) -- every chunk's outputs are bit-identical to the
// offline batch --, k3_decoder_get_best_path for partial hypotheses / end-pointing, and a worker pool that determinizes the lattice of a
// stream that ended and runs its callback (:735-810).  Not provided: segmentation results, the lattice postprocessor, i-vectors per chunk.
#pragma once
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include "../kaldi_amd/host/k3_online.h"      // (the host layer of this repository: k3_host.h + the streaming helpers)
namespace k3host {
namespace cuda_decoder {

struct BatchedThreadedNnet3CudaOnlinePipelineConfig {      // cudadecoder/batched-threaded-nnet3-cuda-online-pipeline.h:50-117, the options this pipeline reads
  int32_t max_batch_size = 400, num_channels = -1, num_worker_threads = -1;
  bool determinize_lattice = true;
  DeterminizeLatticePhonePrunedOptions det_opts;
  k3_feat_opts feature_opts; k3_decoder_config decoder_opts;
  float acoustic_scale = 0.1f; int32_t frame_subsampling_factor = 1, frames_per_chunk = 50, max_utterance_frames = 6000;
  // end-pointing (online2/online-endpoint.h rule 1-style: trailing silence is not known without a silence-phone list; the rule on the final relative cost is applied)
  float endpoint_max_relative_cost = 2.0f; int32_t endpoint_min_frames = 30;
  std::string ivector_extraction_config;      // OnlineNnet2FeaturePipelineConfig::ivector_extraction_config (models with an i-vector input)
  BatchedThreadedNnet3CudaOnlinePipelineConfig() { memset(&feature_opts, 0, sizeof feature_opts); k3_decoder_config_default(&decoder_opts); }
};

class BatchedThreadedNnet3CudaOnlinePipeline {
 public:
  using CorrelationID = uint64_t;
  typedef std::function<void(const std::string &, bool, bool)> BestPathCallback;      // (word ids of the best path, is partial, endpoint detected)
  typedef std::function<void(CompactLattice &)> LatticeCallback;
  BatchedThreadedNnet3CudaOnlinePipeline(const BatchedThreadedNnet3CudaOnlinePipelineConfig &config, const HostFst &decode_fst, k3_nnet *am_nnet, const TransitionInfo &trans_model)
      : config_(config), trans_(trans_model) {
    nch_ = std::max(config_.num_channels, config_.max_batch_size);
    K3H_CHECK_K3(k3_feat_plan_create(&config_.feature_opts, &plan_)); fdim_ = k3_feat_dim(plan_);
    k3_nnet_info ni; K3H_CHECK_K3(k3_nnet_get_info(am_nnet, &ni)); N_ = ni.output_dim;
    if (ni.input_dim != fdim_) K3H_ERR << "Feature dimension " << fdim_ << " does not match the model's input dimension " << ni.input_dim;
    if (!config_.ivector_extraction_config.empty()) { iv_info_ = ReadIvectorExtractionConfig(config_.ivector_extraction_config); ivx_ = CreateIvectorExtractor(iv_info_, fdim_); }
    if ((ni.ivector_dim > 0) != (ivx_ != nullptr) || (ivx_ && ni.ivector_dim != iv_info_.ie.ivector_dim))
      K3H_ERR << "Neural net expects 'ivector' features with dimension " << ni.ivector_dim << " but you provided " << (ivx_ ? iv_info_.ie.ivector_dim : 0);
    if (N_ != trans_.num_pdfs) K3H_ERR << "Model output dimension " << N_ << " != number of pdfs in the transition model " << trans_.num_pdfs;
    std::vector<float> lp; if (ni.has_priors) { lp.resize(N_); K3H_CHECK_K3(k3_nnet_get_priors(am_nnet, lp.data())); for (float &p : lp) p = logf(p); }
    K3H_CHECK_K3(k3_fst_create(decode_fst.NumStates(), decode_fst.start, decode_fst.arc_offsets.data(), decode_fst.ilabel.data(), decode_fst.olabel.data(),
        decode_fst.weight.data(),
                               decode_fst.nextstate.data(), decode_fst.final_cost.data(), trans_.id2pdf.data(), (int32_t)trans_.id2pdf.size(), &fst_));
    graph_start_ = k3_fst_start(fst_);
    K3H_CHECK_K3(k3_decoder_create(fst_, &config_.decoder_opts, nch_, N_, &dec_));
    K3H_CHECK_K3(k3_decoder_init_decoding(dec_, nch_, config_.max_utterance_frames, nullptr));
    const int s = config_.frame_subsampling_factor; C_ = std::max(s, config_.frames_per_chunk / s * s);
    // one work stream for everything DecodeBatch queues, fed from page-locked staging rings: the host does not wait for the device inside a call (k3_online.h:
    // DevBuf::upload_async)
    K3O_HIP(hipStreamCreate(&ws_));
    // token passing on its own stream: a chunk's launch lasts as long as its slowest lane, the next pass's features / network run beside it; the two streams share only the
    // gathered log-likelihood block (two of them, events ev_ll_: filled / ev_tp_: consumed)
    // (the critical path of a round)
    {
      int lo = 0, hi = 0;
      K3O_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      if (lo == hi) K3O_HIP(hipStreamCreate(&ds_));
      else K3O_HIP(hipStreamCreateWithPriority(&ds_, hipStreamDefault, hi));
    }
    for (int k = 0; k < 2; k++) { K3O_HIP(hipEventCreateWithFlags(&ev_ll_[k], hipEventDisableTiming)); K3O_HIP(hipEventCreateWithFlags(&ev_tp_[k], hipEventDisableTiming)); }
    features_.reset(new OnlineFeatures(plan_, config_.feature_opts, nch_, ws_));
    net_.reset(new StaticNnet3(am_nnet, nch_, nch_, C_, s, lp.empty() ? nullptr : lp.data(), config_.acoustic_scale, ws_));
    if (ivx_) ivs_.reset(new OnlineIvectors(ivx_, iv_info_.right_context, nch_, ws_));
    samples_per_chunk_ = C_ * (int)(config_.feature_opts.samp_freq * 0.001 * config_.feature_opts.frame_shift_ms);
    // (pending feature rows of all channels, compact; see DecodeBatch)
    pend_cap_ = (size_t)(2 * C_ + 8);
    for (auto &h : held_) h.need((size_t)nch_ * (pend_cap_ + (size_t)C_ + 16) * fdim_);
    chan_.resize(nch_); for (int c = nch_ - 1; c >= 0; c--) free_.push_back(c);
    const int nw = config_.num_worker_threads > 0 ? config_.num_worker_threads : std::max(1, (int)std::thread::hardware_concurrency());
    for (int i = 0; i < nw; i++) workers_.emplace_back([this] { WorkerLoop(); });
  }
  ~BatchedThreadedNnet3CudaOnlinePipeline() {
    WaitForLatticeCallbacks();
    { std::lock_guard<std::mutex> l(m_); stop_ = true; } wcv_.notify_all();
    for (auto &w : workers_) w.join();
    ivs_.reset(); if (ivx_) k3_ivector_destroy(ivx_);
    net_.reset();
    features_.reset();
    if (ws_) {
      (void)hipStreamSynchronize(ws_);
      (void)hipStreamDestroy(ws_);
    }
    if (ds_) {
      (void)hipStreamSynchronize(ds_);
      (void)hipStreamDestroy(ds_);
    }
    for (int k = 0; k < 2; k++) {
      if (ev_ll_[k]) (void)hipEventDestroy(ev_ll_[k]);
      if (ev_tp_[k]) (void)hipEventDestroy(ev_tp_[k]);
    }
    k3_decoder_destroy(dec_);
    k3_fst_destroy(fst_);
    k3_feat_plan_destroy(plan_);
  }
  int32_t GetNSampsPerChunk() const { return samples_per_chunk_; }
  int32_t GetNInputFramesPerChunk() const { return C_; }
  float GetModelFrequency() const { return config_.feature_opts.samp_freq; }
  // Claims a channel for a new stream (:160-168); false when every channel is busy (the reference can also wait `wait_for` microseconds: the caller retries)
  bool TryInitCorrID(CorrelationID corr_id, int = 0) {
    std::lock_guard<std::mutex> l(m_);
    if (corr2chan_.count(corr_id)) return true;
    if (free_.empty()) return false;
    const int c = free_.back(); free_.pop_back(); corr2chan_[corr_id] = c; chan_[c] = Chan(); return true;
  }
  void SetLatticeCallback(CorrelationID corr_id, const LatticeCallback &cb) { std::lock_guard<std::mutex> l(m_); lat_cb_[corr_id] = cb; }
  void SetBestPathCallback(CorrelationID corr_id, const BestPathCallback &cb) { std::lock_guard<std::mutex> l(m_); best_cb_[corr_id] = cb; }
  // One chunk of audio for each listed stream (at most max_batch_size of them, each at most GetNSampsPerChunk() samples unless it is the stream's last chunk).
  // partial_hypotheses / end_point (optional): the current best path's word ids and the end-point verdict of every stream of the batch.
  void DecodeBatch(const std::vector<CorrelationID> &corr_ids, const std::vector<std::vector<float>> &wave_samples, const std::vector<bool> &is_first_chunk,
      const std::vector<bool> &is_last_chunk,
                   std::vector<std::string> *partial_hypotheses = nullptr, std::vector<bool> *end_point = nullptr) {
    const size_t n = corr_ids.size();
    if (n > (size_t)config_.max_batch_size || wave_samples.size() != n || is_first_chunk.size() != n || is_last_chunk.size() != n) K3H_ERR << "DecodeBatch: bad batch";
    std::vector<int> chs(n); std::vector<char> first(n), last(n);
    for (size_t i = 0; i < n; i++) {
      if (is_first_chunk[i] && !TryInitCorrID(corr_ids[i])) K3H_ERR << "DecodeBatch: no free channel for a new stream (TryInitCorrID first)";
      std::lock_guard<std::mutex> l(m_); auto it = corr2chan_.find(corr_ids[i]); if (it == corr2chan_.end()) K3H_ERR << "DecodeBatch: unknown correlation id " << corr_ids[i];
      chs[i] = it->second; first[i] = is_first_chunk[i]; last[i] = is_last_chunk[i];
    }
    std::vector<int32_t> fresh; for (size_t i = 0; i < n; i++) if (first[i]) { fresh.push_back(chs[i]); net_->Reset(chs[i]); chan_[chs[i]] = Chan(); }
    if (!fresh.empty()) K3H_CHECK_K3(k3_decoder_init_channels(dec_, fresh.data(), (int32_t)fresh.size(), ds_));
    float *d_feats = nullptr; const std::vector<int> nf = features_->ComputeFeaturesBatched(chs, wave_samples, first, &d_feats);
    // Feature rows a channel has computed but not yet fed to the network live compactly in one device buffer, channel after channel, (offset, count) on the
    // host, at most two segments
    // per channel (leftover + this call's rows).  A pass takes its rows with one row gather and the leftovers of all channels move to the other buffer with one more, once per call:
    // per-channel buffers cost ~4 synchronous device copies per channel and call (40 k copies in a 512-channel run of 10 s files, half of its GPU time).
    { int64_t off = 0, tot = 0; for (int k : nf) tot += k;
      if ((size_t)(held_rows_ + tot) * fdim_ > held_[held_cur_].cap) K3H_ERR << "DecodeBatch: pending-frame buffer exceeded";
      if (tot > 0) K3O_HIP(hipMemcpyAsync(held_[held_cur_].p + (size_t)held_rows_ * fdim_, d_feats, (size_t)tot * fdim_ * 4, hipMemcpyDeviceToDevice, ws_));
      for (size_t i = 0; i < n; i++) { Chan &c = chan_[chs[i]]; if ((size_t)(c.pend + nf[i]) > pend_cap_) K3H_ERR << "DecodeBatch: a chunk longer than GetNSampsPerChunk() samples";
        if (c.seg_cnt[1] != 0) K3H_ERR << "DecodeBatch: internal: pending rows not compacted";
        c.seg_off[1] = held_rows_ + off; c.seg_cnt[1] = nf[i]; c.pend += nf[i]; c.frames += nf[i]; off += nf[i]; } }
    if (ivs_) ivs_->AcceptBatch(chs, d_feats, nf, first, last);      // the extractor sees every frame as soon as it exists: all channels of the batch in one launch per stage
    std::vector<char> is_last(nch_, 0), closed(nch_, 0); for (size_t i = 0; i < n; i++) is_last[chs[i]] = last[i];
    bool need_advance = !fresh.empty();
    while (true) {      // network passes of frames_per_chunk frames per channel until every stream of the batch is drained to less than a chunk (or flushed, at its end)
      std::vector<int> run, n_new; std::vector<char> lasts;
      for (int ch : chs) if (!closed[ch] && (chan_[ch].pend >= C_ || is_last[ch])) run.push_back(ch);
      if (run.empty() && !need_advance) break;
      int64_t tot_new = 0; for (int ch : run) { const int k = std::min(C_, chan_[ch].pend); n_new.push_back(k); tot_new += k; }
      new_.need((size_t)std::max<int64_t>(tot_new, 1) * fdim_);
      { std::vector<int32_t> take; take.reserve((size_t)tot_new);
        for (size_t i = 0; i < run.size(); i++) {
          Chan &c = chan_[run[i]]; int k = n_new[i];
          for (int sgm = 0; sgm < 2 && k > 0; sgm++) {
            const int m = std::min(k, c.seg_cnt[sgm]);
            for (int j = 0; j < m; j++) take.push_back((int32_t)(c.seg_off[sgm] + j));
            c.seg_off[sgm] += m;
            c.seg_cnt[sgm] -= m;
            k -= m;
            c.pend -= m;
          }
          const bool end = is_last[run[i]] && c.pend == 0; lasts.push_back(end); if (end) closed[run[i]] = 1;
        }
        if (!take.empty()) { gidx_.upload_async(take, ws_); K3H_CHECK_K3(k3_mat_copy_rows(new_.p, fdim_, (int32_t)take.size(), fdim_, held_[held_cur_].p, fdim_, gidx_.p, ws_)); } }
      // the pass's log-likelihoods are decoded where the network left them (k3_decoder_advance_decoding_strided): two network output buffers in turn, pass k + 1 queued behind the
      // token-passing launch that read its buffer two passes ago (ev_tp_), the launch behind the pass that fills it (ev_ll_)
      const int lb = (int)(pass_no_++ & 1);
      std::vector<const float *> lane_first(nch_, nullptr); std::vector<int32_t> lane_frames(nch_, 0); int64_t ld_rows = N_;
      if (!run.empty()) {
        if (tp_used_[lb]) K3O_HIP(hipStreamWaitEvent(ws_, ev_tp_[lb], 0));
        net_->SelectOut(lb);
        auto res = net_->Pass(run, new_.p, n_new, lasts, ivs_ ? ivs_->Gather(run) : nullptr);
        for (size_t i = 0; i < run.size(); i++) if (res[i].count > 0) {
          lane_first[run[i]] = net_->Out() + (size_t)res[i].first * N_;
          lane_frames[run[i]] = res[i].count;
          ld_rows = (int64_t)res[i].stride * N_;
        }
      }
      K3O_HIP(hipEventRecord(ev_ll_[lb], ws_)); K3O_HIP(hipStreamWaitEvent(ds_, ev_ll_[lb], 0));
      K3H_CHECK_K3(k3_decoder_advance_decoding_strided(dec_, nch_, lane_first.data(), lane_frames.data(), ld_rows, ds_));
      K3O_HIP(hipEventRecord(ev_tp_[lb], ds_)); tp_used_[lb] = true;
      need_advance = false;
      for (int ch : run) if (closed[ch] && net_->Pending(ch)) closed[ch] = 0;
    }
    {      // the rows still waiting, of ALL channels (also those that were not in this batch), into the other buffer; every channel is back to one segment
      std::vector<int32_t> keep; int64_t at = 0;
      for (int ch = 0; ch < nch_; ch++) {
        Chan &c = chan_[ch]; const int k = c.seg_cnt[0] + c.seg_cnt[1];
        for (int sgm = 0; sgm < 2; sgm++) for (int j = 0; j < c.seg_cnt[sgm]; j++) keep.push_back((int32_t)(c.seg_off[sgm] + j));
        c.seg_off[0] = at; c.seg_cnt[0] = k; c.seg_off[1] = 0; c.seg_cnt[1] = 0; at += k;
      }
      if (!keep.empty()) {
        gidx_.upload_async(keep, ws_);
        K3H_CHECK_K3(k3_mat_copy_rows(held_[held_cur_ ^ 1].p, fdim_, (int32_t)keep.size(), fdim_, held_[held_cur_].p, fdim_, gidx_.p, ws_));
      }
      held_cur_ ^= 1; held_rows_ = at;
    }
    // partial hypotheses / end-pointing / best-path callbacks (cuda-decoder.cc:1864-2003) from the tokens the channels hold now
    bool want_best = partial_hypotheses || end_point;
    {
      std::lock_guard<std::mutex> l(m_);
      for (size_t i = 0; i < n && !want_best; i++) want_best = best_cb_.count(corr_ids[i]) > 0;
    }
    if (want_best && n > 0) {
      std::vector<int32_t> c32(chs.begin(), chs.end()); std::vector<int64_t> off(n + 1); const int64_t cap = (int64_t)n * (config_.max_utterance_frames + 16);
      std::vector<int32_t> il(cap), ol(cap); std::vector<float> g(cap), a(cap), fc(n), rc(n); std::vector<int32_t> rf(n);
      K3H_CHECK_K3(k3_decoder_get_best_path(dec_, c32.data(), (int32_t)n, 0, off.data(), cap, il.data(), ol.data(), g.data(), a.data(), fc.data(), rc.data(), rf.data()));
      if (partial_hypotheses) partial_hypotheses->assign(n, std::string()); if (end_point) end_point->assign(n, false);
      for (size_t i = 0; i < n; i++) {
        std::string words; for (int64_t k = off[i]; k < off[i + 1]; k++) if (ol[k] != 0) { if (!words.empty()) words += ' '; words += std::to_string(ol[k]); }
        const bool ep = chan_[chs[i]].frames >= config_.endpoint_min_frames && std::isfinite(rc[i]) && rc[i] <= config_.endpoint_max_relative_cost;
        if (partial_hypotheses) (*partial_hypotheses)[i] = words; if (end_point) (*end_point)[i] = ep;
        BestPathCallback cb; { std::lock_guard<std::mutex> l(m_); auto it = best_cb_.find(corr_ids[i]); if (it != best_cb_.end()) cb = it->second; }
        if (cb) cb(words, !last[i], ep);
      }
    }
    // streams that ended: finalise their channels, hand the raw lattices to the workers, free the channels (:560-640)
    std::vector<int32_t> ended; std::vector<CorrelationID> ended_ids; for (size_t i = 0; i < n; i++) if (last[i]) { ended.push_back(chs[i]); ended_ids.push_back(corr_ids[i]); }
    if (ended.empty()) return;
    K3H_CHECK_K3(k3_decoder_finalize_channels(dec_, ended.data(), (int32_t)ended.size(), ds_));
    const int U = (int)ended.size(); std::vector<int64_t> info(10 * (size_t)U); K3H_LATTICE_INFO(dec_, info.data());
    int64_t NS = 0, NA = 0; for (int u = 0; u < U; u++) { NS += info[10 * u]; NA += info[10 * u + 1]; }
    std::vector<int32_t> sf(NS + 1), ss(NS + 1), as(NA + 1), ad(NA + 1), ai(NA + 1), ao(NA + 1); std::vector<float> sc(NS + 1), sfin(NS + 1), ag(NA + 1), aa(NA + 1);
    if (NS > 0) K3H_CHECK_K3(k3_decoder_get_raw_lattices(dec_, sf.data(), ss.data(), sc.data(), sfin.data(), as.data(), ad.data(), ai.data(), ao.data(), ag.data(), aa.data()));
    int64_t s0 = 0, a0 = 0;
    for (int u = 0; u < U; u++) {
      const int64_t ns = info[10 * u], na = info[10 * u + 1]; auto t = std::make_shared<Task>();
      if (info[10 * u + 2] == 0 && ns > 0) {
        Lattice &lat = t->raw;
        lat.st_frame.assign(sf.begin() + s0, sf.begin() + s0 + ns);
        lat.st_state.assign(ss.begin() + s0, ss.begin() + s0 + ns);
        lat.st_final.assign(sfin.begin() + s0, sfin.begin() + s0 + ns);
        lat.arc_src.assign(as.begin() + a0, as.begin() + a0 + na);
        lat.arc_dst.assign(ad.begin() + a0, ad.begin() + a0 + na);
        lat.arc_ilabel.assign(ai.begin() + a0, ai.begin() + a0 + na);
        lat.arc_olabel.assign(ao.begin() + a0, ao.begin() + a0 + na);
        lat.arc_graph.assign(ag.begin() + a0, ag.begin() + a0 + na);
        lat.arc_ac.assign(aa.begin() + a0, aa.begin() + a0 + na);
        for (int64_t s = 0; s < ns; s++) if (lat.st_frame[s] == 0 && lat.st_state[s] == graph_start_) lat.start = (int32_t)s;
      }
      s0 += ns; a0 += na;
      { std::lock_guard<std::mutex> l(m_);
        auto it = lat_cb_.find(ended_ids[u]); if (it != lat_cb_.end()) { t->callback = it->second; lat_cb_.erase(it); }
        best_cb_.erase(ended_ids[u]); corr2chan_.erase(ended_ids[u]); free_.push_back(ended[u]);
        if (t->callback) { n_callbacks_not_done_++; post_.push_back(t); } }
    }
    wcv_.notify_all();
  }
  void WaitForLatticeCallbacks() noexcept { std::unique_lock<std::mutex> l(m_); done_cv_.wait(l, [&] { return n_callbacks_not_done_ == 0; }); }

 private:
  struct Chan { int pend = 0; int64_t frames = 0; int64_t seg_off[2] = {0, 0}; int seg_cnt[2] = {0, 0}; };      // seg: where the channel's pending rows lie in held_[held_cur_]
  struct Task { Lattice raw; LatticeCallback callback; };
  void WorkerLoop() {
    for (;;) {
      std::shared_ptr<Task> t;
      { std::unique_lock<std::mutex> l(m_); wcv_.wait(l, [&] { return stop_ || !post_.empty(); }); if (post_.empty()) return; t = post_.front(); post_.pop_front(); }
      CompactLattice clat;
      try {
        if (t->raw.NumStates() > 0) {
          Connect(&t->raw);
          if (config_.determinize_lattice) DeterminizeLatticePhonePruned(t->raw, trans_, config_.decoder_opts.lattice_beam, &clat, config_.det_opts);
          else ConvertLattice(t->raw, &clat);
        }
        t->callback(clat);
      } catch (const std::exception &e) { K3H_WARN << "lattice post-processing / callback failed: " << e.what(); }
      { std::lock_guard<std::mutex> l(m_); n_callbacks_not_done_--; } done_cv_.notify_all();
    }
  }
  const BatchedThreadedNnet3CudaOnlinePipelineConfig config_; const TransitionInfo &trans_;
  hipStream_t ws_ = nullptr, ds_ = nullptr;
  hipEvent_t ev_ll_[2] = {nullptr, nullptr}, ev_tp_[2] = {nullptr, nullptr};
  bool tp_used_[2] = {false, false};
  unsigned pass_no_ = 0;
  k3_feat_plan *plan_ = nullptr;
  k3_fst *fst_ = nullptr;
  k3_decoder *dec_ = nullptr;
  int nch_ = 0, fdim_ = 0, N_ = 0, C_ = 0, samples_per_chunk_ = 0;
  int32_t graph_start_ = 0;
  size_t pend_cap_ = 0;
  std::unique_ptr<OnlineFeatures> features_; std::unique_ptr<StaticNnet3> net_; std::unique_ptr<OnlineIvectors> ivs_; k3_ivector *ivx_ = nullptr; IvectorExtractionInfo iv_info_;
  std::vector<Chan> chan_; DevBuf<float> held_[2], new_; DevBuf<int32_t> gidx_; int held_cur_ = 0; int64_t held_rows_ = 0;
  std::mutex m_; std::condition_variable wcv_, done_cv_; bool stop_ = false; int n_callbacks_not_done_ = 0;
  std::map<CorrelationID, int> corr2chan_; std::vector<int> free_; std::map<CorrelationID, LatticeCallback> lat_cb_; std::map<CorrelationID, BestPathCallback> best_cb_;
  std::deque<std::shared_ptr<Task>> post_; std::vector<std::thread> workers_;
};
}  // namespace cuda_decoder
}  // namespace k3host
