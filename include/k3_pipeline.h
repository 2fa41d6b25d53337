// k3_pipeline.h -- the class surface of the reference's offline CUDA pipeline over the C ABI:
//   kaldi::cuda_decoder::BatchedThreadedNnet3CudaPipeline2 (cudadecoder/batched-threaded-nnet3-cuda-pipeline2.h:57-239): the constructor taking
//   (config, decoding graph, acoustic model, transition model), DecodeWithCallback (waveform in, CompactLattice handed to a callback on a worker
//   thread), task groups (CreateTaskGroup / DestroyTaskGroup / WaitForGroup) and WaitForAllTasks, with the reference's names and semantics.
// Types are this host layer's own (k3_host.h: HostFst, TransitionInfo, Wave, CompactLattice) because OpenFst and Kaldi's libraries cannot be
// linked here; INTEGRATION.md shows the one-to-one mapping.  Behind it: a control thread that takes whatever is queued (up to max_batch_size
// utterances, like AcquireTasks :349-380), runs waveforms -> k3_feat_compute_batch -> k3_nnet_forward -> k3_decoder_decode_batch -> raw
// lattices, and a pool of worker threads that does Connect + (phone-)pruned determinization per utterance and calls the callback
// (batched-threaded-nnet3-cuda-online-pipeline.cc:735-810).  SegmentedDecodeWithCallback (:265-337) cuts a waveform into overlapping segments, decodes each as
// an utterance and hands all results to one callback (CudaPipelineResult, lattice results).  The lattice postprocessor (CTM results) is not provided:
// asking for RESULT_TYPE_CTM is an error.
#pragma once
#include <limits>
#include <ostream>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <future>
#include <thread>
#include "../kaldi_amd/host/k3_online.h"      // (the host layer of this repository: k3_host.h + the streaming helpers)
namespace k3host {
namespace cuda_decoder {

// cudadecoder/cuda-pipeline-common.h:36-60
inline int NumberOfSegments(int nsamples, int seg_length, int seg_shift) {
  if (seg_shift <= 0 || seg_length < seg_shift) K3H_ERR << "NumberOfSegments: bad segment length / shift";
  if (nsamples <= seg_length) return 1;
  return ((nsamples - (seg_length - seg_shift)) + seg_shift - 1) / seg_shift;
}
struct CudaPipelineSegmentationConfig {
  double segment_length_s = 20, segment_overlap_s = 1, min_segment_length_s = 1;
  void Register(ParseOptions *po) {
    po->Register("segment-length", &segment_length_s, "Segment length (s)"); po->Register("segment-overlap", &segment_overlap_s, "Overlap between segments (s)");
    po->Register("min-segment-length", &min_segment_length_s, "Min segment length (s, >=1)");
  }
  void Check() const {
    if (min_segment_length_s < 0.5) K3H_ERR << "Min segment length must be at least 0.5 second";
    if (segment_overlap_s > segment_length_s) K3H_ERR << "The segments overlap cannot be larger than segment length";
    if (segment_length_s < min_segment_length_s) K3H_ERR << "Segment length cannot be smaller than min segment length";
    if (segment_overlap_s >= segment_length_s) K3H_ERR << "The segments overlap must be smaller than the segment length";
  }
};
// cudadecoder/cuda-pipeline-common.h:69-140
class CudaPipelineResult {
  int result_type_ = 0; CompactLattice clat_; CtmResult ctm_; float offset_seconds_ = 0; int32_t segment_id_ = 0; bool is_last_segment_ = false;
 public:
  static constexpr int RESULT_TYPE_LATTICE = 1, RESULT_TYPE_CTM = 2;
  int32_t GetResultType() const { return result_type_; }
  bool HasValidResult() const { return result_type_ != 0; }
  int32_t GetSegmentID() const { return segment_id_; }
  bool IsLastSegment() const { return is_last_segment_; }
  float GetTimeOffsetSeconds() const { return offset_seconds_; }
  void SetLatticeResult(CompactLattice &&clat) { result_type_ |= RESULT_TYPE_LATTICE; clat_ = std::move(clat); }
  CompactLattice *GetLatticeResult() { if (!(result_type_ & RESULT_TYPE_LATTICE)) K3H_ERR << "Lattice result was not requested"; return &clat_; }
  void SetCTMResult(CtmResult &&ctm) { result_type_ |= RESULT_TYPE_CTM; ctm_ = std::move(ctm); }
  CtmResult *GetCTMResult() { if (!(result_type_ & RESULT_TYPE_CTM)) K3H_ERR << "CTM result was not requested"; return &ctm_; }
  void SetTimeOffsetSeconds(float offset_seconds) { if (offset_seconds < 0) K3H_ERR << "negative segment offset"; offset_seconds_ = offset_seconds; }
  void SetSegmentID(int segment_id) { segment_id_ = segment_id; }
  void SetAsLastSegment() { is_last_segment_ = true; }
};
struct SegmentedLatticeCallbackParams { std::vector<CudaPipelineResult> results; };
// cudadecoder/cuda-pipeline-common.cc:67-142: the CTM lines of one utterance from its segments' results -- of two overlapping segments the earlier one keeps
// the words that begin before the
// later one starts, the later one the rest; times shifted by the segments' offsets
inline void MergeSegmentsToCTMOutput(std::vector<CudaPipelineResult> &results, const std::string &key, std::ostream &os,
    const std::vector<std::string> *word_syms = nullptr, bool use_segment_offsets = true) {
  if (results.empty()) { K3H_WARN << "Utterance " << key << " has no results. Skipping"; return; }
  for (CudaPipelineResult &r : results) if (!r.HasValidResult()) { K3H_WARN << "Utterance " << key << " has at least one segment with an error. Skipping"; return; }
  os << std::fixed; os.precision(2);
  float previous_segment_word_end = 0;
  for (size_t i = 0; i < results.size(); i++) {
    bool first_word = true; const bool last = i + 1 == results.size(); const float next_offset = last ? std::numeric_limits<float>::max() : results[i + 1].GetTimeOffsetSeconds();
    CudaPipelineResult &r = results[i]; const float offset = use_segment_offsets ? r.GetTimeOffsetSeconds() : 0; const CtmResult &ctm = *r.GetCTMResult();
    for (size_t w = 0; w < ctm.times_seconds.size(); w++) {
      const float from = offset + ctm.times_seconds[w].first, to = offset + ctm.times_seconds[w].second;
      if (first_word) { if (from >= previous_segment_word_end) first_word = false; else continue; }
      if (!last && from >= next_offset) break;
      previous_segment_word_end = to;
      os << key << " " << r.GetSegmentID() << "  " << from << ' ' << (to - from) << ' ';
      const int32_t id = ctm.words[w];
      if (word_syms && id >= 0 && (size_t)id < word_syms->size() && !(*word_syms)[id].empty()) os << (*word_syms)[id]; else os << id;
      os << ' ' << ctm.conf[w] << '\n';
    }
  }
}
typedef std::function<void(SegmentedLatticeCallbackParams &)> SegmentedResultsCallback;

struct BatchedThreadedNnet3CudaPipeline2Config {      // the options of BatchedThreadedNnet3CudaOnlinePipelineConfig / CudaDecoderConfig this pipeline reads
  int32_t max_batch_size = 400, num_worker_threads = -1;      // --max-batch-size, --cuda-worker-threads (-1: hardware concurrency)
  bool determinize_lattice = true;                            // --determinize-lattice
  DeterminizeLatticePhonePrunedOptions det_opts;
  k3_feat_opts feature_opts;                                  // from --feature-type + its config (FeatOptions::Finish())
  k3_decoder_config decoder_opts;                             // beam, lattice-beam, max-active, capacities, literal_order
  float acoustic_scale = 0.1f; int32_t frame_subsampling_factor = 1;
  CudaPipelineSegmentationConfig seg_opts;                    // --segment-length, --segment-overlap, --min-segment-length
  // two decoder objects (twice the lane pools) used in turn: a batch's token passing starts under the previous batch's pruning kernel and lattice copy
  bool alternate_decoders = true;
  BatchedThreadedNnet3CudaPipeline2Config() { memset(&feature_opts, 0, sizeof feature_opts); k3_decoder_config_default(&decoder_opts); }
};

class BatchedThreadedNnet3CudaPipeline2 {
 public:
  using LatticeCallback = std::function<void(CompactLattice &)>;
  BatchedThreadedNnet3CudaPipeline2(const BatchedThreadedNnet3CudaPipeline2Config &config, const HostFst &decode_fst, k3_nnet *am_nnet, const TransitionInfo &trans_model)
      : config_(config), nnet_(am_nnet), trans_(trans_model) {
    K3H_CHECK_K3(k3_feat_plan_create(&config_.feature_opts, &plan_)); fdim_ = k3_feat_dim(plan_);
    K3H_CHECK_K3(k3_nnet_get_info(nnet_, &ninfo_));
    if (ninfo_.input_dim != fdim_) K3H_ERR << "Feature dimension " << fdim_ << " does not match the model's input dimension " << ninfo_.input_dim;
    if (ninfo_.ivector_dim > 0) K3H_ERR << "this pipeline class does not extract i-vectors (the batched-wav-nnet3-cuda2 program does)";
    if (ninfo_.output_dim != trans_.num_pdfs) K3H_ERR << "Model output dimension " << ninfo_.output_dim << " != number of pdfs in the transition model " << trans_.num_pdfs;
    if (ninfo_.has_priors) { log_priors_.resize(ninfo_.output_dim); K3H_CHECK_K3(k3_nnet_get_priors(nnet_, log_priors_.data())); for (float &p : log_priors_) p = logf(p); }
    K3H_CHECK_K3(k3_fst_create(decode_fst.NumStates(), decode_fst.start, decode_fst.arc_offsets.data(), decode_fst.ilabel.data(), decode_fst.olabel.data(),
        decode_fst.weight.data(),
                               decode_fst.nextstate.data(), decode_fst.final_cost.data(), trans_.id2pdf.data(), (int32_t)trans_.id2pdf.size(), &fst_));
    graph_start_ = k3_fst_start(fst_);
    K3H_CHECK_K3(k3_decoder_create(fst_, &config_.decoder_opts, config_.max_batch_size, ninfo_.output_dim, &dec_));
    if (config_.alternate_decoders) K3H_CHECK_K3(k3_decoder_create(fst_, &config_.decoder_opts, config_.max_batch_size, ninfo_.output_dim, &dec_b_));
    K3O_HIP(hipGetDevice(&device_));
    const int nw = config_.num_worker_threads > 0 ? config_.num_worker_threads : std::max(1, (int)std::thread::hardware_concurrency());
    for (int i = 0; i < nw; i++) workers_.emplace_back([this] { WorkerLoop(); });
    control_ = std::thread([this] { ControlLoop(); });
  }
  virtual ~BatchedThreadedNnet3CudaPipeline2() {
    WaitForAllTasks();
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all(); wcv_.notify_all();
    control_.join(); for (auto &w : workers_) w.join();
    for (auto &c : plan_cache_) k3_nnet_batch_destroy(c.second);
    k3_decoder_destroy(dec_); if (dec_b_) k3_decoder_destroy(dec_b_); k3_fst_destroy(fst_); k3_feat_plan_destroy(plan_);
  }
  float GetModelFrequency() const { return config_.feature_opts.samp_freq; }
  // batched-threaded-nnet3-cuda-pipeline2.h:204-208: scales / word insertion penalty / MBR applied to every lattice result; required for RESULT_TYPE_CTM
  void SetLatticePostprocessor(const std::shared_ptr<LatticePostprocessor> &lattice_postprocessor) {
    lattice_postprocessor_ = lattice_postprocessor;
    lattice_postprocessor_->SetDecoderFrameShift(config_.feature_opts.frame_shift_ms * 1.0e-3f * config_.frame_subsampling_factor);
    lattice_postprocessor_->SetTransitionInformation(&trans_);      // (batched-threaded-nnet3-cuda-pipeline2.h:206: SetTransitionModel)
  }
  // Enqueues one utterance; `callback` is called with its lattice from a worker thread ("will be called once the lattice is ready", :118-160).
  // An utterance that cannot be decoded (too short, decoder failure) still gets its callback, with an empty lattice (NumStates() == 0).
  void DecodeWithCallback(const std::vector<float> &wave_data, float sample_rate, const LatticeCallback &callback, const std::string &group = std::string()) {
    if (sample_rate != GetModelFrequency()) K3H_ERR << "DecodeWithCallback: sample rate " << sample_rate << " != model frequency " << GetModelFrequency();
    auto t = std::make_shared<Task>(); t->samples = wave_data; t->callback = callback; Enqueue(t, group);
  }
  void DecodeWithCallback(const std::shared_ptr<Wave> &wave_data, const LatticeCallback &callback, const std::string &group = std::string()) {
    DecodeWithCallback(wave_data->samples, wave_data->samp_freq, callback, group);
  }
  // Extracts segments from wave_data and decodes them; `segmented_callback` gets the results of all segments at once, in segment order, from the worker thread that
  // finishes the last of them (:160-168, 265-337).  A waveform shorter than one segment is one segment; a last piece below min-segment-length is dropped.
  void SegmentedDecodeWithCallback(const std::shared_ptr<Wave> &wave_data, const SegmentedResultsCallback &segmented_callback,
      const int result_type = CudaPipelineResult::RESULT_TYPE_LATTICE) {
    if (!result_type) K3H_ERR << "You must define at least one result type";
    if ((result_type & CudaPipelineResult::RESULT_TYPE_CTM) && !lattice_postprocessor_) K3H_ERR <<
        "A lattice postprocessor must be set with SetLatticePostprocessor() to use RESULT_TYPE_CTM";
    if (wave_data->samp_freq != GetModelFrequency()) K3H_ERR << "SegmentedDecodeWithCallback: sample rate " << wave_data->samp_freq << " != model frequency "
        << GetModelFrequency();
    config_.seg_opts.Check();
    const float freq = GetModelFrequency();
    const int seg_len = (int)(config_.seg_opts.segment_length_s * freq), seg_shift = (int)((config_.seg_opts.segment_length_s - config_.seg_opts.segment_overlap_s) * freq),
              seg_min = (int)(config_.seg_opts.min_segment_length_s * freq), total = (int)wave_data->samples.size();
    if (total == 0) {
      if (segmented_callback) {
        SegmentedLatticeCallbackParams params;
        params.results.resize(1);
        params.results[0].SetLatticeResult(CompactLattice());
        params.results[0].SetAsLastSegment();
        segmented_callback(params);
      }
      return;
    }
    std::vector<std::pair<int, int>> pieces;      // (offset, samples)
    for (int offset = 0;; offset += seg_shift) { const int n = std::min(total - offset, seg_len); if (n >= seg_min) pieces.push_back({offset, n}); if (offset + n >= total) break; }
    if (pieces.empty()) { if (segmented_callback) { SegmentedLatticeCallbackParams params; segmented_callback(params); } return; }
    auto results = std::make_shared<std::vector<CudaPipelineResult>>(pieces.size());
    auto not_done = std::make_shared<std::atomic<int32_t>>((int32_t)pieces.size());
    for (size_t i = 0; i < pieces.size(); i++) {
      CudaPipelineResult &r = (*results)[i];
      r.SetTimeOffsetSeconds(std::floor((float)pieces[i].first / freq));
      r.SetSegmentID((int)i);
      if (i + 1 == pieces.size()) r.SetAsLastSegment();
      auto pp = lattice_postprocessor_;
      LatticeCallback callback = [results, not_done, segmented_callback, i, pp, result_type](CompactLattice &clat) {
        // SetResultUsingLattice (cudadecoder/lattice-postprocessor.cc:112-137)
        if (result_type & CudaPipelineResult::RESULT_TYPE_CTM) { CtmResult ctm; CompactLattice copy = clat; pp->GetCTM(copy, &ctm); (*results)[i].SetCTMResult(std::move(ctm)); }
        if (result_type & CudaPipelineResult::RESULT_TYPE_LATTICE) {
          if (pp) {
            CompactLattice out;
            pp->GetPostprocessedLattice(clat, &out);
            (*results)[i].SetLatticeResult(std::move(out));
          } else (*results)[i].SetLatticeResult(std::move(clat));
        }
        if (not_done->fetch_sub(1) == 1 && segmented_callback) { SegmentedLatticeCallbackParams params; params.results = std::move(*results); segmented_callback(params); }
      };
      std::vector<float> piece(wave_data->samples.begin() + pieces[i].first, wave_data->samples.begin() + pieces[i].first + pieces[i].second);
      DecodeWithCallback(piece, freq, callback);
    }
  }
  void CreateTaskGroup(const std::string &group) {
    std::lock_guard<std::mutex> l(m_);
    if (!groups_.emplace(group, 0).second) K3H_ERR << "Group is already in use: " << group;
  }
  void DestroyTaskGroup(const std::string &group) {
    std::unique_lock<std::mutex> l(m_);
    auto it = groups_.find(group); if (it == groups_.end()) K3H_ERR << "Group does not exist: " << group;
    done_cv_.wait(l, [&] { return it->second == 0; }); groups_.erase(it);
  }
  void WaitForGroup(const std::string &group) {
    std::unique_lock<std::mutex> l(m_);
    auto it = groups_.find(group); if (it == groups_.end()) K3H_ERR << "Group does not exist: " << group;
    done_cv_.wait(l, [&] { return it->second == 0; });
  }
  void WaitForAllTasks() { std::unique_lock<std::mutex> l(m_); done_cv_.wait(l, [&] { return n_tasks_not_done_ == 0; }); }

 private:
  struct Task { std::vector<float> samples; LatticeCallback callback; std::string group; bool has_group = false; Lattice raw; bool failed = false; };
  void Enqueue(const std::shared_ptr<Task> &t, const std::string &group) {
    std::lock_guard<std::mutex> l(m_);
    if (!group.empty()) {
      auto it = groups_.find(group);
      if (it == groups_.end()) K3H_ERR << "Group does not exist: " << group;
      it->second++;
      t->group = group;
      t->has_group = true;
    }
    n_tasks_not_done_++; queue_.push_back(t); cv_.notify_one();
  }
  void Finish(const std::shared_ptr<Task> &t) {
    std::lock_guard<std::mutex> l(m_);
    if (t->has_group) groups_[t->group]--;
    n_tasks_not_done_--; done_cv_.notify_all();
  }
  void WorkerLoop() {      // Connect + determinization + callback, one utterance at a time
    for (;;) {
      std::shared_ptr<Task> t;
      { std::unique_lock<std::mutex> l(m_); wcv_.wait(l, [&] { return stop_ || !post_.empty(); }); if (post_.empty()) return; t = post_.front(); post_.pop_front(); }
      CompactLattice clat;
      try {
        if (!t->failed && t->raw.NumStates() > 0) {
          Connect(&t->raw);
          if (config_.determinize_lattice) DeterminizeLatticePhonePruned(t->raw, trans_, config_.decoder_opts.lattice_beam, &clat, config_.det_opts);
          else ConvertLattice(t->raw, &clat);
        }
        t->callback(clat);
      } catch (const std::exception &e) { K3H_WARN << "lattice post-processing / callback failed: " << e.what(); }
      Finish(t);
    }
  }
  // Batches one apart on two streams (as bench.py and batched-wav-nnet3-cuda2 do): while the decoder of batch k runs, whatever has been queued meanwhile (up to max_batch_size
  // utterances) gets its upload, features and network issued behind it, on the front stream, into the other log-likelihood buffer.  Nothing waits for a next batch to fill: with an
  // empty queue the present batch is simply finished.
  struct InFlight { std::vector<std::shared_ptr<Task>> batch; std::vector<int> idx; std::vector<int64_t> ro; int U = 0, buf = 0; bool valid = false; };
  std::vector<std::shared_ptr<Task>> TakeBatch(bool block) {
    std::vector<std::shared_ptr<Task>> batch;
    std::unique_lock<std::mutex> l(m_);
    if (block) cv_.wait(l, [&] { return stop_ || !queue_.empty(); });
    while (!queue_.empty() && (int)batch.size() < config_.max_batch_size) { batch.push_back(queue_.front()); queue_.pop_front(); }
    return batch;
  }
  void Hand(std::vector<std::shared_ptr<Task>> &batch) { { std::lock_guard<std::mutex> l(m_); for (auto &t : batch) post_.push_back(t); } wcv_.notify_all(); }
  void ControlLoop() {
    K3O_HIP(hipSetDevice(device_));
    K3O_HIP(hipStreamCreateWithFlags(&s_front_, hipStreamNonBlocking));
    K3O_HIP(hipStreamCreateWithFlags(&s_dec_, hipStreamNonBlocking));
    K3O_HIP(hipStreamCreateWithFlags(&s_dec_b_, hipStreamNonBlocking));
    for (auto &e : ev_front_) K3O_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : ev_dec_) K3O_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : ev_h2d_) K3O_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // Batches in flight: `prev` (decoded, lattices not fetched yet -- only with two decoder objects), `cur` (front end issued, decoder next), `nxt` (front end
    // issued behind cur's decoder).
    // With one decoder object a batch's lattices are fetched right after the next batch's front end has been queued; with two, one step later, so that the next
    // batch's token passing is
    // already queued on the other object's stream while this batch's pruning kernel, compaction and copy run.  Nothing waits for new work while a decoded batch is unfetched.
    InFlight prev, cur, nxt; int parity = 0; bool prev_decoding = false;
    auto start = [&](InFlight *f, std::vector<std::shared_ptr<Task>> &&batch) {
      f->batch = std::move(batch); f->valid = false; f->buf = parity; parity ^= 1;
      try { FrontEnd(f); } catch (const std::exception &e) { K3H_WARN << "batch failed: " << e.what(); for (auto &t : f->batch) t->failed = true; f->valid = false; }
    };
    auto finish = [&](InFlight *f, bool decoding) {
      if (decoding) { try { Fetch(f); } catch (const std::exception &e) { K3H_WARN << "batch failed: " << e.what(); for (auto &t : f->batch) t->failed = true; } }
      Hand(f->batch); *f = InFlight();
    };
    // Two decoder objects: a decoded batch's lattices are fetched (Fetch blocks until that batch's pruning / output kernels and the copy are through) on a helper thread, so that
    // this loop goes on to QUEUE the next front end at once; it only waits for the fetch of the batch that used the same decoder object two batches ago, right
    // before it launches on that
    // object again.  (Fetched on this thread, every other batch's front end reached the stream 30 ms late: the pruning kernel cannot start beside the other object's resident launch and the
    // loop sat in Fetch while the CUs that launch freed stayed idle -- tools/pipeline_overlap.py on the program's kernel trace.)
    std::future<void> fetching[2];
    auto finish_async = [&](InFlight &&f, bool decoding) {
      const int b = f.buf & 1; auto sp = std::make_shared<InFlight>(std::move(f));
      fetching[b] = std::async(std::launch::async, [this, sp, decoding, &finish]() { (void)hipSetDevice(device_); finish(sp.get(), decoding); });
    };
    auto fetched = [&](int b) { if (fetching[b & 1].valid()) fetching[b & 1].get(); };
    for (;;) {
      if (cur.batch.empty()) {
        // idle: hand the last decoded batch over (after the one before it) before blocking for new work
        if (!prev.batch.empty()) {
          fetched(prev.buf ^ 1);
          finish(&prev, prev_decoding);
        }
        auto b = TakeBatch(true); if (b.empty()) break; start(&cur, std::move(b));
      }
      // the next batch's front end first: nothing it needs waits for this batch's decoder launch (its log-likelihood buffer is guarded on the device, its staging is its own)
      nxt = InFlight(); { auto b = TakeBatch(false); if (!b.empty()) start(&nxt, std::move(b)); }
      bool decoding = false;
      try {
        if (cur.valid) {
          fetched(cur.buf);      // this decoder object's previous batch: its lattices must be out before the object is launched again
          K3O_HIP(hipStreamWaitEvent(DecStream(cur.buf), ev_front_[cur.buf], 0));
          K3H_CHECK_K3(k3_decoder_decode_batch(Dec(cur.buf), cur.U, d_ll_[cur.buf].p, ninfo_.output_dim, cur.ro.data(), DecStream(cur.buf))); decoding = true;
          K3O_HIP(hipEventRecord(ev_dec_[cur.buf], DecStream(cur.buf)));
        }
      } catch (const std::exception &e) {
        K3H_WARN << "batch failed: " << e.what(); for (auto &t : cur.batch) t->failed = true; decoding = false;
      }
      if (nxt.batch.empty()) { auto b = TakeBatch(false); if (!b.empty()) start(&nxt, std::move(b)); }      // (it may have arrived while this thread waited)
      if (dec_b_) {
        if (!prev.batch.empty()) finish_async(std::move(prev), prev_decoding);
        prev = std::move(cur); prev_decoding = decoding;
      } else finish(&cur, decoding);
      cur = std::move(nxt); nxt = InFlight();
    }
    fetched(0); fetched(1);
    if (!prev.batch.empty()) finish(&prev, prev_decoding);
    K3O_HIP(hipStreamSynchronize(s_front_)); K3O_HIP(hipStreamSynchronize(s_dec_)); K3O_HIP(hipStreamSynchronize(s_dec_b_));
    for (auto &e : ev_front_) (void)hipEventDestroy(e); for (auto &e : ev_dec_) (void)hipEventDestroy(e); for (auto &e : ev_h2d_) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(s_front_); (void)hipStreamDestroy(s_dec_); (void)hipStreamDestroy(s_dec_b_);
  }
  void FrontEnd(InFlight *f) {      // upload + features + network of f->batch on the front stream, log-likelihoods into buffer f->buf
    std::vector<std::shared_ptr<Task>> &batch = f->batch;
    std::vector<int> &idx = f->idx; idx.clear(); std::vector<int64_t> woff(1, 0), foff(1, 0); std::vector<int32_t> nframes;
    for (size_t i = 0; i < batch.size(); i++) {
      const int nf = k3_feat_num_frames(plan_, (int64_t)batch[i]->samples.size());
      if (nf == 0) { batch[i]->failed = true; continue; }      // too short to decode
      idx.push_back((int)i); woff.push_back(woff.back() + (int64_t)batch[i]->samples.size()); foff.push_back(foff.back() + nf); nframes.push_back(nf);
    }
    const int U = (int)idx.size(); f->U = U; if (U == 0) return;
    const int64_t tot = foff.back();
    // The batch's samples are gathered into a page-locked staging buffer by a few threads and copied asynchronously on the front stream.  Staging, waveform and offset buffers are
    // double-buffered by f->buf, so this front end is QUEUED while the previous one may not even have started (its kernels wait for CUs the decoder's lanes
    // free): the only host wait
    // is for the copy out of this staging buffer two batches ago.  (A pageable synchronous copy of a 512 x 10 s batch, 328 MB, behind a host wait for the previous front end put this
    // batch's kernels on the stream 30 - 60 ms after the decoder launch they hide behind: tools/pipeline_overlap.py on the program's trace.)
    const int B = f->buf & 1;
    if (h2d_recorded_[B]) K3O_HIP(hipEventSynchronize(ev_h2d_[B]));
    {
      float *stage = h_w_[B].need((size_t)std::max<int64_t>(woff.back(), 1));
      const int nthr = (int)std::min<size_t>(4, (size_t)U); std::vector<std::thread> th;
      auto gather = [&](int t) {
        for (int k = t; k < U; k += nthr) {
          const auto &s_ = batch[idx[k]]->samples;
          if (!s_.empty()) memcpy(stage + woff[k], s_.data(), s_.size() * sizeof(float));
        }
      };
      for (int t = 1; t < nthr; t++) th.emplace_back(gather, t);
      gather(0); for (auto &t : th) t.join();
      K3O_HIP(hipMemcpyAsync(d_w_[B].need((size_t)std::max<int64_t>(woff.back(), 1)), stage, (size_t)woff.back() * sizeof(float), hipMemcpyHostToDevice, s_front_));
      // The offsets travel on the FRONT stream from this slot's page-locked buffer, behind the previous front end that read d_wo_[B] / d_fo_[B] (a synchronous copy
      // on the null stream does not wait for the non-blocking front stream and could overwrite them while batch k - 2's feature kernel is still queued).  ev_h2d_[B], recorded
      // behind all three copies, is what guards the reuse of both staging buffers.
      int64_t *ho = h_off_[B].need(woff.size() + foff.size());
      memcpy(ho, woff.data(), woff.size() * sizeof(int64_t)); memcpy(ho + woff.size(), foff.data(), foff.size() * sizeof(int64_t));
      K3O_HIP(hipMemcpyAsync(d_wo_[B].need(woff.size()), ho, woff.size() * sizeof(int64_t), hipMemcpyHostToDevice, s_front_));
      K3O_HIP(hipMemcpyAsync(d_fo_[B].need(foff.size()), ho + woff.size(), foff.size() * sizeof(int64_t), hipMemcpyHostToDevice, s_front_));
      K3O_HIP(hipEventRecord(ev_h2d_[B], s_front_)); h2d_recorded_[B] = true;
    }
    // (d_f_ and the network's buffers: one set, ordered by the stream)
    K3H_CHECK_K3(k3_feat_compute_batch(plan_, d_w_[B].p, d_wo_[B].p, d_fo_[B].p, U, tot, d_f_.need((size_t)tot * fdim_), fdim_, s_front_));
    k3_nnet_batch *nb = nullptr;
    for (auto &c : plan_cache_) if (c.first == nframes) { nb = c.second; break; }
    if (!nb) {
      K3H_CHECK_K3(k3_nnet_batch_create(nnet_, U, nframes.data(), config_.frame_subsampling_factor, log_priors_.empty() ? nullptr : log_priors_.data(),
          config_.acoustic_scale, &nb));
      plan_cache_.push_back({nframes, nb});
      // (a queued front end may still use the plan)
      if (plan_cache_.size() > 4) {
        K3O_HIP(hipStreamSynchronize(s_front_));
        k3_nnet_batch_destroy(plan_cache_.front().second);
        plan_cache_.erase(plan_cache_.begin());
      }
    }
    f->ro.assign(U + 1, 0); const int64_t rows = k3_nnet_batch_output_rows(nb, f->ro.data());
    // (two decoder objects: the batch before the last may still be decoding from this log-likelihood buffer.  Its last reader is that decoder's token-passing launch; the pruning and
    // output kernels behind it cannot run beside the other decoder's resident launch and would hold this front end back by tens of milliseconds)
    if (Dec(0) != Dec(1)) K3H_CHECK_K3(k3_decoder_stream_wait_token_passing(Dec(f->buf), s_front_));
    else K3O_HIP(hipStreamWaitEvent(s_front_, ev_dec_[f->buf], 0));
    // (growing the buffer frees it: only once that decoder is through)
    if ((size_t)rows * ninfo_.output_dim > d_ll_[f->buf].cap) K3O_HIP(hipEventSynchronize(ev_dec_[f->buf]));
    K3H_CHECK_K3(k3_nnet_forward(nb, d_f_.p, fdim_, d_ll_[f->buf].need((size_t)rows * ninfo_.output_dim), ninfo_.output_dim, s_front_));
    K3O_HIP(hipEventRecord(ev_front_[f->buf], s_front_)); f->valid = true;
  }
  void Fetch(InFlight *f) {      // waits for the decoder of f->batch and turns its raw lattices into the tasks' Lattice objects
    std::vector<std::shared_ptr<Task>> &batch = f->batch; const std::vector<int> &idx = f->idx; const int U = f->U;
    k3_decoder *dec = Dec(f->buf);
    std::vector<int64_t> info(10 * (size_t)U); K3H_LATTICE_INFO(dec, info.data());
    int64_t NS = 0, NA = 0; for (int u = 0; u < U; u++) { NS += info[10 * u]; NA += info[10 * u + 1]; }
    std::vector<int32_t> sf(NS + 1), ss(NS + 1), as(NA + 1), ad(NA + 1), ai(NA + 1), ao(NA + 1); std::vector<float> sc(NS + 1), sfin(NS + 1), ag(NA + 1), aa(NA + 1);
    K3H_CHECK_K3(k3_decoder_get_raw_lattices(dec, sf.data(), ss.data(), sc.data(), sfin.data(), as.data(), ad.data(), ai.data(), ao.data(), ag.data(), aa.data()));
    int64_t s0 = 0, a0 = 0;
    for (int u = 0; u < U; u++) {
      Task &t = *batch[idx[u]]; const int64_t ns = info[10 * u], na = info[10 * u + 1];
      if (info[10 * u + 2] != 0 || ns == 0) t.failed = true;
      else {
        Lattice &lat = t.raw;
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
    }
  }
  std::shared_ptr<LatticePostprocessor> lattice_postprocessor_;
  const BatchedThreadedNnet3CudaPipeline2Config config_; k3_nnet *nnet_; const TransitionInfo &trans_;
  k3_feat_plan *plan_ = nullptr;
  k3_fst *fst_ = nullptr;
  k3_decoder *dec_ = nullptr, *dec_b_ = nullptr;
  k3_nnet_info ninfo_;
  int fdim_ = 0, device_ = 0;
  int32_t graph_start_ = 0;
  std::vector<float> log_priors_;
  std::vector<std::pair<std::vector<int32_t>, k3_nnet_batch *>> plan_cache_;
  DevBuf<float> d_w_[2], d_f_, d_ll_[2];
  DevBuf<int64_t> d_wo_[2], d_fo_[2];
  PinnedBuf<float> h_w_[2];
  PinnedBuf<int64_t> h_off_[2];
  hipEvent_t ev_h2d_[2] = {nullptr, nullptr};
  bool h2d_recorded_[2] = {false, false};
  hipStream_t s_front_ = nullptr, s_dec_ = nullptr, s_dec_b_ = nullptr; hipEvent_t ev_front_[2] = {nullptr, nullptr}, ev_dec_[2] = {nullptr, nullptr};
  k3_decoder *Dec(int buf) const { return (buf & 1) && dec_b_ ? dec_b_ : dec_; }
  hipStream_t DecStream(int buf) const { return (buf & 1) && dec_b_ ? s_dec_b_ : s_dec_; }
  std::mutex m_; std::condition_variable cv_, wcv_, done_cv_; bool stop_ = false;
  std::deque<std::shared_ptr<Task>> queue_, post_; std::map<std::string, int> groups_; int n_tasks_not_done_ = 0;
  std::thread control_; std::vector<std::thread> workers_;
};
}  // namespace cuda_decoder
}  // namespace k3host
