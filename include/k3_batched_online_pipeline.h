// k3_batched_online_pipeline.h -- kaldi::cuda_decoder::BatchedThreadedNnet3CudaOnlinePipeline with the REFERENCE's constructor, DecodeBatch and callback types
// (cudadecoder/batched-threaded-nnet3-cuda-online-pipeline.h:119-330): built from an fst::Fst<fst::StdArc>, an nnet3::AmNnetSimple and a TransitionModel, fed
// std::vector<SubVector<BaseFloat>> chunks per correlation id, handing kaldi::CompactLattice objects to the lattice callbacks.  Header-only adapter for a Kaldi build, the streaming
// twin of include/k3_batched_pipeline.h (whose conversions it reuses); the work is done by include/k3_online_pipeline.h over the C ABI.  tests/adapter/cuda_online_pipeline_example.cc
// compiles it against the reference's headers; tests/test_cuda_decoder_adapter_gpu.py runs it on the GPU.
#ifndef K3_BATCHED_ONLINE_PIPELINE_H_
#define K3_BATCHED_ONLINE_PIPELINE_H_
#include "k3_batched_pipeline.h"
#include "k3_online_pipeline.h"          // k3host::cuda_decoder::BatchedThreadedNnet3CudaOnlinePipeline

namespace kaldi {
namespace cuda_decoder {

struct BatchedThreadedNnet3CudaOnlinePipelineConfig {      // :51-117, the options the accelerated path reads, under the reference's member names
  // OnlineNnet2FeaturePipelineConfig's feature_type / *_config / ivector_extraction_config
  BatchedThreadedNnet3CudaPipeline2Config::FeatureOpts feature_opts;
  std::string ivector_extraction_config;
  struct ComputeOpts { BaseFloat acoustic_scale = 0.1; int32 frame_subsampling_factor = 1, frames_per_chunk = 50; } compute_opts;
  BatchedThreadedNnet3CudaPipeline2Config::DecoderOpts decoder_opts; BatchedThreadedNnet3CudaPipeline2Config::DetOpts det_opts;
  int32 max_batch_size = 400, num_channels = -1, num_worker_threads = -1, num_decoder_copy_threads = 2;
  bool determinize_lattice = true, use_gpu_feature_extraction = true, reset_on_endpoint = false;
  int32 max_utterance_frames = 6000;      // (the lane pools are sized per channel: frames a stream may have)
};

class BatchedThreadedNnet3CudaOnlinePipeline {
 public:
  using CorrelationID = uint64_t;
  typedef std::function<void(const std::string &, bool, bool)> BestPathCallback;
  typedef std::function<void(CompactLattice &)> LatticeCallback;
  BatchedThreadedNnet3CudaOnlinePipeline(const BatchedThreadedNnet3CudaOnlinePipelineConfig &config, const fst::Fst<fst::StdArc> &decode_fst,
      const nnet3::AmNnetSimple &am_nnet, const TransitionModel &trans_model)
      : config_(config) {
    char tmpl[] = "/tmp/k3_online_pipeline_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd < 0) KALDI_ERR << "cannot create a temporary file for the model";
    close(fd);
    const std::string mdl = tmpl;
    { std::ofstream os(mdl, std::ios::binary); os << '\0' << 'B'; trans_model.Write(os, true); am_nnet.Write(os, true); if (!os) KALDI_ERR << "cannot write " << mdl; }
    if (k3_nnet_load(mdl.c_str(), &nnet_) != 0) { const std::string e = k3_last_error(); unlink(mdl.c_str()); KALDI_ERR << "k3_nnet_load: " << e; }
    trans_ = k3host::ReadTransitionModel(mdl); unlink(mdl.c_str());
    k3host::HostFst h; h.start = decode_fst.Start(); h.arc_offsets.push_back(0);
    for (fst::StateIterator<fst::Fst<fst::StdArc> > siter(decode_fst); !siter.Done(); siter.Next()) {
      const int32 s = siter.Value();
      for (fst::ArcIterator<fst::Fst<fst::StdArc> > aiter(decode_fst, s); !aiter.Done(); aiter.Next()) {
        const fst::StdArc &arc = aiter.Value();
        h.ilabel.push_back(arc.ilabel);
        h.olabel.push_back(arc.olabel);
        h.nextstate.push_back(arc.nextstate);
        h.weight.push_back(arc.weight.Value());
      }
      h.arc_offsets.push_back((int32)h.ilabel.size()); h.final_cost.push_back(decode_fst.Final(s).Value());
    }
    k3host::cuda_decoder::BatchedThreadedNnet3CudaOnlinePipelineConfig c;
    const bool mfcc = config.feature_opts.feature_type == "mfcc";
    if (!mfcc && config.feature_opts.feature_type != "fbank") KALDI_ERR << "Invalid feature type: " << config.feature_opts.feature_type << " (supported: mfcc, fbank)";
    k3host::FeatOptions fo(mfcc);
    {
      k3host::ParseOptions fpo("");
      fo.Register(&fpo);
      const std::string &cfg = mfcc ? config.feature_opts.mfcc_config : config.feature_opts.fbank_config;
      if (!cfg.empty()) fpo.ReadConfigFile(cfg);
    }
    c.feature_opts = fo.Finish(); c.ivector_extraction_config = config.ivector_extraction_config;
    c.max_batch_size = config.max_batch_size;
    c.num_channels = config.num_channels;
    c.num_worker_threads = config.num_worker_threads;
    c.determinize_lattice = config.determinize_lattice;
    c.det_opts.delta = config.det_opts.delta;
    c.det_opts.max_mem = config.det_opts.max_mem;
    c.det_opts.phone_determinize = config.det_opts.phone_determinize;
    c.det_opts.word_determinize = config.det_opts.word_determinize;
    c.det_opts.minimize = config.det_opts.minimize;
    c.acoustic_scale = config.compute_opts.acoustic_scale;
    c.frame_subsampling_factor = config.compute_opts.frame_subsampling_factor;
    c.frames_per_chunk = config.compute_opts.frames_per_chunk;
    c.max_utterance_frames = config.max_utterance_frames;
    k3_decoder_config &dc = c.decoder_opts; const BatchedThreadedNnet3CudaPipeline2Config::DecoderOpts &o = config.decoder_opts;
    const int32 mq = o.main_q_capacity == -1 ? 4 * o.max_active : o.main_q_capacity, aq = o.aux_q_capacity == -1 ? 3 * mq : o.aux_q_capacity;
    dc.beam = o.default_beam; dc.lattice_beam = o.lattice_beam; dc.max_active = o.max_active; dc.min_active = std::min(200, o.max_active - 1);
    dc.frame_tokens_cap = std::min(65536, std::max(mq, 4096));
    dc.frame_cands_cap = std::max(aq, 2 * dc.frame_tokens_cap);
    dc.lane_tokens_cap = std::max<int64_t>(o.ntokens_pre_allocated, dc.frame_tokens_cap);
    dc.lane_links_cap = 2 * dc.lane_tokens_cap;
    dc.literal_order = 1;
    impl_.reset(new k3host::cuda_decoder::BatchedThreadedNnet3CudaOnlinePipeline(c, h, nnet_, trans_));
  }
  virtual ~BatchedThreadedNnet3CudaOnlinePipeline() { impl_.reset(); if (nnet_) k3_nnet_destroy(nnet_); }
  const BatchedThreadedNnet3CudaOnlinePipelineConfig &GetConfig() { return config_; }
  bool TryInitCorrID(CorrelationID corr_id, int wait_for = 0) { return impl_->TryInitCorrID(corr_id, wait_for); }      // :170
  void SetBestPathCallback(CorrelationID corr_id, const BestPathCallback &callback) { impl_->SetBestPathCallback(corr_id, callback); }      // :172-176
  void SetLatticeCallback(CorrelationID corr_id, const LatticeCallback &callback) {      // :179-183
    impl_->SetLatticeCallback(corr_id, [callback](k3host::CompactLattice &c) { CompactLattice clat; BatchedThreadedNnet3CudaPipeline2::ToKaldi(c, &clat); callback(clat); });
  }
  // :209-215.  One chunk (at most GetNSampsPerChunk() samples unless it is the stream's last) per listed stream; partial_hypotheses (optional): the current best path's word ids,
  // valid until the next call
  void DecodeBatch(const std::vector<CorrelationID> &corr_ids, const std::vector<SubVector<BaseFloat> > &wave_samples, const std::vector<bool> &is_first_chunk,
      const std::vector<bool> &is_last_chunk,
                   std::vector<const std::string *> *partial_hypotheses = nullptr, std::vector<bool> *end_point = nullptr) {
    std::vector<std::vector<float> > chunks(wave_samples.size());
    for (size_t i = 0; i < wave_samples.size(); i++) chunks[i].assign(wave_samples[i].Data(), wave_samples[i].Data() + wave_samples[i].Dim());
    impl_->DecodeBatch(corr_ids, chunks, is_first_chunk, is_last_chunk, partial_hypotheses ? &partial_ : nullptr, end_point);
    if (partial_hypotheses) { partial_hypotheses->resize(partial_.size()); for (size_t i = 0; i < partial_.size(); i++) (*partial_hypotheses)[i] = &partial_[i]; }
  }
  int32 GetNSampsPerChunk() const { return impl_->GetNSampsPerChunk(); }
  int32 GetNInputFramesPerChunk() const { return impl_->GetNInputFramesPerChunk(); }
  BaseFloat GetModelFrequency() const { return impl_->GetModelFrequency(); }
  void WaitForLatticeCallbacks() noexcept { impl_->WaitForLatticeCallbacks(); }
 private:
  BatchedThreadedNnet3CudaOnlinePipelineConfig config_; k3_nnet *nnet_ = NULL; k3host::TransitionInfo trans_; std::vector<std::string> partial_;
  std::unique_ptr<k3host::cuda_decoder::BatchedThreadedNnet3CudaOnlinePipeline> impl_;
};

}  // namespace cuda_decoder
}  // namespace kaldi
#endif  // K3_BATCHED_ONLINE_PIPELINE_H_
