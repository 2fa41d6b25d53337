// k3_batched_pipeline.h -- kaldi::cuda_decoder::BatchedThreadedNnet3CudaPipeline2 with the REFERENCE's constructor and callback types
// (cudadecoder/batched-threaded-nnet3-cuda-pipeline2.h:153-239): built from an fst::Fst<fst::StdArc>, an nnet3::AmNnetSimple and a TransitionModel -- the objects a Kaldi
// program holds after reading final.mdl and HCLG.fst -- and handing kaldi::CompactLattice objects to the callbacks.  Header-only adapter for a Kaldi build; it needs what the
// reference's own header needs (OpenFst's public interface, nnet3/am-nnet-simple.h, hmm/transition-model.h, feat/wave-reader.h) and nothing from CUDA.  The work is done by
// the host layer of this repository (kaldi_amd/host/k3_pipeline.h over the C ABI of include/k3hip.h): the Kaldi objects are converted once, in the constructor --
//   decode_fst  -> CSR arrays -> k3_fst_create (transition-ids mapped to pdf-ids like cuda-fst.cc:166-175)
//   am_nnet + trans_model -> written as a .mdl into a temporary file with the reference's own Write() -> k3_nnet_load (the fused TDNN / TDNN-F forward)
//   trans_model -> the transition-id -> pdf / phone / self-loop map the phone-level determinization pass needs.
// tests/adapter/cuda_pipeline_example.cc compiles this header against the reference's headers and runs it on the GPU (tests/test_cuda_decoder_adapter_gpu.py).
#ifndef K3_BATCHED_PIPELINE_H_
#define K3_BATCHED_PIPELINE_H_
#include <cstdio>
#include <fstream>
#include <functional>
#include <memory>
#include <string>
#include <vector>
#include <unistd.h>
#include "k3_pipeline.h"                 // k3host::cuda_decoder::BatchedThreadedNnet3CudaPipeline2
#include "../kaldi_amd/host/k3_feat_options.h"
#include "feat/wave-reader.h"
#include "hmm/transition-model.h"
#include "lat/kaldi-lattice.h"
#include "nnet3/am-nnet-simple.h"

namespace kaldi {
namespace cuda_decoder {

// The options of the reference's config the accelerated path reads, under the reference's member names (batched-threaded-nnet3-cuda-online-pipeline.h:51-135,
// batched-threaded-nnet3-cuda-pipeline2.h:39-55): feature_opts = OnlineNnet2FeaturePipelineConfig's feature_type / mfcc_config / fbank_config, compute_opts =
// NnetSimpleComputationOptions' acoustic_scale / frame_subsampling_factor, decoder_opts = CudaDecoderConfig, det_opts, seg_opts.
struct BatchedThreadedNnet3CudaPipeline2Config {
  struct FeatureOpts { std::string feature_type = "mfcc", mfcc_config, fbank_config; } feature_opts;
  struct ComputeOpts { BaseFloat acoustic_scale = 0.1; int32 frame_subsampling_factor = 1; } compute_opts;
  struct DecoderOpts {
    BaseFloat default_beam = 15.0, lattice_beam = 10.0;
    int32 max_active = 10000, main_q_capacity = -1, aux_q_capacity = -1, ntokens_pre_allocated = 1000000;
  }
  decoder_opts;
  struct DetOpts { BaseFloat delta = 1.0f / 1024.0f; int32 max_mem = 50000000; bool phone_determinize = true, word_determinize = true, minimize = false; } det_opts;
  k3host::cuda_decoder::CudaPipelineSegmentationConfig seg_opts;
  int32 max_batch_size = 400, num_channels = -1, num_worker_threads = -1; bool determinize_lattice = true, use_gpu_feature_extraction = true;
};

class BatchedThreadedNnet3CudaPipeline2 {
 public:
  typedef std::function<void(CompactLattice &)> LatticeCallback;
  BatchedThreadedNnet3CudaPipeline2(const BatchedThreadedNnet3CudaPipeline2Config &config, const fst::Fst<fst::StdArc> &decode_fst,
      const nnet3::AmNnetSimple &am_nnet, const TransitionModel &trans_model) {
    // the model: the reference's own writers, read back by the loader of the fused path
    char tmpl[] = "/tmp/k3_pipeline_XXXXXX"; const int fd = mkstemp(tmpl); if (fd < 0) KALDI_ERR << "cannot create a temporary file for the model"; close(fd); mdl_path_ = tmpl;
    { std::ofstream os(mdl_path_, std::ios::binary); os << '\0' << 'B'; trans_model.Write(os, true); am_nnet.Write(os, true); if (!os) KALDI_ERR << "cannot write " << mdl_path_; }
    if (k3_nnet_load(mdl_path_.c_str(), &nnet_) != 0) { const std::string e = k3_last_error(); unlink(mdl_path_.c_str()); KALDI_ERR << "k3_nnet_load: " << e; }
    trans_ = k3host::ReadTransitionModel(mdl_path_);
    unlink(mdl_path_.c_str());
    // the graph
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
    // the options
    k3host::cuda_decoder::BatchedThreadedNnet3CudaPipeline2Config c;
    const bool mfcc = config.feature_opts.feature_type == "mfcc";
    if (!mfcc && config.feature_opts.feature_type != "fbank") KALDI_ERR << "Invalid feature type: " << config.feature_opts.feature_type << " (supported: mfcc, fbank)";
    k3host::FeatOptions fo(mfcc);
    {
      k3host::ParseOptions fpo("");
      fo.Register(&fpo);
      const std::string &cfg = mfcc ? config.feature_opts.mfcc_config : config.feature_opts.fbank_config;
      if (!cfg.empty()) fpo.ReadConfigFile(cfg);
    }
    c.feature_opts = fo.Finish(); c.max_batch_size = config.max_batch_size; c.num_worker_threads = config.num_worker_threads; c.determinize_lattice = config.determinize_lattice;
    c.det_opts.delta = config.det_opts.delta;
    c.det_opts.max_mem = config.det_opts.max_mem;
    c.det_opts.phone_determinize = config.det_opts.phone_determinize;
    c.det_opts.word_determinize = config.det_opts.word_determinize;
    c.det_opts.minimize = config.det_opts.minimize;
    c.acoustic_scale = config.compute_opts.acoustic_scale; c.frame_subsampling_factor = config.compute_opts.frame_subsampling_factor; c.seg_opts = config.seg_opts;
    k3_decoder_config &dc = c.decoder_opts; const DecoderOptsResolved d(config.decoder_opts);
    dc.beam = d.beam; dc.lattice_beam = d.lattice_beam; dc.max_active = d.max_active; dc.min_active = std::min(200, d.max_active - 1);
    dc.frame_tokens_cap = d.frame_tokens_cap;
    dc.frame_cands_cap = d.frame_cands_cap;
    dc.lane_tokens_cap = d.lane_tokens_cap;
    dc.lane_links_cap = 2 * d.lane_tokens_cap;
    dc.literal_order = 1;
    impl_.reset(new k3host::cuda_decoder::BatchedThreadedNnet3CudaPipeline2(c, h, nnet_, trans_));
  }
  virtual ~BatchedThreadedNnet3CudaPipeline2() { impl_.reset(); if (nnet_) k3_nnet_destroy(nnet_); }

  // :171-200.  The callback runs on a worker thread once the (determinized) lattice of the utterance is ready; an utterance that could not be decoded gets an empty lattice.
  void DecodeWithCallback(const std::shared_ptr<WaveData> &wave_data, const LatticeCallback &callback, const std::string &group = std::string()) {
    const SubVector<BaseFloat> ch0(wave_data->Data(), 0);
    DecodeWithCallback(ch0, wave_data->SampFreq(), callback, group);
  }
  void DecodeWithCallback(const VectorBase<BaseFloat> &wave_data, float sample_rate, const LatticeCallback &callback, const std::string &group = std::string()) {
    std::vector<float> samples(wave_data.Data(), wave_data.Data() + wave_data.Dim());
    impl_->DecodeWithCallback(samples, sample_rate, [callback](k3host::CompactLattice &c) { CompactLattice clat; ToKaldi(c, &clat); callback(clat); }, group);
  }
  void CreateTaskGroup(const std::string &group) { impl_->CreateTaskGroup(group); }
  void DestroyTaskGroup(const std::string &group) { impl_->DestroyTaskGroup(group); }
  void WaitForGroup(const std::string &group) { impl_->WaitForGroup(group); }
  void WaitForAllTasks() { impl_->WaitForAllTasks(); }
  BaseFloat GetModelFrequency() const { return impl_->GetModelFrequency(); }
  void SetLatticePostprocessor(const std::shared_ptr<k3host::LatticePostprocessor> &pp) { impl_->SetLatticePostprocessor(pp); }

  // k3host::CompactLattice <-> kaldi::CompactLattice
  static void ToKaldi(const k3host::CompactLattice &c, CompactLattice *out) {
    out->DeleteStates();
    for (int32 s = 0; s < c.NumStates(); s++) out->AddState();
    if (c.NumStates() == 0) return;
    out->SetStart(c.start);
    for (int32 s = 0; s < c.NumStates(); s++) if (c.is_final[s]) out->SetFinal(s, CompactLatticeWeight(LatticeWeight(c.fin_graph[s], c.fin_ac[s]), c.fin_str[s]));
    for (size_t a = 0; a < c.arc_src.size(); a++) out->AddArc(c.arc_src[a],
        CompactLatticeArc(c.arc_label[a], c.arc_label[a], CompactLatticeWeight(LatticeWeight(c.arc_graph[a], c.arc_ac[a]), c.arc_str[a]), c.arc_dst[a]));
  }
  static void FromKaldi(const CompactLattice &clat, k3host::CompactLattice *c) {
    *c = k3host::CompactLattice(); const int32 n = clat.NumStates(); if (n == 0) return;
    for (int32 s = 0; s < n; s++) c->AddState();
    c->start = clat.Start();
    for (int32 s = 0; s < n; s++) {
      const CompactLatticeWeight f = clat.Final(s);
      if (f != CompactLatticeWeight::Zero()) { c->is_final[s] = 1; c->fin_graph[s] = f.Weight().Value1(); c->fin_ac[s] = f.Weight().Value2(); c->fin_str[s] = f.String(); }
      for (fst::ArcIterator<CompactLattice> it(clat, s); !it.Done(); it.Next()) {
        const CompactLatticeArc &arc = it.Value();
        c->arc_src.push_back(s);
        c->arc_dst.push_back(arc.nextstate);
        c->arc_label.push_back(arc.ilabel);
        c->arc_graph.push_back(arc.weight.Weight().Value1());
        c->arc_ac.push_back(arc.weight.Weight().Value2());
        c->arc_str.push_back(arc.weight.String());
      }
    }
  }

 private:
  struct DecoderOptsResolved {      // CudaDecoderConfig::ComputeConfig (cuda-decoder.h:150-162) + the mapping of include/k3_cuda_decoder.h
    BaseFloat beam, lattice_beam; int32 max_active, frame_tokens_cap, frame_cands_cap; int64_t lane_tokens_cap;
    explicit DecoderOptsResolved(const BatchedThreadedNnet3CudaPipeline2Config::DecoderOpts &o) : beam(o.default_beam), lattice_beam(o.lattice_beam), max_active(o.max_active) {
      const int32 mq = o.main_q_capacity == -1 ? 4 * o.max_active : o.main_q_capacity, aq = o.aux_q_capacity == -1 ? 3 * mq : o.aux_q_capacity;
      frame_tokens_cap = std::min(65536, std::max(mq, 4096));
      frame_cands_cap = std::max(aq, 2 * frame_tokens_cap);
      lane_tokens_cap = std::max<int64_t>(o.ntokens_pre_allocated, frame_tokens_cap);
    }
  };
  std::string mdl_path_; k3_nnet *nnet_ = NULL; k3host::TransitionInfo trans_;
  std::unique_ptr<k3host::cuda_decoder::BatchedThreadedNnet3CudaPipeline2> impl_;
};

}  // namespace cuda_decoder
}  // namespace kaldi
#endif  // K3_BATCHED_PIPELINE_H_
