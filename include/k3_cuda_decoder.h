// k3_cuda_decoder.h -- kaldi::cuda_decoder::CudaFst and kaldi::cuda_decoder::CudaDecoder with the REFERENCE's public signatures
// (cudadecoder/cuda-fst.h:62-149, cudadecoder/cuda-decoder.h:176-345) implemented over the C ABI of libk3hip.so (include/k3hip.h).
//
// Header-only adapter for a Kaldi build: it needs what the reference's own headers need (OpenFst's public Fst interface, kaldi::Lattice,
// TransitionInformation) and nothing from CUDA.  A maintainer who replaces src/cudadecoder/cuda-{fst,decoder}.{h,cc,cu} by this header keeps
// every caller of these two classes (the batched pipelines, cudadecoderbin/*) compiling; what the classes do happens in the HIP kernels.
// Differences from the reference, all of them in the direction of the CPU decoder:
//   * results: with config.literal_order the raw lattice is LatticeFasterDecoder's own, bit for bit (the reference's GPU decoder is not);
//   * a lane is a channel: the decoder keeps `nchannels` lanes resident; nlanes (<= nchannels) bounds the batch of one AdvanceDecoding call only;
//   * queue overflow is an error for that channel (CudaDecoderException, recoverable), never a silently narrowed beam (cuda-decoder.cc:944-976);
//   * lattice-beam pruning and the raw-lattice build run on the GPU inside GetRawLattice / PrepareForGetRawLattice, so
//     ConcurrentGetRawLatticeSingleChannel only unpacks host arrays (thread-safe per channel like the reference's).
// tests/adapter/cuda_decoder_example.cc compiles this header against the reference's lattice types and runs it on the GPU
// (tests/test_cuda_decoder_adapter_gpu.py).
#ifndef K3_CUDA_DECODER_H_
#define K3_CUDA_DECODER_H_
#include <cfloat>
#include <cmath>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "k3hip.h"
#include "itf/decodable-itf.h"               // kaldi::DecodableInterface (the deprecated AdvanceDecoding overload)
#include "itf/transition-information.h"      // kaldi::TransitionInformation
#include "lat/kaldi-lattice.h"               // kaldi::Lattice, LatticeArc, LatticeWeight (pulls in the Fst interface)

namespace kaldi {
namespace cuda_decoder {

typedef int32 ChannelId;
typedef int32 LaneId;

// cudadecoder/cuda-decodable-itf.h:30-33: a decodable whose log-likelihood rows live in DEVICE memory
class CudaDecodableInterface : public DecodableInterface {
 public:
  virtual BaseFloat *GetLogLikelihoodsCudaPointer(int32 subsampled_frame) = 0;
};

class CudaDecoderException : public std::exception {      // cuda-decoder-common.h:100-127
 public:
  CudaDecoderException(const char *str_, const char *file_, int line_, bool recoverable_) : str(str_), file(file_), line(line_), recoverable(recoverable_),
      buffer(std::string(file_) + ":" + std::to_string(line_) + " :" + str_) {}
  const char *what() const throw() { return buffer.c_str(); }
  const char *str; const char *file; const int line; const bool recoverable; const std::string buffer;
};
#define K3_CUDEC_CALL(expr) do { if ((expr) != 0) throw ::kaldi::cuda_decoder::CudaDecoderException(k3_last_error(), __FILE__, __LINE__, false); } while (0)

// cuda-decoder.h:58-163 (the fields the k3 decoder has a use for; the queue capacities map to the k3 capacities)
struct CudaDecoderConfig {
  BaseFloat default_beam = 15.0, lattice_beam = 10.0;
  int32 ntokens_pre_allocated = 1000000, main_q_capacity = -1, aux_q_capacity = -1, max_active = 10000;
  int32 min_active = 200;              // LatticeFasterDecoderConfig::min_active (the reference GPU decoder has no such knob; the CPU decoder does)
  BaseFloat beam_delta = 0.5, hash_ratio = 2.0;
  bool literal_order = true;           // raw lattices identical to LatticeFasterDecoder's (k3_decoder_config.literal_order)
  // frames a channel can hold before GetRawLattice (the reference keeps a channel's tokens in host memory and has no such bound; here they stay in HBM)
  int32 max_frames_per_channel = 3000;
  void Check() const { KALDI_ASSERT(default_beam > 0.0 && max_active > 1 && lattice_beam > 0.0 && (aux_q_capacity == -1 || aux_q_capacity >= main_q_capacity)); }
  void ComputeConfig() { if (main_q_capacity == -1) main_q_capacity = 4 * max_active; if (aux_q_capacity == -1) aux_q_capacity = 3 * main_q_capacity; }
};

struct PartialHypothesis { std::string out_str; std::vector<int32> words; void clear() { out_str.clear(); words.clear(); } };      // cuda-decoder-common.h:590-594 (+ the word ids)

// ---------------------------------------------------------------------------------------------------------------- CudaFst
class CudaFst {
 public:
  CudaFst() {}
  // cuda-fst.h:64 / cuda-fst.cc:38-197: the decoding graph as a CSR in device memory, transition-ids mapped to pdf-ids on the ilabels
  CudaFst(const fst::StdFst &fst, const TransitionInformation *trans_model = NULL) { Initialize(fst, trans_model); }
  void Initialize(const fst::StdFst &fst, const TransitionInformation *trans_model = NULL) {
    Finalize();
    std::vector<int32> off(1, 0), il, ol, nx; std::vector<float> w, fin; int32 max_tid = 0;
    for (fst::StateIterator<fst::StdFst> siter(fst); !siter.Done(); siter.Next()) {
      const int32 s = siter.Value();
      for (fst::ArcIterator<fst::StdFst> aiter(fst, s); !aiter.Done(); aiter.Next()) {
        const fst::StdArc &arc = aiter.Value();
        il.push_back(arc.ilabel); ol.push_back(arc.olabel); nx.push_back(arc.nextstate); w.push_back(arc.weight.Value()); max_tid = std::max<int32>(max_tid, arc.ilabel);
      }
      off.push_back((int32)il.size()); fin.push_back(fst.Final(s).Value());
    }
    num_states_ = (int32)fin.size(); start_ = fst.Start(); num_arcs_ = (int64)il.size();
    std::vector<int32> tid2pdf(max_tid + 1, 0);
    for (int32 t = 1; t <= max_tid; t++) tid2pdf[t] = trans_model ? trans_model->TransitionIdToPdf(t) : t - 1;      // no model: ilabels are pdf-ids + 1 (cuda-fst.cc:166-175)
    num_pdfs_ = 0;      // columns of a log-likelihood row the graph can ask for (the reference sizes its rows by max_ilabel_ the same way, cuda-fst.cc:139-141)
    for (int32 l : il) if (l != 0) num_pdfs_ = std::max<int32>(num_pdfs_, tid2pdf[l] + 1);
    K3_CUDEC_CALL(k3_fst_create(num_states_, start_, off.data(), il.data(), ol.data(), w.data(), nx.data(), fin.data(), tid2pdf.data(), (int32)tid2pdf.size(), &fst_));
  }
  void Finalize() { if (fst_) k3_fst_destroy(fst_); fst_ = NULL; }
  ~CudaFst() { Finalize(); }
  inline uint32_t NumStates() const { return (uint32_t)num_states_; }
  inline int32 Start() const { return start_; }
  inline int32 NumPdfs() const { return num_pdfs_; }      // 1 + the largest pdf-id on an arc
  const k3_fst *Handle() const { return fst_; }
 private:
  CudaFst(const CudaFst &); CudaFst &operator=(const CudaFst &);
  k3_fst *fst_ = NULL; int32 num_states_ = 0, start_ = 0, num_pdfs_ = 0; int64 num_arcs_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------- CudaDecoder
class CudaDecoder {
 public:
  // cuda-decoder.h:224-232, the reference's two constructors with the reference's argument lists.  nlanes = the largest batch AdvanceDecoding is
  // called with (the reference's "lanes"); every channel keeps its own state resident, so nlanes only bounds the batch (checked there).  The
  // width of a log-likelihood row is the graph's: CudaFst::NumPdfs().
  CudaDecoder(const CudaFst &fst, const CudaDecoderConfig &config, int32 nlanes, int32 nchannels)
      : fst_(fst), nlanes_(nlanes), nchannels_(nchannels), num_pdfs_(fst.NumPdfs()), raw_(nchannels), raw_lock_(nchannels) {
    KALDI_ASSERT(nlanes > 0 && nchannels > 0 && nlanes <= nchannels);      // cuda-decoder.cc:66-68
    CudaDecoderConfig c = config; c.Check(); c.ComputeConfig();
    k3_decoder_config kc; k3_decoder_config_default(&kc);
    kc.beam = c.default_beam;
    kc.lattice_beam = c.lattice_beam;
    kc.max_active = c.max_active;
    kc.min_active = std::min(c.min_active, c.max_active - 1);
    kc.beam_delta = c.beam_delta;
    kc.frame_tokens_cap = std::min(65536, std::max(c.main_q_capacity, 4096)); kc.frame_cands_cap = std::max(c.aux_q_capacity, 2 * kc.frame_tokens_cap);
    kc.lane_tokens_cap = std::max<int64_t>(c.ntokens_pre_allocated, kc.frame_tokens_cap); kc.lane_links_cap = 2 * kc.lane_tokens_cap;
    kc.literal_order = c.literal_order ? 1 : 0; kc.hash_ratio = c.hash_ratio;
    K3_CUDEC_CALL(k3_decoder_create(fst.Handle(), &kc, nchannels, num_pdfs_, &dec_));
    K3_CUDEC_CALL(k3_decoder_init_decoding(dec_, nchannels, c.max_frames_per_channel, NULL));
    partial_.resize(nchannels); endpoint_.assign(nchannels, false);
  }
  CudaDecoder(const CudaFst &fst, const CudaDecoderConfig &config, int32 nchannels) : CudaDecoder(fst, config, nchannels, nchannels) {}
  // cuda-decoder.h:338-345: the reference hands the host-side lattice preparation to CPU workers; here that work runs on the GPU (pruning + compaction kernels), nothing to start
  template <typename ThreadPoolT> void SetThreadPoolAndStartCPUWorkers(ThreadPoolT *, int32) {}
  virtual ~CudaDecoder() { if (dec_) k3_decoder_destroy(dec_); }

  // cuda-decoder.h:240: (re)start the listed channels
  void InitDecoding(const std::vector<ChannelId> &channels) {
    K3_CUDEC_CALL(k3_decoder_init_channels(dec_, channels.data(), (int32)channels.size(), NULL));
    for (ChannelId c : channels) { partial_[c].clear(); endpoint_[c] = false; }
  }
  // cuda-decoder.h:262: one more frame for every listed channel; the second member of a pair is a DEVICE pointer to that frame's log-likelihoods
  void AdvanceDecoding(const std::vector<std::pair<ChannelId, const BaseFloat *>> &lanes_assignements) {
    KALDI_ASSERT((int32)lanes_assignements.size() <= nlanes_);      // cuda-decoder.cc:1183
    std::vector<ChannelId> ch; std::vector<const float *> rows;
    for (const auto &p : lanes_assignements) { ch.push_back(p.first); rows.push_back(p.second); }
    K3_CUDEC_CALL(k3_decoder_advance_decoding_lanes(dec_, (int32)ch.size(), ch.data(), rows.data(), 1, num_pdfs_, NULL));
    if (generate_partial_hypotheses_ || endpointing_) UpdatePartial(ch);
  }
  // cuda-decoder.h:267-270 / cuda-decoder.cc:798-830, "Version with deprecated API - will be removed at some point": as many frames as EVERY listed channel's decodable has
  // ready beyond what the channel has decoded (at most max_num_frames when that is >= 0), frame by frame through the call above
  void AdvanceDecoding(const std::vector<ChannelId> &channels, std::vector<CudaDecodableInterface *> &decodables, int32 max_num_frames = -1) {
    KALDI_ASSERT(channels.size() == decodables.size());
    int32 nframes_to_decode = std::numeric_limits<int32>::max();
    for (size_t ilane = 0; ilane < channels.size(); ++ilane) {
      const int32 num_frames_decoded = NumFramesDecoded(channels[ilane]), num_frames_ready = decodables[ilane]->NumFramesReady();
      KALDI_ASSERT(num_frames_decoded >= 0 && "You must call InitDecoding() before AdvanceDecoding()");
      KALDI_ASSERT(num_frames_ready >= num_frames_decoded);      // (the decodable object must not change between calls)
      nframes_to_decode = std::min(nframes_to_decode, num_frames_ready - num_frames_decoded);
    }
    if (max_num_frames >= 0) nframes_to_decode = std::min(nframes_to_decode, max_num_frames);
    std::vector<std::pair<ChannelId, const BaseFloat *>> lanes_assignments;
    for (int32 f = 0; f < nframes_to_decode; ++f) {
      lanes_assignments.clear();
      for (size_t ilane = 0; ilane < channels.size(); ++ilane)
        lanes_assignments.push_back({channels[ilane], decodables[ilane]->GetLogLikelihoodsCudaPointer(NumFramesDecoded(channels[ilane]))});
      AdvanceDecoding(lanes_assignments);
    }
  }
  // several frames at once (rows ld floats apart): what BatchedThreadedNnet3CudaPipeline2 does frame by frame, in one launch
  void AdvanceDecoding(const std::vector<std::pair<ChannelId, const BaseFloat *>> &lanes_assignements, int32 num_frames, int64 ld) {
    std::vector<ChannelId> ch; std::vector<const float *> rows;
    for (const auto &p : lanes_assignements) { ch.push_back(p.first); rows.push_back(p.second); }
    K3_CUDEC_CALL(k3_decoder_advance_decoding_lanes(dec_, (int32)ch.size(), ch.data(), rows.data(), num_frames, ld, NULL));
    if (generate_partial_hypotheses_ || endpointing_) UpdatePartial(ch);
  }
  void AllowPartialHypotheses() { generate_partial_hypotheses_ = true; }
  void AllowEndpointing() { if (frame_shift_seconds_ == FLT_MAX) KALDI_ERR << "You must call SetOutputFrameShiftInSeconds() to use endpointing"; endpointing_ = true; }
  void SetOutputFrameShiftInSeconds(BaseFloat f) { frame_shift_seconds_ = f; }
  // the silence transition-ids and the five rules of kaldi::EndpointDetected (online2/online-endpoint.cc:26-72): {must_contain_nonsilence,
  // min_trailing_silence, max_relative_cost, min_utterance_length}
  struct EndpointRule { bool must_contain_nonsilence; BaseFloat min_trailing_silence, max_relative_cost, min_utterance_length; };
  void SetEndpointing(const std::set<int32> &silence_transition_ids, const std::vector<EndpointRule> &rules) { silence_tids_ = silence_transition_ids; rules_ = rules; }
  void GetPartialHypothesis(ChannelId ichannel, PartialHypothesis **out) { KALDI_ASSERT(generate_partial_hypotheses_); *out = &partial_[ichannel]; }
  bool EndpointDetected(ChannelId ichannel) { return endpoint_[ichannel]; }
  int32 NumFramesDecoded(ChannelId ichannel) const { return k3_decoder_num_frames_decoded(dec_, ichannel); }
  void SetSymbolTable(const std::vector<std::string> &word_syms) { word_syms_ = word_syms; }      // (fst::SymbolTable in the reference: a table id -> word)

  // cuda-decoder.h:306
  void GetBestPath(const std::vector<ChannelId> &channels, std::vector<Lattice *> &fst_out_vec, bool use_final_probs = true) {
    KALDI_ASSERT(channels.size() == fst_out_vec.size());
    std::vector<int64_t> off; std::vector<int32> il, ol; std::vector<float> g, ac, fc;
    BestPaths(channels, use_final_probs, &off, &il, &ol, &g, &ac, &fc, NULL);
    for (size_t u = 0; u < channels.size(); u++) {
      Lattice *lat = fst_out_vec[u]; lat->DeleteStates();
      int32 cur = lat->AddState(); lat->SetStart(cur);
      for (int64_t k = off[u]; k < off[u + 1]; k++) { const int32 nxt = lat->AddState(); lat->AddArc(cur, LatticeArc(il[k], ol[k], LatticeWeight(g[k], ac[k]), nxt)); cur = nxt; }
      lat->SetFinal(cur, LatticeWeight(fc[u], 0.0));
    }
  }
  // cuda-decoder.h:309-338.  PrepareForGetRawLattice runs FinalizeDecoding (lattice-beam pruning with or without final-probs is the decoder's:
  // final-probs are used when a final state was reached, like LatticeFasterDecoder) on the GPU and brings the lattices to the host.
  void PrepareForGetRawLattice(const std::vector<ChannelId> &channels, bool use_final_probs) {
    (void)use_final_probs;
    K3_CUDEC_CALL(k3_decoder_finalize_channels(dec_, channels.data(), (int32)channels.size(), NULL));
    const int32 n = (int32)channels.size(); std::vector<int64_t> info(10 * (size_t)n);
    const int rc = k3_decoder_lattice_info(dec_, info.data());
    if (rc != 0 && rc != K3_ERR_OVERFLOW) throw CudaDecoderException(k3_last_error(), __FILE__, __LINE__, false);
    int64_t NS = 0, NA = 0; for (int32 u = 0; u < n; u++) { NS += info[10 * u]; NA += info[10 * u + 1]; }
    std::vector<int32> sf(NS + 1), ss(NS + 1), as(NA + 1), ad(NA + 1), ai(NA + 1), ao(NA + 1); std::vector<float> sc(NS + 1), sfin(NS + 1), ag(NA + 1), aa(NA + 1);
    K3_CUDEC_CALL(k3_decoder_get_raw_lattices(dec_, sf.data(), ss.data(), sc.data(), sfin.data(), as.data(), ad.data(), ai.data(), ao.data(), ag.data(), aa.data()));
    int64_t s0 = 0, a0 = 0;
    for (int32 u = 0; u < n; u++) {
      const int64_t ns = info[10 * u], na = info[10 * u + 1]; std::lock_guard<std::mutex> lk(raw_lock_[channels[u]]);
      Raw &r = raw_[channels[u]]; r.failed = info[10 * u + 2] < 0;
      r.frame.assign(sf.begin() + s0, sf.begin() + s0 + ns); r.state.assign(ss.begin() + s0, ss.begin() + s0 + ns); r.fin.assign(sfin.begin() + s0, sfin.begin() + s0 + ns);
      r.src.assign(as.begin() + a0, as.begin() + a0 + na);
      r.dst.assign(ad.begin() + a0, ad.begin() + a0 + na);
      r.il.assign(ai.begin() + a0, ai.begin() + a0 + na);
      r.ol.assign(ao.begin() + a0, ao.begin() + a0 + na);
      r.g.assign(ag.begin() + a0, ag.begin() + a0 + na); r.ac.assign(aa.begin() + a0, aa.begin() + a0 + na);
      s0 += ns; a0 += na;
    }
  }
  void ConcurrentGetRawLatticeSingleChannel(ChannelId ichannel, Lattice *fst_out) {      // thread-safe per channel
    std::lock_guard<std::mutex> lk(raw_lock_[ichannel]);
    const Raw &r = raw_[ichannel];
    if (r.failed) throw CudaDecoderException("the channel exceeded the decoder capacities", __FILE__, __LINE__, true);
    fst_out->DeleteStates();
    const int32 ns = (int32)r.frame.size(); int32 start = -1;
    for (int32 s = 0; s < ns; s++) { fst_out->AddState(); if (r.frame[s] == 0 && r.state[s] == fst_.Start()) start = s; }
    if (start >= 0) fst_out->SetStart(start);
    for (int32 s = 0; s < ns; s++) if (r.fin[s] != std::numeric_limits<float>::infinity()) fst_out->SetFinal(s, LatticeWeight(r.fin[s], 0.0));
    for (size_t a = 0; a < r.src.size(); a++) fst_out->AddArc(r.src[a], LatticeArc(r.il[a], r.ol[a], LatticeWeight(r.g[a], r.ac[a]), r.dst[a]));
  }
  void GetRawLattice(const std::vector<ChannelId> &channels, std::vector<Lattice *> &fst_out_vec, bool use_final_probs) {
    KALDI_ASSERT(channels.size() == fst_out_vec.size());
    PrepareForGetRawLattice(channels, use_final_probs);
    for (size_t u = 0; u < channels.size(); u++) ConcurrentGetRawLatticeSingleChannel(channels[u], fst_out_vec[u]);
  }
  int64_t OrderSensitiveEvents(ChannelId) const { return 0; }
  k3_decoder *Handle() { return dec_; }

 private:
  CudaDecoder(const CudaDecoder &); CudaDecoder &operator=(const CudaDecoder &);
  struct Raw { bool failed = false; std::vector<int32> frame, state, src, dst, il, ol; std::vector<float> fin, g, ac; };
  void BestPaths(const std::vector<ChannelId> &channels, bool use_final, std::vector<int64_t> *off, std::vector<int32> *il, std::vector<int32> *ol,
      std::vector<float> *g, std::vector<float> *ac,
                 std::vector<float> *fc, std::vector<float> *rel) {
    const int32 n = (int32)channels.size(); int64_t cap = 0; for (ChannelId c : channels) cap += 4 * std::max(1, NumFramesDecoded(c)) + 64;
    off->assign(n + 1, 0);
    il->assign(cap, 0);
    ol->assign(cap, 0);
    g->assign(cap, 0.0f);
    ac->assign(cap, 0.0f);
    fc->assign(n, 0.0f);
    std::vector<float> r(n);
    std::vector<int32> rf(n);
    K3_CUDEC_CALL(k3_decoder_get_best_path(dec_, channels.data(), n, use_final ? 1 : 0, off->data(), cap, il->data(), ol->data(), g->data(), ac->data(),
        fc->data(), r.data(), rf.data()));
    if (rel) *rel = r;
  }
  void UpdatePartial(const std::vector<ChannelId> &channels) {      // GeneratePartialPath + EndpointDetected + BuildPartialHypothesisOutput (cuda-decoder.cc:1864-2003)
    std::vector<int64_t> off; std::vector<int32> il, ol; std::vector<float> g, ac, fc, rel;
    BestPaths(channels, false, &off, &il, &ol, &g, &ac, &fc, &rel);
    for (size_t u = 0; u < channels.size(); u++) {
      const ChannelId c = channels[u];
      if (generate_partial_hypotheses_) {
        PartialHypothesis &ph = partial_[c]; ph.clear();
        for (int64_t k = off[u]; k < off[u + 1]; k++) if (ol[k] != 0) {
          ph.words.push_back(ol[k]);
          if (!ph.out_str.empty()) ph.out_str += " ";
          ph.out_str += (size_t)ol[k] < word_syms_.size() ? word_syms_[ol[k]] : std::to_string(ol[k]);
        }
      }
      if (endpointing_) {
        int32 sil = 0, frames = 0;
        for (int64_t k = off[u + 1] - 1; k >= off[u]; k--) { if (il[k] == 0) continue; if (!silence_tids_.count(il[k])) break; sil++; }
        for (int64_t k = off[u]; k < off[u + 1]; k++) frames += il[k] != 0;
        const BaseFloat len = frames * frame_shift_seconds_, ts = sil * frame_shift_seconds_; bool ans = false;
        for (const EndpointRule &r : rules_) ans = ans ||
            (((len > ts) || !r.must_contain_nonsilence) && ts >= r.min_trailing_silence && rel[u] <= r.max_relative_cost && len >= r.min_utterance_length);
        endpoint_[c] = ans;
      }
    }
  }
  const CudaFst &fst_; int32 nlanes_, nchannels_, num_pdfs_; k3_decoder *dec_ = NULL;
  bool generate_partial_hypotheses_ = false, endpointing_ = false; BaseFloat frame_shift_seconds_ = FLT_MAX;
  std::vector<PartialHypothesis> partial_; std::vector<bool> endpoint_; std::vector<std::string> word_syms_; std::set<int32> silence_tids_; std::vector<EndpointRule> rules_;
  std::vector<Raw> raw_; std::vector<std::mutex> raw_lock_;
};

}  // namespace cuda_decoder
}  // namespace kaldi
#endif  // K3_CUDA_DECODER_H_
