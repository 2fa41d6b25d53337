// k3-online-pipeline-example -- a caller written against the BatchedThreadedNnet3CudaOnlinePipeline class surface (k3_online_pipeline.h), the way
// cudadecoderbin/batched-wav-nnet3-cuda-online.cc:160-330 drives the reference's class: the wav files are played as concurrent streams, one chunk of
// GetNSampsPerChunk() samples per stream and DecodeBatch call, a lattice callback per correlation id, partial hypotheses printed on request.
//   k3-online-pipeline-example [options] <nnet3-in> <fst-in> <wav-rspecifier> <lattice-wspecifier>
// Its lattices must equal batched-wav-nnet3-cuda-online's / batched-wav-nnet3-cuda2's for the same options (tests/test_cli_gpu.py).
#include <cstring>
#include <iostream>
#include "k3_host.h"
#include "../../include/k3hip.h"
#include "k3_feat_options.h"
#include "k3_online_pipeline.h"
using namespace k3host;
int main(int argc, char **argv) {
  try {
    ParseOptions po("Usage: k3-online-pipeline-example [options] <nnet3-in> <fst-in> <wav-rspecifier> <lattice-wspecifier>\n");
    cuda_decoder::BatchedThreadedNnet3CudaOnlinePipelineConfig cfg; cfg.max_batch_size = 3; cfg.num_worker_threads = 4;
    std::string feature_type = "fbank", mfcc_config, fbank_config;
    bool literal_order = true, print_partial = false;
    float beam = 15.0f, lattice_beam = 10.0f;
    int32_t max_active = 10000;
    po.Register("max-batch-size", &cfg.max_batch_size, "streams per DecodeBatch call"); po.Register("num-channels", &cfg.num_channels, "concurrent streams (-1 = max-batch-size)");
    po.Register("frames-per-chunk", &cfg.frames_per_chunk, "feature frames per chunk and channel");
    po.Register("max-utterance-frames", &cfg.max_utterance_frames, "bound on the decoded frames of a stream");
    po.Register("beam", &beam, "decoding beam"); po.Register("lattice-beam", &lattice_beam, "lattice beam"); po.Register("max-active", &max_active, "max active states");
    po.Register("acoustic-scale", &cfg.acoustic_scale, "acoustic scale"); po.Register("frame-subsampling-factor", &cfg.frame_subsampling_factor, "output frame subsampling");
    po.Register("determinize-lattice", &cfg.determinize_lattice, "determinize before output");
    po.Register("literal-order", &literal_order, "lattices identical to the CPU decoder's");
    po.Register("print-partial-hypotheses", &print_partial, "print the partial hypothesis of every stream after every chunk");
    po.Register("feature-type", &feature_type, "mfcc | fbank"); po.Register("mfcc-config", &mfcc_config, "MFCC config"); po.Register("fbank-config", &fbank_config, "fbank config");
    po.Read(argc, argv);
    if (po.NumArgs() != 4) { po.PrintUsage(); return 1; }
    const bool mfcc = feature_type == "mfcc"; FeatOptions fo(mfcc);
    { ParseOptions fpo(""); fo.Register(&fpo); const std::string &c = mfcc ? mfcc_config : fbank_config; if (!c.empty()) fpo.ReadConfigFile(c); }
    cfg.feature_opts = fo.Finish();
    k3_decoder_config &dc = cfg.decoder_opts; dc.beam = beam; dc.lattice_beam = lattice_beam; dc.max_active = max_active; dc.min_active = std::min(200, max_active - 1);
    dc.frame_tokens_cap = std::min(65536, std::max(4 * max_active, 4096));
    dc.frame_cands_cap = 3 * dc.frame_tokens_cap;
    dc.lane_tokens_cap = 1000000;
    dc.lane_links_cap = 2000000;
    dc.literal_order = literal_order ? 1 : 0;
    TransitionInfo ti = ReadTransitionModel(po.GetArg(1)); k3_nnet *nnet = nullptr; K3H_CHECK_K3(k3_nnet_load(po.GetArg(1).c_str(), &nnet));
    HostFst hfst = ReadFstKaldiGeneric(po.GetArg(2));
    auto scp = ReadScp(po.GetArg(3)); TableWriter writer(po.GetArg(4));
    std::vector<CompactLattice> results(scp.size()); std::vector<char> got(scp.size(), 0); int n_partial = 0, n_endpoint = 0;
    {
      cuda_decoder::BatchedThreadedNnet3CudaOnlinePipeline pipeline(cfg, hfst, nnet, ti);
      const size_t chunk = (size_t)pipeline.GetNSampsPerChunk();
      struct Stream { size_t utt; Wave wave; size_t pos = 0; };
      std::vector<Stream> live; size_t next = 0;
      while (next < scp.size() || !live.empty()) {
        while (next < scp.size() && pipeline.TryInitCorrID(next)) {      // a new stream per free channel; its lattice comes back through the callback
          Stream s; s.utt = next; s.wave = ReadWave(scp[next].second);
          pipeline.SetLatticeCallback(next, [&results, &got, u = next](CompactLattice &clat) { results[u] = clat; got[u] = 1; });
          live.push_back(std::move(s)); next++;
        }
        std::vector<uint64_t> ids; std::vector<std::vector<float>> chunks; std::vector<bool> first, last;
        for (size_t i = 0; i < live.size() && (int)ids.size() < cfg.max_batch_size; i++) {
          Stream &s = live[i]; const size_t len = std::min(chunk, s.wave.samples.size() - s.pos);
          ids.push_back(s.utt);
          chunks.emplace_back(s.wave.samples.begin() + s.pos, s.wave.samples.begin() + s.pos + len);
          first.push_back(s.pos == 0);
          s.pos += len;
          last.push_back(s.pos == s.wave.samples.size());
        }
        std::vector<std::string> partial; std::vector<bool> endpoint;
        pipeline.DecodeBatch(ids, chunks, first, last, print_partial ? &partial : nullptr, print_partial ? &endpoint : nullptr);
        if (print_partial) for (size_t i = 0; i < ids.size(); i++) {
          n_partial += !partial[i].empty();
          n_endpoint += endpoint[i];
          K3H_LOG << "stream " << scp[ids[i]].first << (last[i] ? " final: " : " partial: ") << partial[i] << (endpoint[i] ? " [endpoint]" : "");
        }
        std::vector<Stream> keep; const size_t nb = ids.size();
        for (size_t i = 0; i < live.size(); i++) if (!(i < nb && last[i])) keep.push_back(std::move(live[i]));
        if (keep.size() > nb) std::rotate(keep.begin(), keep.begin() + std::min(nb, keep.size()), keep.end());      // round robin over the live streams
        live.swap(keep);
      }
      pipeline.WaitForLatticeCallbacks();
    }
    int n_err = 0;
    for (size_t i = 0; i < scp.size(); i++) {
      if (!got[i] || results[i].NumStates() == 0) {
        K3H_WARN << "Failed to decode utterance with id " << scp[i].first;
        n_err++;
        continue;
      }
      writer.WriteCompactLattice(scp[i].first, results[i]);
    }
    writer.Flush(); k3_nnet_destroy(nnet);
    K3H_LOG << "Decoded " << scp.size() << " utterances, " << n_err << " with errors." <<
        (print_partial ? " Non-empty partial hypotheses: " + std::to_string(n_partial) + ", endpoints: " + std::to_string(n_endpoint) : std::string());
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
