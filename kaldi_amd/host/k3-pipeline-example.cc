// k3-pipeline-example -- a caller written against the BatchedThreadedNnet3CudaPipeline2 class surface (k3_pipeline.h), the way
// cudadecoderbin/batched-wav-nnet3-cuda2.cc:170-245 drives the reference's class: one DecodeWithCallback per utterance, callbacks that hand the
// lattice to the writer, two task groups waited for separately, then WaitForAllTasks.
// k3-pipeline-example [--max-batch-size=N] [--beam= --lattice-beam= --max-active= --acoustic-scale= --frame-subsampling-factor= --fbank-config=] <nnet3-in>
// <fst-in> <wav-rspecifier> <lattice-wspecifier>
// Its output must equal batched-wav-nnet3-cuda2's for the same options (tests/test_cli_gpu.py).
#include <fstream>
#include <cstring>
#include <iostream>
#include <mutex>
#include <sstream>
#include "k3_host.h"
#include "../../include/k3hip.h"
#include "k3_feat_options.h"
#include "k3_pipeline.h"
using namespace k3host;
int main(int argc, char **argv) {
  try {
    ParseOptions po("Usage: k3-pipeline-example [options] <nnet3-in> <fst-in> <wav-rspecifier> <lattice-wspecifier>\n");
    cuda_decoder::BatchedThreadedNnet3CudaPipeline2Config cfg; std::string feature_type = "fbank", mfcc_config, fbank_config; int32_t worker_threads = 4; bool literal_order = true;
    float beam = 15.0f, lattice_beam = 10.0f; int32_t max_active = 10000;
    po.Register("max-batch-size", &cfg.max_batch_size, "utterances decoded together"); po.Register("cuda-worker-threads", &worker_threads, "post-processing threads");
    po.Register("beam", &beam, "decoding beam"); po.Register("lattice-beam", &lattice_beam, "lattice beam"); po.Register("max-active", &max_active, "max active states");
    po.Register("acoustic-scale", &cfg.acoustic_scale, "acoustic scale"); po.Register("frame-subsampling-factor", &cfg.frame_subsampling_factor, "output frame subsampling");
    po.Register("determinize-lattice", &cfg.determinize_lattice, "determinize before output");
    po.Register("literal-order", &literal_order, "lattices identical to the CPU decoder's");
    po.Register("feature-type", &feature_type, "mfcc | fbank"); po.Register("mfcc-config", &mfcc_config, "MFCC config"); po.Register("fbank-config", &fbank_config, "fbank config");
    std::string postproc, ctm_out;
    po.Register("lattice-postprocessor-rxfilename", &postproc, "Config file for the lattice postprocessor (SetLatticePostprocessor)");
    po.Register("ctm-out", &ctm_out, "with --segmentation: also write the merged CTM of every file here (RESULT_TYPE_CTM)");
    bool segmentation = false;
    po.Register("segmentation", &segmentation, "Split audio files into segments (SegmentedDecodeWithCallback; keys [utt]-[offset])");
    cfg.seg_opts.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() != 4) { po.PrintUsage(); return 1; }
    const bool mfcc = feature_type == "mfcc"; FeatOptions fo(mfcc);
    { ParseOptions fpo(""); fo.Register(&fpo); const std::string &c = mfcc ? mfcc_config : fbank_config; if (!c.empty()) fpo.ReadConfigFile(c); }
    cfg.feature_opts = fo.Finish(); cfg.num_worker_threads = worker_threads;
    k3_decoder_config &dc = cfg.decoder_opts; dc.beam = beam; dc.lattice_beam = lattice_beam; dc.max_active = max_active; dc.min_active = std::min(200, max_active - 1);
    dc.frame_tokens_cap = std::min(65536, std::max(4 * max_active, 4096));
    dc.frame_cands_cap = 3 * dc.frame_tokens_cap;
    dc.lane_tokens_cap = 1000000;
    dc.lane_links_cap = 2000000;
    dc.literal_order = literal_order ? 1 : 0;
    TransitionInfo ti = ReadTransitionModel(po.GetArg(1)); k3_nnet *nnet = nullptr; K3H_CHECK_K3(k3_nnet_load(po.GetArg(1).c_str(), &nnet));
    HostFst hfst = ReadFstKaldiGeneric(po.GetArg(2));
    auto scp = ReadScp(po.GetArg(3)); TableWriter writer(po.GetArg(4));
    if (segmentation) {      // the way cudadecoderbin/batched-wav-nnet3-cuda2.cc:196-232 drives it: one segmented callback per file, WriteLattices with print_offsets
      std::mutex wm; int n_seg = 0; std::ofstream ctm; if (!ctm_out.empty()) { ctm.open(ctm_out); if (!ctm) K3H_ERR << "cannot open " << ctm_out; }
      const int result_type = cuda_decoder::CudaPipelineResult::RESULT_TYPE_LATTICE | (ctm_out.empty() ? 0 : cuda_decoder::CudaPipelineResult::RESULT_TYPE_CTM);
      {
        cuda_decoder::BatchedThreadedNnet3CudaPipeline2 pipeline(cfg, hfst, nnet, ti);
        if (!postproc.empty()) pipeline.SetLatticePostprocessor(LoadLatticePostprocessor(postproc));
        for (size_t i = 0; i < scp.size(); i++) {
          auto wave = std::make_shared<Wave>(ReadWave(scp[i].second)); const std::string key = scp[i].first;
          pipeline.SegmentedDecodeWithCallback(wave, [&writer, &wm, &n_seg, &ctm, &ctm_out, key](cuda_decoder::SegmentedLatticeCallbackParams &params) {
            std::lock_guard<std::mutex> lk(wm);
            if (!ctm_out.empty()) cuda_decoder::MergeSegmentsToCTMOutput(params.results, key, ctm);
            for (cuda_decoder::CudaPipelineResult &r : params.results) {
              std::ostringstream k; k << key << "-" << (double)r.GetTimeOffsetSeconds();
              if (!r.HasValidResult() || r.GetLatticeResult()->NumStates() == 0) {
                K3H_WARN << "Utterance " << key << ": segment with offset " << r.GetTimeOffsetSeconds() << " is not valid. Skipping";
                continue;
              }
              writer.WriteCompactLattice(k.str(), *r.GetLatticeResult()); n_seg++;
            }
          }, result_type);
        }
        pipeline.WaitForAllTasks();
      }
      writer.Flush(); k3_nnet_destroy(nnet);
      K3H_LOG << "Decoded " << scp.size() << " files in " << n_seg << " segments.";
      return 0;
    }
    std::vector<CompactLattice> results(scp.size()); std::vector<char> got(scp.size(), 0);
    {
      cuda_decoder::BatchedThreadedNnet3CudaPipeline2 pipeline(cfg, hfst, nnet, ti);
      pipeline.CreateTaskGroup("even"); pipeline.CreateTaskGroup("odd");
      for (size_t i = 0; i < scp.size(); i++) {
        auto wave = std::make_shared<Wave>(ReadWave(scp[i].second));
        pipeline.DecodeWithCallback(wave, [&results, &got, i](CompactLattice &clat) { results[i] = clat; got[i] = 1; }, i % 2 ? "odd" : "even");
      }
      pipeline.WaitForGroup("even");
      for (size_t i = 0; i < scp.size(); i += 2) if (!got[i]) K3H_ERR << "WaitForGroup returned before the group's callbacks ran";
      pipeline.WaitForAllTasks();
      pipeline.DestroyTaskGroup("even"); pipeline.DestroyTaskGroup("odd");
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
    K3H_LOG << "Decoded " << scp.size() << " utterances, " << n_err << " with errors.";
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
