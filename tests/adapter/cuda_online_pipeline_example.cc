// tests/adapter/cuda_online_pipeline_example.cc -- a caller of kaldi::cuda_decoder::BatchedThreadedNnet3CudaOnlinePipeline written against the REFERENCE's constructor, DecodeBatch
// (std::vector<SubVector<BaseFloat>> chunks per correlation id) and lattice-callback types (cudadecoder/batched-threaded-nnet3-cuda-online-pipeline.h:119-330), compiled against
// include/k3_batched_online_pipeline.h + the reference's own headers.  Like cudadecoderbin/batched-wav-nnet3-cuda-online.cc it plays the wave files of an scp as concurrent
// streams, one chunk of GetNSampsPerChunk() samples per stream and DecodeBatch call, and writes the lattices the callbacks received:
//   cuda-online-pipeline-example <final.mdl> <graph.bin> <wav.scp> <fbank.conf> <lattice-wspecifier>
#include <fstream>
#include <iostream>
#include <mutex>
#include "k3_batched_online_pipeline.h"
#include "nnet3/nnet-utils.h"
#include "util/common-utils.h"
using namespace kaldi; using namespace kaldi::cuda_decoder;
int main(int argc, char **argv) {
  if (argc != 6) { std::cerr << "usage: cuda-online-pipeline-example <final.mdl> <graph.bin> <wav.scp> <fbank.conf> <lattice-wspecifier>\n"; return 1; }
  try {
    TransitionModel trans_model; nnet3::AmNnetSimple am_nnet;
    { bool binary; Input ki(argv[1], &binary); trans_model.Read(ki.Stream(), binary); am_nnet.Read(ki.Stream(), binary);
      nnet3::SetBatchnormTestMode(true, &(am_nnet.GetNnet())); nnet3::SetDropoutTestMode(true, &(am_nnet.GetNnet())); }
    fst::VectorFst<fst::StdArc> decode_fst;
    { FILE *f = fopen(argv[2], "rb"); if (!f) KALDI_ERR << "cannot open " << argv[2];
      int32_t h[3]; if (fread(h, 4, 3, f) != 3) KALDI_ERR << "short read"; const int32_t S = h[0], start = h[1], A = h[2];
      std::vector<int32_t> off(S + 1), il(A), ol(A), nx(A); std::vector<float> w(A), fin(S);
      if (fread(off.data(), 4, S + 1, f) != (size_t)S + 1 || fread(il.data(), 4, A, f) != (size_t)A || fread(ol.data(), 4, A, f) != (size_t)A || fread(nx.data(), 4, A, f) != (size_t)A || fread(w.data(), 4, A, f) != (size_t)A || fread(fin.data(), 4, S, f) != (size_t)S) KALDI_ERR << "short read";
      fclose(f);
      for (int32_t s = 0; s < S; s++) decode_fst.AddState();
      decode_fst.SetStart(start);
      for (int32_t s = 0; s < S; s++) { decode_fst.SetFinal(s, fst::TropicalWeight(fin[s])); for (int32_t a = off[s]; a < off[s + 1]; a++) decode_fst.AddArc(s, fst::StdArc(il[a], ol[a], fst::TropicalWeight(w[a]), nx[a])); } }
    BatchedThreadedNnet3CudaOnlinePipelineConfig config;
    config.feature_opts.feature_type = "fbank"; config.feature_opts.fbank_config = argv[4]; config.compute_opts.acoustic_scale = 1.0; config.compute_opts.frame_subsampling_factor = 3; config.compute_opts.frames_per_chunk = 51;
    config.decoder_opts.default_beam = 15.0; config.decoder_opts.lattice_beam = 8.0; config.max_batch_size = 2; config.num_channels = 2; config.num_worker_threads = 2; config.determinize_lattice = false; config.max_utterance_frames = 400;
    BatchedThreadedNnet3CudaOnlinePipeline cuda_pipeline(config, decode_fst, am_nnet, trans_model);
    std::vector<std::pair<std::string, std::string> > scp; { std::ifstream in(argv[3]); std::string k, p; while (in >> k >> p) scp.push_back({k, p}); }
    std::vector<WaveData> waves(scp.size()); for (size_t i = 0; i < scp.size(); i++) { bool binary; Input ki(scp[i].second, &binary); waves[i].Read(ki.Stream()); }
    std::vector<k3host::CompactLattice> results(scp.size()); std::mutex m; std::vector<size_t> pos(scp.size(), 0); std::vector<char> started(scp.size(), 0), done(scp.size(), 0);
    const int32 chunk = cuda_pipeline.GetNSampsPerChunk(); size_t n_done = 0; int32 n_partial = 0;
    while (n_done < scp.size()) {      // streams are admitted while channels are free (TryInitCorrID), every active stream gets one chunk per round
      std::vector<BatchedThreadedNnet3CudaOnlinePipeline::CorrelationID> ids; std::vector<SubVector<BaseFloat> > chunks; std::vector<bool> first, last;
      for (size_t i = 0; i < scp.size() && (int32)ids.size() < config.max_batch_size; i++) {
        if (done[i]) continue;
        if (!started[i]) { if (!cuda_pipeline.TryInitCorrID(i)) continue; started[i] = 1;
          cuda_pipeline.SetLatticeCallback(i, [&results, &m, i](CompactLattice &clat) { std::lock_guard<std::mutex> lk(m); BatchedThreadedNnet3CudaPipeline2::FromKaldi(clat, &results[i]); }); }
        const SubVector<BaseFloat> all(waves[i].Data(), 0); const int32 n = std::min<int32>(chunk, all.Dim() - (int32)pos[i]);
        ids.push_back(i); chunks.push_back(SubVector<BaseFloat>(all, (int32)pos[i], n)); first.push_back(pos[i] == 0); pos[i] += n; last.push_back((int32)pos[i] == all.Dim());
        if (last.back()) { done[i] = 1; n_done++; }
      }
      std::vector<const std::string *> partial;
      cuda_pipeline.DecodeBatch(ids, chunks, first, last, &partial);
      for (const std::string *p : partial) if (p && !p->empty()) n_partial++;
    }
    cuda_pipeline.WaitForLatticeCallbacks();
    k3host::TableWriter writer(argv[5]); int32 n_err = 0;
    for (size_t i = 0; i < scp.size(); i++) { if (results[i].NumStates() == 0) { n_err++; continue; } writer.WriteCompactLattice(scp[i].first, results[i]); }
    writer.Flush();
    KALDI_LOG << "Decoded " << scp.size() << " utterances, " << n_err << " with errors; " << n_partial << " non-empty partial hypotheses on the way.";
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
