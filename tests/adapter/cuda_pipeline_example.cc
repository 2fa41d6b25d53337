// tests/adapter/cuda_pipeline_example.cc -- a caller of kaldi::cuda_decoder::BatchedThreadedNnet3CudaPipeline2 written against the REFERENCE's constructor
// (cudadecoder/batched-threaded-nnet3-cuda-pipeline2.h:153-156: config, fst::Fst<fst::StdArc>, nnet3::AmNnetSimple, TransitionModel) and callback type (CompactLattice &), compiled
// against include/k3_batched_pipeline.h + the reference's own headers.  It reads final.mdl with the reference's readers (the way cudadecoderbin/batched-wav-nnet3-cuda2.cc:138-148
// does), builds the graph object arc by arc (the OpenFst file reader is not part of the stand-in), decodes the wave files of an scp through DecodeWithCallback in two task groups
// and writes the lattices the callbacks received:    cuda-pipeline-example <final.mdl> <graph.bin> <wav.scp> <fbank.conf> <lattice-wspecifier>
#include <fstream>
#include <iostream>
#include <mutex>
#include "k3_batched_pipeline.h"
#include "nnet3/nnet-utils.h"
#include "util/common-utils.h"
using namespace kaldi; using namespace kaldi::cuda_decoder;
int main(int argc, char **argv) {
  if (argc != 6) { std::cerr << "usage: cuda-pipeline-example <final.mdl> <graph.bin> <wav.scp> <fbank.conf> <lattice-wspecifier>\n"; return 1; }
  try {
    TransitionModel trans_model; nnet3::AmNnetSimple am_nnet;
    { bool binary; Input ki(argv[1], &binary); trans_model.Read(ki.Stream(), binary); am_nnet.Read(ki.Stream(), binary);
      nnet3::SetBatchnormTestMode(true, &(am_nnet.GetNnet())); nnet3::SetDropoutTestMode(true, &(am_nnet.GetNnet())); }      // (:144-146; CollapseModel is an optimisation of the reference's own executor: the fused path folds batch-norm itself)
    fst::VectorFst<fst::StdArc> decode_fst;
    { FILE *f = fopen(argv[2], "rb"); if (!f) KALDI_ERR << "cannot open " << argv[2];
      int32_t h[3]; if (fread(h, 4, 3, f) != 3) KALDI_ERR << "short read"; const int32_t S = h[0], start = h[1], A = h[2];
      std::vector<int32_t> off(S + 1), il(A), ol(A), nx(A); std::vector<float> w(A), fin(S);
      if (fread(off.data(), 4, S + 1, f) != (size_t)S + 1 || fread(il.data(), 4, A, f) != (size_t)A || fread(ol.data(), 4, A, f) != (size_t)A || fread(nx.data(), 4, A, f) != (size_t)A || fread(w.data(), 4, A, f) != (size_t)A || fread(fin.data(), 4, S, f) != (size_t)S) KALDI_ERR << "short read";
      fclose(f);
      for (int32_t s = 0; s < S; s++) decode_fst.AddState();
      decode_fst.SetStart(start);
      for (int32_t s = 0; s < S; s++) { decode_fst.SetFinal(s, fst::TropicalWeight(fin[s])); for (int32_t a = off[s]; a < off[s + 1]; a++) decode_fst.AddArc(s, fst::StdArc(il[a], ol[a], fst::TropicalWeight(w[a]), nx[a])); } }
    BatchedThreadedNnet3CudaPipeline2Config config;
    config.feature_opts.feature_type = "fbank"; config.feature_opts.fbank_config = argv[4]; config.compute_opts.acoustic_scale = 1.0; config.compute_opts.frame_subsampling_factor = 3;
    config.decoder_opts.default_beam = 15.0; config.decoder_opts.lattice_beam = 8.0; config.max_batch_size = 2; config.num_worker_threads = 2;
    BatchedThreadedNnet3CudaPipeline2 cuda_pipeline(config, decode_fst, am_nnet, trans_model);
    std::vector<std::pair<std::string, std::string> > scp; { std::ifstream in(argv[3]); std::string k, p; while (in >> k >> p) scp.push_back({k, p}); }
    std::vector<k3host::CompactLattice> results(scp.size()); std::mutex m;
    cuda_pipeline.CreateTaskGroup("all");
    for (size_t i = 0; i < scp.size(); i++) {
      std::shared_ptr<WaveData> wave_data = std::make_shared<WaveData>();
      { bool binary; Input ki(scp[i].second, &binary); wave_data->Read(ki.Stream()); }
      cuda_pipeline.DecodeWithCallback(wave_data, [&results, &m, i](CompactLattice &clat) { std::lock_guard<std::mutex> lk(m); BatchedThreadedNnet3CudaPipeline2::FromKaldi(clat, &results[i]); }, "all");
    }
    cuda_pipeline.WaitForGroup("all"); cuda_pipeline.DestroyTaskGroup("all"); cuda_pipeline.WaitForAllTasks();
    k3host::TableWriter writer(argv[5]); int32 n_err = 0;
    for (size_t i = 0; i < scp.size(); i++) { if (results[i].NumStates() == 0) { n_err++; continue; } writer.WriteCompactLattice(scp[i].first, results[i]); }
    writer.Flush();
    KALDI_LOG << "Decoded " << scp.size() << " utterances, " << n_err << " with errors.";
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
