// nnet3-compute -- drop-in for nnet3bin/nnet3-compute.cc:33-200 with the forward pass on MI355X:
//   nnet3-compute [options] <nnet-in> <features-rspecifier> <matrix-wspecifier>
// Outputs one matrix per utterance with ceil(T / frame-subsampling-factor) rows (nnet-am-decodable-simple.cc:45-47);
// all utterances of the table go through k3_nnet_forward in batches of --max-batch-size.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <iostream>
#include "k3_host.h"
#include "k3_nnet_ivector_cli.h"
#include "../../include/k3hip.h"
using namespace k3host;
#define HIPCHK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)
int main(int argc, char **argv) {
  try {
    const char *usage = "Propagate the features through raw neural network model and write the output.\n"
                        "Usage: nnet3-compute [options] <nnet-in> <features-rspecifier> <matrix-wspecifier>\n e.g.: nnet3-compute final.raw scp:feats.scp ark:nnet_prediction.ark\n";
    ParseOptions po(usage);
    bool apply_exp = false, use_priors = false, debug_comp = false; std::string use_gpu = "yes", ivector_rspecifier, online_ivector_rspecifier, utt2spk;
    int32_t subsampling = 1, frames_per_chunk = 50, elc = 0, erc = 0, elci = -1, ercf = -1, online_ivector_period = 0, max_batch = 512; float acoustic_scale = 1.0f;
    po.Register("apply-exp", &apply_exp, "If true, apply exp function to output");
    po.Register("use-priors", &use_priors, "If true, subtract the logs of the priors stored with the model (in this case, a .mdl file is expected as input).");
    po.Register("use-gpu", &use_gpu, "yes|no|optional|wait (this build always uses the GPU)");
    po.Register("frame-subsampling-factor", &subsampling, "Required if the frame-rate of the output is less than the frame-rate of the input");
    po.Register("frames-per-chunk", &frames_per_chunk,
        "Number of frames in each chunk that is separately evaluated by the neural net (matters with --online-ivectors: one i-vector per chunk; without i-vectors the result does not depend on it and utterances are evaluated whole)");
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic log-likelihoods");
    po.Register("extra-left-context", &elc, "(only 0 is supported)");
    po.Register("extra-right-context", &erc, "(only 0 is supported)");
    po.Register("extra-left-context-initial", &elci, "(accepted)");
    po.Register("extra-right-context-final", &ercf, "(accepted)");
    po.Register("ivectors", &ivector_rspecifier,
        "Rspecifier for iVectors as vectors (i.e. not estimated online); per utterance by default, or per speaker if you provide the --utt2spk option.");
    po.Register("online-ivectors", &online_ivector_rspecifier,
        "Rspecifier for iVectors estimated online, as matrices.  If you supply this, you must set the --online-ivector-period option.");
    po.Register("online-ivector-period", &online_ivector_period, "Number of frames between iVectors in matrices supplied to the --online-ivectors option");
    po.Register("utt2spk", &utt2spk, "Rspecifier for utt2spk option used to get ivectors per speaker");
    po.Register("debug-computation", &debug_comp, "(accepted, unused)");
    po.Register("max-batch-size", &max_batch, "Utterances per GPU batch");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) { po.PrintUsage(); return 1; }
    if (elc || erc) K3H_ERR << "extra context is not supported by this program (feed-forward TDNN / TDNN-F models do not use it)";
    IvectorInputs iv; iv.Open(ivector_rspecifier, online_ivector_rspecifier, utt2spk, online_ivector_period);
    k3_nnet *nnet = nullptr; K3H_CHECK_K3(k3_nnet_load(po.GetArg(1).c_str(), &nnet));
    k3_nnet_info ni; K3H_CHECK_K3(k3_nnet_get_info(nnet, &ni));
    std::vector<float> log_priors;
    if (use_priors) {
      if (!ni.has_priors) K3H_ERR << "Priors vector is empty (a .mdl with priors is expected with --use-priors)";
      log_priors.resize(ni.output_dim);
      K3H_CHECK_K3(k3_nnet_get_priors(nnet, log_priors.data()));
      for (float &p : log_priors) p = logf(p);
    }
    auto feats = ReadMatrixTable(po.GetArg(2)); TableWriter writer(po.GetArg(3));
    int num_success = 0, num_fail = 0; int64_t frame_count = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t b0 = 0; b0 < feats.size(); b0 += max_batch) {
      const size_t b1 = std::min(feats.size(), b0 + (size_t)max_batch);
      std::vector<size_t> idx; std::vector<int32_t> nf; std::vector<float> all; std::vector<const Matrix *> utt_iv;
      for (size_t i = b0; i < b1; i++) {
        const Matrix &m = feats[i].second;
        if (m.rows == 0) { K3H_WARN << "Zero-length utterance: " << feats[i].first; num_fail++; continue; }
        if (m.cols != ni.input_dim) K3H_ERR << "Neural net expects 'input' features with dimension " << ni.input_dim << " but you provided " << m.cols;
        const Matrix *v = iv.Any() ? iv.Get(feats[i].first) : nullptr;
        if (iv.Any() && !v) { K3H_WARN << "No iVector available for utterance " << feats[i].first; num_fail++; continue; }
        idx.push_back(i); nf.push_back(m.rows); all.insert(all.end(), m.data.begin(), m.data.end()); utt_iv.push_back(v);
      }
      if (idx.empty()) continue;
      k3_nnet_batch *nb = nullptr; std::vector<int64_t> ro; float *d_o = nullptr;
      RunNnetBatch(nnet, ni, nf, all, iv, utt_iv, subsampling, frames_per_chunk, log_priors, acoustic_scale, &nb, &ro, &d_o);
      const int64_t rows = ro.back();
      std::vector<float> h((size_t)rows * ni.output_dim); HIPCHK(hipMemcpy(h.data(), d_o, h.size() * 4, hipMemcpyDeviceToHost));
      if (apply_exp) for (float &v : h) v = expf(v);
      for (size_t u = 0; u < idx.size(); u++) {
        writer.WriteMatrix(feats[idx[u]].first, h.data() + ro[u] * ni.output_dim, (int32_t)(ro[u + 1] - ro[u]), ni.output_dim, ni.output_dim);
        frame_count += nf[u];
        num_success++;
      }
      k3_nnet_batch_destroy(nb); HIPCHK(hipFree(d_o));
    }
    writer.Flush();
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    K3H_LOG << "Time taken " << elapsed << "s: real-time factor assuming 100 frames/sec is " << (elapsed * 100.0 / std::max<int64_t>(frame_count, 1));
    K3H_LOG << "Done " << num_success << " utterances, failed for " << num_fail;
    k3_nnet_destroy(nnet);
    return num_success != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
