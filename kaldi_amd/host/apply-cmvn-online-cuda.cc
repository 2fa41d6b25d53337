// apply-cmvn-online-cuda -- drop-in for cudafeatbin/apply-cmvn-online-cuda.cc (and, under the name apply-cmvn-online, for
// online2bin/apply-cmvn-online.cc:28-140 including its --spk2utt option):
//   apply-cmvn-online-cuda [options] <global-cmvn-stats> <feature-rspecifier> <feature-wspecifier>
// Options = OnlineCmvnOptions::Register (feat/online-feature.h:231-247).  All utterances go through ONE k3_cmvn_online_batch call; with
// --spk2utt the speaker statistics an utterance starts from (OnlineCmvn::GetState, feat/online-feature.cc:470-486: sums over every frame
// of the speaker's earlier utterances) are accumulated on the host in double and handed to the kernel per utterance.
#include <hip/hip_runtime.h>
#include <iostream>
#include <map>
#include "k3_host.h"
#include "../../include/k3hip.h"
using namespace k3host;
#define HIPCHK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)

int main(int argc, char **argv) {
  try {
    const char *usage =
        "Apply online cepstral mean (and possibly variance) computation online (GPU).\n"
        "Usage: apply-cmvn-online-cuda [options] <global-cmvn-stats> <feature-rspecifier> <feature-wspecifier>\n"
        "e.g. apply-cmvn-online-cuda 'matrix-sum scp:data/train/cmvn.scp -|' data/train/split8/1/feats.scp ark:-\n";
    ParseOptions po(usage);
    k3_online_cmvn_opts o; k3_online_cmvn_opts_default(&o);
    bool norm_vars = false, norm_means = true; std::string skip_dims_str, spk2utt;
    po.Register("cmn-window", &o.cmn_window, "Number of frames of sliding context for cepstral mean normalization.");
    po.Register("global-frames", &o.global_frames, "Number of frames of global-average cepstral mean normalization stats to use for first utterance of a speaker");
    po.Register("speaker-frames", &o.speaker_frames, "Number of frames of previous utterance(s) from this speaker to use in cepstral mean normalization");
    po.Register("norm-vars", &norm_vars, "If true, do cepstral variance normalization in addition to cepstral mean normalization ");
    po.Register("norm-means", &norm_means, "If true, do mean normalization (note: you cannot normalize the variance but not the mean)");
    po.Register("skip-dims", &skip_dims_str, "Dimensions to skip normalization of (colon-separated list of integers)");
    po.Register("spk2utt", &spk2utt, "rspecifier for speaker to utterance-list map");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) { po.PrintUsage(); return 1; }
    o.normalize_mean = norm_means; o.normalize_variance = norm_vars;
    std::vector<int32_t> skip;
    for (size_t p = 0; p < skip_dims_str.size();) {
      size_t q = skip_dims_str.find(':', p); if (q == std::string::npos) q = skip_dims_str.size();
      char *e = nullptr; const std::string t = skip_dims_str.substr(p, q - p); const long v = strtol(t.c_str(), &e, 10);
      if (t.empty() || *e) K3H_ERR << "Bad --skip-dims option (should be colon-separated list of integers)";
      skip.push_back((int32_t)v); p = q + 1;
    }
    const MatrixD gstats = ReadDoubleMatrix(po.GetArg(1));
    if (gstats.rows != 2 || gstats.cols < 2) K3H_ERR << "Bad global CMVN stats: " << gstats.rows << " x " << gstats.cols;
    const int32_t dim = gstats.cols - 1;
    auto table = ReadMatrixTable(po.GetArg(2)); TableWriter writer(po.GetArg(3));
    // processing order + starting speaker stats
    std::vector<size_t> order; std::vector<double> spk_stats;      // [n x 2 x (dim+1)] when --spk2utt is given
    if (!spk2utt.empty()) {
      std::map<std::string, size_t> index; for (size_t i = 0; i < table.size(); i++) index[table[i].first] = i;
      for (auto &sp : ReadTokenVectorTable(spk2utt)) {
        std::vector<double> acc(2 * (size_t)(dim + 1), 0.0);
        for (auto &utt : sp.second) {
          auto it = index.find(utt);
          if (it == index.end()) { K3H_WARN << "No features for utterance " << utt; continue; }
          const Matrix &m = table[it->second].second;
          if (m.cols != dim) K3H_ERR << "Dim mismatch: cmvn stats " << dim << " vs features " << m.cols << " for " << utt;
          order.push_back(it->second); spk_stats.insert(spk_stats.end(), acc.begin(), acc.end());
          for (int32_t t = 0; t < m.rows; t++) {
            for (int32_t d = 0; d < dim; d++) { const double x = m.data[(size_t)t * dim + d]; acc[d] += x; acc[dim + 1 + d] += x * x; }
            acc[dim] += 1.0;
          }
        }
      }
    } else for (size_t i = 0; i < table.size(); i++) order.push_back(i);
    int32_t num_done = 0; int64_t tot_t = 0;
    if (!order.empty()) {
      std::vector<int64_t> foff(1, 0); std::vector<float> all;
      for (size_t i : order) {
        const Matrix &m = table[i].second;
        if (m.cols != dim) K3H_ERR << "Dim mismatch: cmvn stats " << dim << " vs features " << m.cols << " for " << table[i].first;
        all.insert(all.end(), m.data.begin(), m.data.end()); foff.push_back(foff.back() + m.rows);
      }
      float *d_in, *d_out; int64_t *d_fo; double *d_g, *d_s = nullptr; const size_t nel = all.size();
      HIPCHK(hipMalloc((void **)&d_in, std::max<size_t>(4, nel * 4))); HIPCHK(hipMalloc((void **)&d_out, std::max<size_t>(4, nel * 4)));
      HIPCHK(hipMalloc((void **)&d_fo, foff.size() * 8)); HIPCHK(hipMalloc((void **)&d_g, gstats.data.size() * 8));
      HIPCHK(hipMemcpy(d_in, all.data(), nel * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_fo, foff.data(), foff.size() * 8, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d_g, gstats.data.data(), gstats.data.size() * 8, hipMemcpyHostToDevice));
      if (!spk_stats.empty()) { HIPCHK(hipMalloc((void **)&d_s, spk_stats.size() * 8)); HIPCHK(hipMemcpy(d_s, spk_stats.data(), spk_stats.size() * 8, hipMemcpyHostToDevice)); }
      K3H_CHECK_K3(k3_cmvn_online_batch(d_in, dim, d_out, dim, dim, d_fo, (int32_t)order.size(), &o, d_g, d_s, skip.data(), (int32_t)skip.size(), nullptr));
      HIPCHK(hipMemcpy(all.data(), d_out, nel * 4, hipMemcpyDeviceToHost));
      for (size_t k = 0; k < order.size(); k++) {
        const int32_t rows = (int32_t)(foff[k + 1] - foff[k]);
        writer.WriteMatrix(table[order[k]].first, all.data() + foff[k] * dim, rows, dim, dim); num_done++; tot_t += rows;
      }
      HIPCHK(hipFree(d_in)); HIPCHK(hipFree(d_out)); HIPCHK(hipFree(d_fo)); HIPCHK(hipFree(d_g)); if (d_s) HIPCHK(hipFree(d_s));
    }
    writer.Flush();
    K3H_LOG << "Applied online CMVN to " << num_done << " files, or " << tot_t << " frames.";
    return num_done != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
