// k3_nnet_ivector_cli.h -- the i-vector side of nnet3-compute / nnet3-latgen-faster (nnet3bin/nnet3-compute.cc:110-168,
// nnet3bin/nnet3-latgen-faster.cc:133-185): --ivectors (one vector per utterance, or per speaker with --utt2spk) or --online-ivectors (a matrix
// per utterance, one row per --online-ivector-period frames), and the forward pass of one batch with or without them.
#ifndef K3_NNET_IVECTOR_CLI_H_
#define K3_NNET_IVECTOR_CLI_H_
#include <hip/hip_runtime.h>
#include <map>
#include "k3_host.h"
#include "../../include/k3hip.h"
namespace k3host {

struct IvectorInputs {
  bool per_utt = false, online = false; int32_t period = 0;
  std::map<std::string, Matrix> table; std::map<std::string, std::string> utt2spk;
  void Open(const std::string &ivector_rspecifier, const std::string &online_ivector_rspecifier, const std::string &utt2spk_rspecifier, int32_t online_ivector_period) {
    if (!ivector_rspecifier.empty() && !online_ivector_rspecifier.empty()) K3H_ERR << "You cannot specify both --ivectors and --online-ivectors";      // the reference asserts this
    per_utt = !ivector_rspecifier.empty(); online = !online_ivector_rspecifier.empty(); period = online_ivector_period;
    if (!per_utt && !online) return;
    if (online && period <= 0) K3H_ERR << "--online-ivector-period must be set (> 0) with --online-ivectors";
    for (auto &kv : ReadMatrixTable(per_utt ? ivector_rspecifier : online_ivector_rspecifier)) table[kv.first] = std::move(kv.second);
    if (!utt2spk_rspecifier.empty()) for (auto &kv : ReadTokenVectorTable(utt2spk_rspecifier)) if (!kv.second.empty()) utt2spk[kv.first] = kv.second[0];
  }
  bool Any() const { return per_utt || online; }
  // the rows of one utterance, nullptr when the table has none (the caller warns "No iVector available for utterance" and skips it)
  const Matrix *Get(const std::string &utt) const {
    std::string key = utt;
    if (per_utt && !utt2spk.empty()) { auto it = utt2spk.find(utt); if (it == utt2spk.end()) return nullptr; key = it->second; }
    auto it = table.find(key); return it == table.end() ? nullptr : &it->second;
  }
};

// plans and runs one batch; the caller owns *nb (k3_nnet_batch_destroy) and d_out (allocated here: [rows x output_dim])
inline void RunNnetBatch(k3_nnet *nnet, const k3_nnet_info &ni, const std::vector<int32_t> &nf, const std::vector<float> &all_feats, const IvectorInputs &iv,
                         const std::vector<const Matrix *> &utt_iv, int32_t subsampling, int32_t frames_per_chunk, const std::vector<float> &log_priors, float acoustic_scale,
                         k3_nnet_batch **nb, std::vector<int64_t> *row_offsets, float **d_out) {
#define K3N_HIP(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)
  const int32_t U = (int32_t)nf.size(); const float *lp = log_priors.empty() ? nullptr : log_priors.data();
  std::vector<float> ivs; std::vector<int32_t> iv_rows;
  if (ni.ivector_dim > 0) {
    if (!iv.Any()) K3H_ERR << "Neural net expects 'ivector' features with dimension " << ni.ivector_dim << " but you provided 0";      // nnet-am-decodable-simple.cc:105-107
    for (int32_t u = 0; u < U; u++) {
      const Matrix &m = *utt_iv[u];
      if (m.cols != ni.ivector_dim) K3H_ERR << "Neural net expects 'ivector' features with dimension " << ni.ivector_dim << " but you provided " << m.cols;
      if (iv.per_utt && m.rows != 1) K3H_ERR << "--ivectors must be a table of vectors";
      iv_rows.push_back(m.rows); ivs.insert(ivs.end(), m.data.begin(), m.data.end());
    }
    K3H_CHECK_K3(k3_nnet_batch_create_ivector(nnet, U, nf.data(), subsampling, lp, acoustic_scale, frames_per_chunk, iv.online ? iv.period : 0,
        iv.online ? iv_rows.data() : nullptr, nb));
  } else {
    if (iv.Any()) K3H_ERR << "Neural net expects 'ivector' features with dimension 0 but you provided " << (utt_iv.empty() || !utt_iv[0] ? 0 : utt_iv[0]->cols);
    K3H_CHECK_K3(k3_nnet_batch_create(nnet, U, nf.data(), subsampling, lp, acoustic_scale, nb));
  }
  row_offsets->assign(U + 1, 0); const int64_t rows = k3_nnet_batch_output_rows(*nb, row_offsets->data());
  float *d_f = nullptr, *d_iv = nullptr;
  K3N_HIP(hipMalloc((void **)&d_f, all_feats.size() * 4)); K3N_HIP(hipMalloc((void **)d_out, (size_t)rows * ni.output_dim * 4));
  K3N_HIP(hipMemcpy(d_f, all_feats.data(), all_feats.size() * 4, hipMemcpyHostToDevice));
  if (ni.ivector_dim > 0) {
    K3N_HIP(hipMalloc((void **)&d_iv, ivs.size() * 4)); K3N_HIP(hipMemcpy(d_iv, ivs.data(), ivs.size() * 4, hipMemcpyHostToDevice));
    K3H_CHECK_K3(k3_nnet_forward_ivector(*nb, d_f, ni.input_dim, d_iv, ni.ivector_dim, *d_out, ni.output_dim, nullptr));
  } else K3H_CHECK_K3(k3_nnet_forward(*nb, d_f, ni.input_dim, *d_out, ni.output_dim, nullptr));
  K3N_HIP(hipDeviceSynchronize()); K3N_HIP(hipFree(d_f)); if (d_iv) K3N_HIP(hipFree(d_iv));
#undef K3N_HIP
}

}  // namespace k3host
#endif
