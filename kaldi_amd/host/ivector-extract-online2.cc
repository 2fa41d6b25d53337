// ivector-extract-online2 -- drop-in for online2bin/ivector-extract-online2.cc:32-200 with the extraction on MI355X:
//   ivector-extract-online2 [options] <spk2utt-rspecifier> <feature-rspecifier> <ivector-wspecifier>
// One matrix per utterance, a row per --ivector-period frames (--repeat=true: a row per frame).  The adaptation state (CMVN and i-vector
// statistics) is carried from one utterance of a speaker to the next like the reference does; the utterances of a batch are the k-th
// utterances of all speakers ("round k"), so the GPU sees as many independent utterances as there are speakers.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <iostream>
#include <map>
#include "k3_host.h"
#include "../../include/k3hip.h"
using namespace k3host;
#define HIPCHK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)
int main(int argc, char **argv) {
  try {
    const char *usage =
        "Extract iVectors for utterances every --ivector-period frames, using a trained\niVector extractor and features and Gaussian-level posteriors.  Similar to\n"
        "ivector-extract-online but uses the actual online decoder code to do it,\nand does everything in-memory instead of using multiple processes.\n"
        "Note: the value of the --use-most-recent-ivector config variable is ignored;\nit's set to false.  The <spk2utt-rspecifier> is mandatory, to simplify the code;\n"
        "if you want to do it separately per utterance, just make it of the form\n<utterance-id> <utterance-id>.\nThe iVectors are output as an archive of matrices, indexed by utterance-id;\n"
        "each row corresponds to an iVector.  If --repeat=true, outputs the whole matrix\nof iVectors, not just every (ivector-period)'th frame\n"
        "Usage:  ivector-extract-online2 [options] <spk2utt-rspecifier> <feature-rspecifier> <ivector-wspecifier>\n"
        "e.g.: \n  ivector-extract-online2 --config=exp/nnet2_online/nnet_online/conf/ivector_extractor.conf \\\n    ark:data/train/spk2utt scp:data/train/feats.scp ark,t:ivectors.1.ark\n";
    ParseOptions po(usage);
    IvectorExtractionInfo info; info.Register(&po);
    int32_t length_tolerance = 0, max_batch = 512; bool repeat = false, exact_solve = false; std::string frame_weights;
    po.Register("repeat", &repeat, "If true, output the same number of iVectors as input frames (including repeated data).");
    po.Register("frame-weights-rspecifier", &frame_weights, "Archive of frame weights to scale stats");
    po.Register("length-tolerance", &length_tolerance, "Tolerance on the difference in number of frames for feats and weights");
    po.Register("max-batch-size", &max_batch, "Utterances per GPU batch");
    po.Register("exact-solve", &exact_solve, "Solve for the iVector directly (Cholesky) instead of by conjugate gradient, like the GPU reference");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) { po.PrintUsage(); return 1; }
    info.use_most_recent_ivector = false;
    info.Init();
    k3_ivector_model m; memset(&m, 0, sizeof m);
    m.feat_dim = info.global_cmvn_stats.cols - 1; m.lda_rows = info.lda_rows; m.lda_cols = info.lda_cols; m.num_gauss = info.ubm.num_gauss; m.ivector_dim = info.ie.ivector_dim;
    m.lda = info.lda.data();
    m.global_cmvn_stats = info.global_cmvn_stats.data.data();
    m.gconsts = info.ubm.gconsts.data();
    m.means_invvars = info.ubm.means_invvars.data();
    m.inv_vars = info.ubm.inv_vars.data();
    m.M = info.ie.M.data(); m.sigma_inv = info.ie.sigma_inv.data(); m.prior_offset = info.ie.prior_offset;
    k3_ivector_opts o; k3_ivector_opts_default(&o);
    o.left_context = info.left_context;
    o.right_context = info.right_context;
    o.num_gselect = info.num_gselect;
    o.min_post = info.min_post;
    o.posterior_scale = info.posterior_scale;
    o.max_count = info.max_count;
    o.ivector_period = info.ivector_period; o.num_cg_iters = info.num_cg_iters; o.exact_solve = exact_solve; o.online_cmvn_iextractor = info.online_cmvn_iextractor;
    o.cmvn.cmn_window = info.cmn_window;
    o.cmvn.speaker_frames = info.speaker_frames;
    o.cmvn.global_frames = info.global_frames;
    o.cmvn.normalize_mean = info.normalize_mean;
    o.cmvn.normalize_variance = info.normalize_variance;
    k3_ivector *iv = nullptr; K3H_CHECK_K3(k3_ivector_create(&m, &o, &iv));
    // --repeat=true: the adaptation state handed to the speaker's next utterance holds every frame (GetFrame(T - 1), :121-127)
    k3_ivector_set_accumulate_tail(iv, repeat ? 1 : 0);
    const int32_t F = m.feat_dim, R = m.ivector_dim, P = info.ivector_period; const int64_t SS = k3_ivector_stats_size(iv);
    auto table = ReadMatrixTable(po.GetArg(2)); TableWriter writer(po.GetArg(3));
    std::map<std::string, size_t> index; for (size_t i = 0; i < table.size(); i++) index[table[i].first] = i;
    std::map<std::string, std::vector<float>> weights_of;      // --frame-weights-rspecifier: a vector per utterance (:130-153)
    if (!frame_weights.empty()) for (auto &kv : ReadMatrixTable(frame_weights)) weights_of[kv.first] = std::move(kv.second.data);
    // speakers -> the utterances that exist, in spk2utt order
    struct Spk { std::vector<size_t> utts; std::vector<double> cmvn, stats; bool has_stats = false; };
    std::vector<Spk> spks; int32_t num_done = 0, num_err = 0; bool warned_dim = false; size_t rounds = 0;
    for (auto &sp : ReadTokenVectorTable(po.GetArg(1))) {
      Spk s; s.cmvn.assign(2 * (size_t)(F + 1), 0.0);
      for (auto &utt : sp.second) {
        auto it = index.find(utt);
        if (it == index.end()) {
          K3H_WARN << "Did not find audio for utterance " << utt;
          num_err++;
          continue;
        }
        s.utts.push_back(it->second);
      }
      rounds = std::max(rounds, s.utts.size()); spks.push_back(std::move(s));
    }
    double tot_t = 0, tot_length = 0, tot_length_end = 0;
    for (size_t r = 0; r < rounds; r++) {
      std::vector<size_t> who; for (size_t k = 0; k < spks.size(); k++) if (r < spks[k].utts.size()) who.push_back(k);
      for (size_t b0 = 0; b0 < who.size(); b0 += max_batch) {
        const size_t U = std::min(who.size() - b0, (size_t)max_batch);
        std::vector<int64_t> fo(1, 0); std::vector<float> all, fw; std::vector<double> cm, st; bool any_state = false; std::vector<size_t> keep;
        for (size_t k = 0; k < U; k++) {
          Spk &s = spks[who[b0 + k]]; const Matrix &f = table[s.utts[r]].second; int32_t dim = f.cols;
          if (dim == F + 3) {
            if (!warned_dim) {
              K3H_WARN << "Feature dimension is too large by 3, assuming there are pitch features and removing the last 3 dims.";
              warned_dim = true;
            }
            dim -= 3;
          }
          if (dim != F) K3H_ERR << "Feature dimension " << f.cols << " does not match the extractor's " << F << " for utterance " << table[s.utts[r]].first;
          if (f.rows == 0) { K3H_WARN << "Empty feature matrix for utterance " << table[s.utts[r]].first; num_err++; continue; }
          if (!frame_weights.empty()) {      // frames past the end of the weights weigh 0; a length off by more than --length-tolerance is an error (:137-149)
            auto w = weights_of.find(table[s.utts[r]].first);
            if (w == weights_of.end()) { K3H_WARN << "Did not find weights for utterance " << table[s.utts[r]].first; num_err++; continue; }
            if (std::abs((int32_t)w->second.size() - f.rows) > length_tolerance) { num_err++; continue; }
            for (int32_t t = 0; t < f.rows; t++) fw.push_back(t < (int32_t)w->second.size() ? w->second[t] : 0.0f);
          }
          for (int32_t t = 0; t < f.rows; t++) all.insert(all.end(), f.data.begin() + (size_t)t * f.cols, f.data.begin() + (size_t)t * f.cols + F);
          fo.push_back(fo.back() + f.rows); cm.insert(cm.end(), s.cmvn.begin(), s.cmvn.end()); any_state = any_state || s.has_stats; keep.push_back(who[b0 + k]);
        }
        const size_t n = keep.size(); if (!n) continue;
        if (any_state) for (size_t k = 0; k < n; k++) {            // a record per utterance; speakers on their first utterance get the fresh statistics
          Spk &s = spks[keep[k]];
          if (s.has_stats) st.insert(st.end(), s.stats.begin(), s.stats.end());
          else {
            std::vector<double> fresh(SS, 0.0);
            fresh[1] = m.prior_offset;
            for (int32_t i = 0; i < R; i++) fresh[1 + R + (size_t)i * R + i] = 1.0;
            st.insert(st.end(), fresh.begin(), fresh.end());
          }
        }
        std::vector<int64_t> ro(n + 1); const int64_t rows = k3_ivector_num_rows(iv, (int32_t)n, fo.data(), ro.data());
        float *d_f, *d_iv, *d_fw = nullptr; double *d_cm, *d_si = nullptr, *d_so;
        HIPCHK(hipMalloc((void **)&d_f, all.size() * 4));
        HIPCHK(hipMalloc((void **)&d_iv, (size_t)rows * R * 4));
        HIPCHK(hipMalloc((void **)&d_cm, cm.size() * 8));
        HIPCHK(hipMalloc((void **)&d_so, n * SS * 8));
        HIPCHK(hipMemcpy(d_f, all.data(), all.size() * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_cm, cm.data(), cm.size() * 8, hipMemcpyHostToDevice));
        if (any_state) { HIPCHK(hipMalloc((void **)&d_si, st.size() * 8)); HIPCHK(hipMemcpy(d_si, st.data(), st.size() * 8, hipMemcpyHostToDevice)); }
        if (!frame_weights.empty()) { HIPCHK(hipMalloc((void **)&d_fw, fw.size() * 4)); HIPCHK(hipMemcpy(d_fw, fw.data(), fw.size() * 4, hipMemcpyHostToDevice)); }
        K3H_CHECK_K3(k3_ivector_extract_batch_weighted(iv, d_f, F, fo.data(), (int32_t)n, d_fw, d_iv, R, d_cm, d_si, d_so, nullptr));
        std::vector<float> h((size_t)rows * R); std::vector<double> so(n * SS);
        HIPCHK(hipMemcpy(h.data(), d_iv, h.size() * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(so.data(), d_so, so.size() * 8, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < n; k++) {
          Spk &s = spks[keep[k]]; const std::string &utt = table[s.utts[r]].first; const int32_t T = (int32_t)(fo[k + 1] - fo[k]), nr = (int32_t)(ro[k + 1] - ro[k]);
          const float *src = h.data() + ro[k] * R;
          if (!repeat) writer.WriteMatrix(utt, src, nr, R, R);
          else {
            std::vector<float> rep((size_t)T * R);
            for (int32_t t = 0; t < T; t++) memcpy(&rep[(size_t)t * R], src + (size_t)(t / P) * R, sizeof(float) * R);
            writer.WriteMatrix(utt, rep.data(), T, R, R);
          }
          auto norm = [&](const float *v) { double a = 0; for (int32_t i = 0; i < R; i++) a += (double)v[i] * v[i]; return std::sqrt(a); };
          tot_length_end += T * norm(src + (size_t)(nr - 1) * R); for (int32_t i = 0; i < nr; i++) tot_length += T * norm(src + (size_t)i * R) / nr; tot_t += T;
          // the state the speaker's next utterance starts from: i-vector statistics from the GPU; CMVN statistics of all frames so far
          s.stats.assign(so.begin() + k * SS, so.begin() + (k + 1) * SS); s.has_stats = true;
          const Matrix &f = table[s.utts[r]].second;
          for (int32_t t = 0; t < f.rows; t++) {
            for (int32_t d = 0; d < F; d++) {
              const double x = f.data[(size_t)t * f.cols + d];
              s.cmvn[d] += x;
              s.cmvn[F + 1 + d] += x * x;
            }
            s.cmvn[F] += 1.0;
          }
          num_done++;
        }
        HIPCHK(hipFree(d_f)); HIPCHK(hipFree(d_iv)); HIPCHK(hipFree(d_cm)); HIPCHK(hipFree(d_so)); if (d_si) HIPCHK(hipFree(d_si)); if (d_fw) HIPCHK(hipFree(d_fw));
      }
    }
    writer.Flush();
    K3H_LOG << "Estimated iVectors for " << num_done << " files, " << num_err << " with errors.";
    if (tot_t > 0) K3H_LOG << "Average iVector length per frame was " << (tot_length / tot_t) << " and at utterance-end was " << (tot_length_end / tot_t) <<
        ", over " << tot_t << " frames (weighted by frames).";
    k3_ivector_destroy(iv);
    return num_done != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
