// apply-batched-cmvn-online-cuda -- drop-in for cudafeatbin/apply-batched-cmvn-online-cuda.cc:49-300 on MI355X:
//   apply-batched-cmvn-online-cuda [options] <global-cmvn-stats> <feature-rspecifier> <feature-wspecifier>
// The reference's test driver of CudaOnlineBatchedCmvn (cudafeat/feature-online-batched-cmvn-cuda.h:40-110): every utterance's features go through online
// CMVN in chunks of --chunk-length FRAMES, --batch-size chunks (one per active utterance) per GPU call, each utterance on one of --num-channels channels
// that is reused when it ends; no speaker statistics (spk2utt is not supported, as there).  Options: OnlineCmvnOptions::Register
// (feat/online-feature.h:231-247) + --num-channels, --batch-size, --chunk-length, --stats-coarsening-factor.
// A channel's state between two calls is what OnlineCmvn keeps between two GetFrame calls (feat/online-feature.cc:361-468): the window's float64 (sum, sum of
// squares, count) per column -- k3_cmvn_online_batch_resume's carry -- and the last min(cmn_window, frames so far) raw rows the window still subtracts; the
// rows written are those of apply-cmvn-online on the whole utterance, bit for bit, for any chunking.  --stats-coarsening-factor is accepted and has no effect:
// the reference's GPU kernel keeps prefix sums of the statistics every `factor` frames to save memory (an approximation when > 1); the recursion here is exact.
#include <hip/hip_runtime.h>
#include <iostream>
#include <deque>
#include "k3_online.h"
using namespace k3host;
#define HIPCHK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)

int main(int argc, char **argv) {
  try {
    const char *usage =
        "Apply online cepstral mean (and possibly variance) computation online,\nusing the same code as used for online decoding in the 'new' setup in\nonline2/ and online2bin/.'\n"
        "The computation is done on the device in chunks that are batched. spk2utt is not supported.\n\n"
        "Usage: apply-batched-cmvn-online-cuda [options] <global-cmvn-stats> <feature-rspecifier> <feature-wspecifier>\n"
        "e.g. apply-batched-cmvn-online-cuda 'matrix-sum scp:data/train/cmvn.scp -|' data/train/split8/1/feats.scp ark:-\n";
    ParseOptions po(usage);
    int32_t num_channels = 200, batch_size = 100, chunk_length = 10000, coarsening = 1;
    po.Register("num-channels", &num_channels, "The number of channels used for compute");
    po.Register("batch-size", &batch_size, "The number of chunks from audio cuts processed in a single batch");
    po.Register("chunk-length", &chunk_length, "The length of a chunk of audio in frames that is processed at one time");
    po.Register("stats-coarsening-factor", &coarsening, " Coarsen CMVN stats by this factor.  (accepted; the statistics here are exact, i.e. the behaviour of factor 1)");
    k3_online_cmvn_opts o; k3_online_cmvn_opts_default(&o);
    bool norm_vars = false, norm_means = true; std::string skip_dims_str, use_gpu = "yes";
    po.Register("cmn-window", &o.cmn_window, "Number of frames of sliding context for cepstral mean normalization.");
    po.Register("global-frames", &o.global_frames, "Number of frames of global-average cepstral mean normalization stats to use for first utterance of a speaker");
    po.Register("speaker-frames", &o.speaker_frames, "Number of frames of previous utterance(s) from this speaker to use in cepstral mean normalization");
    po.Register("norm-vars", &norm_vars, "If true, do cepstral variance normalization in addition to cepstral mean normalization ");
    po.Register("norm-means", &norm_means, "If true, do mean normalization (note: you cannot normalize the variance but not the mean)");
    po.Register("skip-dims", &skip_dims_str, "Dimensions to skip normalization of (colon-separated list of integers)");
    po.Register("use-gpu", &use_gpu, "(accepted; always the GPU)");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) { po.PrintUsage(); return 1; }
    if (num_channels < batch_size) K3H_ERR << "--num-channels must be at least --batch-size";
    if (chunk_length < 1 || batch_size < 1) K3H_ERR << "--chunk-length and --batch-size must be positive";
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
    auto table = ReadMatrixTable(po.GetArg(2)); TableWriter writer(po.GetArg(3));      // "preload data for batching" (:147-158)
    struct Utt { size_t idx; int64_t cur = 0; std::vector<float> out; std::vector<double> carry; };
    std::vector<Utt> utts(table.size());
    for (size_t i = 0; i < table.size(); i++) {
      if (table[i].second.cols != dim) K3H_ERR << "Dim mismatch: cmvn stats " << dim << " vs features " << table[i].second.cols << " for " << table[i].first;
      utts[i].idx = i; utts[i].out.resize(table[i].second.data.size()); utts[i].carry.assign((size_t)dim * 3, 0.0);
    }
    DevBuf<double> d_g, d_carry; DevBuf<float> d_in, d_out; DevBuf<int64_t> d_fo, d_tb;
    d_g.upload(gstats.data);
    std::deque<size_t> lanes; size_t not_done = 0; int32_t free_channels = num_channels;
    for (;;) {
      // fill the batch with new work (:177-197); an empty utterance is done at once
      while ((int)lanes.size() < batch_size && not_done < utts.size() && free_channels > 0) {
        if (table[not_done].second.rows > 0) {
          lanes.push_back(not_done);
          free_channels--;
        }
        not_done++;
      }
      if (lanes.empty()) break;
      const int n = (int)lanes.size(); const int64_t W = o.cmn_window;
      // a lane's rows of this call: the history the window still reads, then the chunk
      std::vector<float> in; std::vector<int64_t> fo(1, 0), tb(n); std::vector<double> carry((size_t)n * dim * 3); std::vector<int64_t> nchunk(n);
      for (int i = 0; i < n; i++) {
        Utt &u = utts[lanes[i]]; const Matrix &m = table[u.idx].second;
        const int64_t h0 = std::max<int64_t>(0, u.cur - W), c1 = std::min<int64_t>(m.rows, u.cur + chunk_length);
        in.insert(in.end(), m.data.begin() + h0 * dim, m.data.begin() + c1 * dim); fo.push_back(fo.back() + (c1 - h0)); tb[i] = u.cur - h0; nchunk[i] = c1 - u.cur;
        std::copy(u.carry.begin(), u.carry.end(), carry.begin() + (size_t)i * dim * 3);
      }
      d_in.upload(in); d_fo.upload(fo); d_tb.upload(tb); d_carry.upload(carry); d_out.need(in.size());
      K3H_CHECK_K3(k3_cmvn_online_batch_resume(d_in.p, dim, d_out.p, dim, dim, d_fo.p, n, &o, d_g.p, nullptr, skip.data(), (int32_t)skip.size(), d_tb.p, d_carry.p, nullptr));
      std::vector<float> out(in.size());
      HIPCHK(hipMemcpy(out.data(), d_out.p, out.size() * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(carry.data(), d_carry.p, carry.size() * 8, hipMemcpyDeviceToHost));
      std::deque<size_t> keep;
      for (int i = 0; i < n; i++) {
        Utt &u = utts[lanes[i]]; const Matrix &m = table[u.idx].second;
        std::copy(out.begin() + (fo[i] + tb[i]) * dim, out.begin() + fo[i + 1] * dim, u.out.begin() + u.cur * dim);
        std::copy(carry.begin() + (size_t)i * dim * 3, carry.begin() + (size_t)(i + 1) * dim * 3, u.carry.begin());
        u.cur += nchunk[i];
        if (u.cur >= m.rows) free_channels++; else keep.push_back(lanes[i]);      // a finished lane frees its channel (:262-279)
      }
      lanes.swap(keep);
    }
    int32_t num_done = 0; int64_t tot_t = 0;
    // "output all utterances" (:281-292)
    for (auto &u : utts) {
      const Matrix &m = table[u.idx].second;
      writer.WriteMatrix(table[u.idx].first, u.out.data(), m.rows, dim, dim);
      num_done++;
      tot_t += m.rows;
    }
    writer.Flush();
    K3H_LOG << "Applied online CMVN to " << num_done << " files, or " << tot_t << " frames.";
    return num_done != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
