// compute-fbank-feats-cuda / compute-mfcc-feats-cuda -- drop-in for the reference's cudafeatbin programs
// (cudafeatbin/compute-fbank-feats-cuda.cc, compute-mfcc-feats-cuda.cc; CPU twins featbin/compute-fbank-feats.cc:79-190):
//   compute-fbank-feats-cuda [options] <wav-rspecifier> <feats-wspecifier>
// All utterances of the script file go through ONE k3_feat_compute_batch call per --max-batch-size utterances.
// The program computes MFCCs when invoked under a name containing "mfcc".
#include <hip/hip_runtime.h>
#include <cstring>
#include <iostream>
#include "k3_feat_options.h"
using namespace k3host;
#define HIPCHK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)

int main(int argc, char **argv) {
  try {
    const bool mfcc = strstr(argv[0], "mfcc") != nullptr;
    const char *usage = "Create fbank / MFCC feature files (GPU).\nUsage:  compute-fbank-feats-cuda [options...] <wav-rspecifier> <feats-wspecifier>\n";
    ParseOptions po(usage);
    FeatOptions fo(mfcc); fo.Register(&po);
    int32_t max_batch = 512, channel = -1; float min_duration = 0.0f; std::string use_gpu = "yes";
    po.Register("max-batch-size", &max_batch, "Utterances per GPU batch"); po.Register("channel", &channel, "Channel to extract (-1 -> expect mono, 0 -> left)");
    po.Register("min-duration", &min_duration, "Minimum duration of segments to process (in seconds).");
    po.Register("use-gpu", &use_gpu, "(accepted; this program always uses the GPU)");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); return 1; }
    const k3_feat_opts &opts = fo.Finish();
    k3_feat_plan *plan = nullptr; K3H_CHECK_K3(k3_feat_plan_create(&opts, &plan));
    const int dim = k3_feat_dim(plan);
    auto scp = ReadScp(po.GetArg(1)); TableWriter writer(po.GetArg(2));
    int num_done = 0, num_err = 0; (void)num_err;
    for (size_t b0 = 0; b0 < scp.size(); b0 += max_batch) {
      const size_t b1 = std::min(scp.size(), b0 + (size_t)max_batch);
      std::vector<std::string> keys; std::vector<float> all; std::vector<int64_t> woff(1, 0), foff(1, 0);
      for (size_t i = b0; i < b1; i++) {
        Wave w;
        // --channel like featbin/compute-fbank-feats.cc:140-160: -1 = expect mono (a multi-channel file: warn, take the left channel), else that channel or
        // skip the file
        try {
          int nch = 0; w = ReadWave(scp[i].second, 0, &nch);
          if (channel == -1) { if (nch != 1) K3H_WARN << "Channel not specified but you have data with " << nch << " channels; defaulting to zero"; }
          else if (channel >= nch) {
            K3H_WARN << "File with id " << scp[i].first << " has " << nch << " channels but you specified channel " << channel << ", producing no output.";
            num_err++;
            continue;
          }
          else if (channel > 0) w = ReadWave(scp[i].second, channel);
        } catch (const FatalError &) { num_err++; continue; }
        if (w.samples.size() / w.samp_freq < min_duration) { K3H_WARN << "File: " << scp[i].first << " is too short: producing no output."; num_err++; continue; }
        if (!MatchSampleRate(&w, opts.samp_freq, fo.allow_downsample, fo.allow_upsample)) {
          K3H_WARN << "Waveform and config sample Frequency mismatch: " << w.samp_freq << " .vs " << opts.samp_freq <<
              " (use --allow-downsample=true / --allow-upsample=true to resample); failed to compute features for utterance " << scp[i].first;
          num_err++;
          continue;
        }
        const int nf = k3_feat_num_frames(plan, (int64_t)w.samples.size());
        if (nf == 0) { K3H_WARN << "No frames fit in file " << scp[i].first << " (#samp = " << w.samples.size() << ")"; num_err++; continue; }
        keys.push_back(scp[i].first); all.insert(all.end(), w.samples.begin(), w.samples.end());
        woff.push_back((int64_t)all.size()); foff.push_back(foff.back() + nf);
      }
      if (keys.empty()) continue;
      float *d_w, *d_f; int64_t *d_wo, *d_fo; const int64_t tot = foff.back();
      HIPCHK(hipMalloc((void **)&d_w, all.size() * 4)); HIPCHK(hipMalloc((void **)&d_f, (size_t)tot * dim * 4));
      HIPCHK(hipMalloc((void **)&d_wo, woff.size() * 8)); HIPCHK(hipMalloc((void **)&d_fo, foff.size() * 8));
      HIPCHK(hipMemcpy(d_w, all.data(), all.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d_wo, woff.data(), woff.size() * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_fo, foff.data(), foff.size() * 8, hipMemcpyHostToDevice));
      K3H_CHECK_K3(k3_feat_compute_batch(plan, d_w, d_wo, d_fo, (int32_t)keys.size(), tot, d_f, dim, nullptr));
      std::vector<float> h((size_t)tot * dim); HIPCHK(hipMemcpy(h.data(), d_f, h.size() * 4, hipMemcpyDeviceToHost));
      for (size_t u = 0; u < keys.size(); u++) { writer.WriteMatrix(keys[u], h.data() + foff[u] * dim, (int32_t)(foff[u + 1] - foff[u]), dim, dim); num_done++; }
      HIPCHK(hipFree(d_w)); HIPCHK(hipFree(d_f)); HIPCHK(hipFree(d_wo)); HIPCHK(hipFree(d_fo));
    }
    writer.Flush(); k3_feat_plan_destroy(plan);
    K3H_LOG << " Done " << num_done << " out of " << scp.size() << " utterances.";
    return num_done != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
