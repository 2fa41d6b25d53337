// compute-fbank-online-batched-cuda / compute-mfcc-online-batched-cuda -- drop-ins for cudafeatbin/compute-fbank-online-batched-cuda.cc:64-388 and
// cudafeatbin/compute-mfcc-online-batched-cuda.cc:64-388 on MI355X:
//   compute-fbank-online-batched-cuda [options] <wave-rspecifier> <feature-wspecifier>
// The reference's test driver of CudaOnlineBatchedSpectralFeatures (cudafeat/feature-online-batched-spectral-cuda.h:73-140): every utterance is fed in chunks
// of --chunk-length SAMPLES, --batch-size chunks (one per active utterance) per GPU call, each utterance on one of --num-channels channels that is reused
// when its audio ends.  Options: FbankOptions / MfccOptions registered directly (feat/feature-fbank.h:62-79, feature-mfcc.h:62-79), --num-channels,
// --batch-size, --chunk-length.  The per-channel state is the sample stash of the frame overlap (k3_online.h: OnlineFeatures::ComputeFeaturesBatched), so a
// chunked stream gives the rows of the whole-utterance computation bit for bit (compute-fbank-feats-cuda); like the reference, all waves are read first
// and the features are written after the last batch, and the closing log line reports the compute time only.
// The program computes MFCCs when invoked under a name containing "mfcc".
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstring>
#include <deque>
#include <iostream>
#include "k3_feat_options.h"
#include "k3_online.h"
using namespace k3host;
#define HIPCHK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)

int main(int argc, char **argv) {
  try {
    const bool mfcc = strstr(argv[0], "mfcc") != nullptr;
    const std::string usage = std::string("Compute online ") + (mfcc ? "mfcc" : "fbank") +
        " features.\n\nThis binary processes the audio in chunks of samples. In addition, the computation is batched and done on the GPU. "
                              "This binary is not intended to demonstrate how to achieve maximum performance.  Instead it is intended to demonstrate how to use the batched online feature class and provide "
                              "a mechanism to test this class independently.\n\nUsage: ./" +
                                  (mfcc ? "compute-mfcc-online-batched-cuda" : "compute-fbank-online-batched-cuda") +
                                  " --batch-size=50 <wave-rspecifier> <feature-wspecifier> \n";
    ParseOptions po(usage.c_str());
    FeatOptions fo(mfcc); fo.Register(&po);
    int32_t num_channels = 50, num_lanes = 10, chunk_len = 10000; std::string use_gpu = "yes";
    po.Register("num-channels", &num_channels, "The number of channels used for compute");
    po.Register("batch-size", &num_lanes, "The number of chunks from audio cuts processed in a single batch");
    po.Register("chunk-length", &chunk_len, "The length of a chunk of audio in terms of samples."); po.Register("use-gpu", &use_gpu, "(accepted; always the GPU)");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); return 1; }
    if (num_channels < num_lanes) K3H_ERR << "--num-channels must be at least --batch-size";
    if (chunk_len < 1) K3H_ERR << "--chunk-length must be positive";
    const k3_feat_opts &fopts = fo.Finish();
    k3_feat_plan *plan = nullptr; K3H_CHECK_K3(k3_feat_plan_create(&fopts, &plan));
    const int dim = k3_feat_dim(plan);
    // preload data for batching (cudafeatbin/compute-fbank-online-batched-cuda.cc:188-201): a file at another sampling rate is an error, as there
    struct Utt { std::string key; Wave wave; size_t pos = 0; std::vector<float> feats; int channel = -1; bool started = false; };
    std::vector<Utt> utts; double duration = 0.0;
    for (auto &e : ReadScp(po.GetArg(1))) {
      Utt u; u.key = e.first; u.wave = ReadWave(e.second);
      if (u.wave.samp_freq != fopts.samp_freq) K3H_ERR << "File: " << u.key << " has an mismatched sampling rate (config= " << fopts.samp_freq << " vs file="
          << u.wave.samp_freq << ".";
      duration += u.wave.samples.size() / (double)u.wave.samp_freq; utts.push_back(std::move(u));
    }
    TableWriter feature_writer(po.GetArg(2));
    OnlineFeatures online(plan, fopts, num_channels);
    std::vector<int> free_channels; for (int c = num_channels - 1; c >= 0; c--) free_channels.push_back(c);
    std::deque<size_t> lanes; size_t not_done = 0; int num_done = 0; int64_t tot_t = 0;
    const auto t0 = std::chrono::steady_clock::now();      // "Timing just compute, we don't want to include disc I/O in this timer."
    for (;;) {
      // fill the batch (:212-236)
      while ((int)lanes.size() < num_lanes && not_done < utts.size()) {
        utts[not_done].channel = free_channels.back();
        free_channels.pop_back();
        lanes.push_back(not_done++);
      }
      if (lanes.empty()) break;
      const int n = (int)lanes.size();
      std::vector<int> channels(n); std::vector<std::vector<float>> chunks(n); std::vector<char> first(n);
      for (int i = 0; i < n; i++) {
        Utt &u = utts[lanes[i]]; const size_t len = std::min<size_t>((size_t)chunk_len, u.wave.samples.size() - u.pos);
        channels[i] = u.channel; first[i] = !u.started; u.started = true; chunks[i].assign(u.wave.samples.begin() + u.pos, u.wave.samples.begin() + u.pos + len); u.pos += len;
      }
      float *d_feats = nullptr; const std::vector<int> nf = online.ComputeFeaturesBatched(channels, chunks, first, &d_feats);
      int64_t tot = 0; for (int v : nf) tot += v;
      std::vector<float> h((size_t)tot * dim); if (tot > 0) HIPCHK(hipMemcpy(h.data(), d_feats, h.size() * 4, hipMemcpyDeviceToHost));
      int64_t off = 0; std::deque<size_t> keep;
      for (int i = 0; i < n; i++) {
        Utt &u = utts[lanes[i]]; u.feats.insert(u.feats.end(), h.begin() + off * dim, h.begin() + (off + nf[i]) * dim); off += nf[i];
        if (u.pos >= u.wave.samples.size()) { free_channels.push_back(u.channel); num_done++; } else keep.push_back(lanes[i]);      // a finished lane frees its channel (:340-352)
      }
      lanes.swap(keep);
    }
    const double total_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // "output all utterances" (:361-366)
    for (auto &u : utts) {
      const int nf = (int)(u.feats.size() / dim);
      tot_t += nf;
      feature_writer.WriteMatrix(u.key, u.feats.data(), nf, dim, dim);
    }
    feature_writer.Flush(); k3_feat_plan_destroy(plan);
    K3H_LOG << "Computed Online Features for  " << num_done << " files, and " << tot_t << " frames.";
    K3H_LOG << "Total Audio: " << duration << " seconds, Total Time: " << total_time << " seconds, RTFX: " << duration / total_time;
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
