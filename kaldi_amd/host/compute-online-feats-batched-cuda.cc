// compute-online-feats-batched-cuda -- drop-in for cudafeatbin/compute-online-feats-batched-cuda.cc:104-330 on MI355X:
//   compute-online-feats-batched-cuda [options] <wave-rspecifier> <ivector-wspecifier> <feature-wspecifier>
// The audio is fed in chunks of --chunk-length samples, --batch-size chunks (one per active utterance) per GPU call, each utterance on one of
// --num-channels channels that is reused when its audio ends -- the streaming feature driver of k3_online.h (OnlineFeatures::ComputeFeaturesBatched:
// per-channel sample stash for the frame overlap), the class behind cudafeat/online-batched-feature-pipeline-cuda.h:83-92.  Features are
// bit-identical to the whole-utterance programs (compute-fbank-feats-cuda).  Options as OnlineNnet2FeaturePipelineConfig registers them
// (--feature-type, --mfcc-config, --fbank-config, --ivector-extraction-config; online2/online-nnet2-feature-pipeline.h:74-120).
// I-vectors: the reference's batched pipeline re-estimates the i-vector at every chunk and the program writes the LAST estimate per utterance.  Here the
// estimate is made once, after the utterance's last chunk, by the whole-utterance extractor (k3_ivector_extract_batch: the statistics of all
// periods up to the last one that starts inside the utterance, OnlineIvectorFeature's schedule) -- the i-vector ivector-extract-online2 ends on.
// Without --ivector-extraction-config the i-vector table is written with empty vectors like the reference (IvectorDim() == 0).
// Under the name compute-online-feats-cuda (no "batched" in argv[0]) the program is the drop-in for cudafeatbin/compute-online-feats-cuda.cc:30-125:
//   compute-online-feats-cuda [options] <wave-rspecifier> <ivector-wspecifier> <feats-wspecifier>
// OnlineCudaFeaturePipeline::ComputeFeatures (cudafeat/online-cuda-feature-pipeline.h:33-51) on one whole utterance after the other: the same options
// (OnlineNnet2FeaturePipelineConfig), every file one chunk, one file per GPU call, a file that fails is counted and skipped, exit code 1 when nothing succeeded.
#include <hip/hip_runtime.h>
#include <cstring>
#include <deque>
#include <iostream>
#include "k3_feat_options.h"
#include "k3_online.h"
using namespace k3host;
#define HIPCHK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)

int main(int argc, char **argv) {
  try {
    const bool whole = strstr(argv[0], "batched") == nullptr;      // compute-online-feats-cuda: one whole utterance per call
    const char *usage = whole ? "Extract features and ivectors for utterances using the online\nfeature pipeline on the GPU. This class models the online feature pipeline.\n\n"
                                "Usage:  compute-online-feats-cuda [options] <wave-rspecifier> <ivector-wspecifier> <feats-wspecifier>\ne.g.: \n"
                                "  ./compute-online-feats-cuda --config=feature_config wav.scp ark,scp:ivector.ark,ivector.scp ark,scp:feat.ark,feat.scp\n" :
                        "Compute online features and ivector features.\n\nThis binary processes the audio in chunks of samples. In addition, the computation is batched and done on the GPU.\n\n"
                        "Usage: ./compute-online-feats-batched-cuda --batch-size=100 <wave-rspecifier> <ivector-wspecifier> <feature-wspecifier> \n";
    ParseOptions po(usage);
    int32_t num_channels = 50, num_lanes = 10, chunk_len = 10000;
    std::string feature_type = "mfcc", mfcc_config, fbank_config, plp_config, ivector_config, cmvn_config, global_cmvn, pitch_config, use_gpu = "yes";
    bool add_pitch = false;
    po.Register("num-channels", &num_channels, "The number of channels used for compute");
    po.Register("batch-size", &num_lanes, "The number of chunks from audio cuts processed in a single batch");
    po.Register("chunk-length", &chunk_len, "The length of a chunk of audio in terms of samples.");
    po.Register("feature-type", &feature_type, "Base feature type [mfcc, plp, fbank]");
    po.Register("mfcc-config", &mfcc_config, "Configuration file for MFCC features (e.g. conf/mfcc.conf)");
    po.Register("fbank-config", &fbank_config, "Configuration file for filterbank features (e.g. conf/fbank.conf)");
    po.Register("plp-config", &plp_config, "(PLP features are not supported)");
    po.Register("add-pitch", &add_pitch, "(pitch features are not supported)"); po.Register("online-pitch-config", &pitch_config, "(not supported)");
    po.Register("cmvn-config", &cmvn_config, "(online CMVN of the network features is not supported here; see apply-cmvn-online-cuda)");
    po.Register("global-cmvn-stats", &global_cmvn, "(not supported)");
    po.Register("ivector-extraction-config", &ivector_config, "Configuration file for online iVector extraction, see class OnlineIvectorExtractionConfig in the code");
    po.Register("use-gpu", &use_gpu, "(accepted; always the GPU)");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) { po.PrintUsage(); return 1; }
    if (whole) { num_lanes = 1; num_channels = 1; chunk_len = 0x7FFFFFFF; }
    if (add_pitch || !plp_config.empty() || !cmvn_config.empty() || !global_cmvn.empty()) K3H_ERR <<
        "an option that needs a component outside the accelerated path was given (pitch / PLP / CMVN)";
    if (num_channels < num_lanes) K3H_ERR << "--num-channels must be at least --batch-size";
    const bool mfcc = feature_type == "mfcc";
    if (!mfcc && feature_type != "fbank") K3H_ERR << "Invalid feature type: " << feature_type << " (supported: mfcc, fbank)";
    FeatOptions fo(mfcc);
    { ParseOptions fpo(""); fo.Register(&fpo); const std::string &cfg = mfcc ? mfcc_config : fbank_config; if (!cfg.empty()) fpo.ReadConfigFile(cfg); }
    const k3_feat_opts &fopts = fo.Finish();
    k3_feat_plan *plan = nullptr; K3H_CHECK_K3(k3_feat_plan_create(&fopts, &plan));
    const int dim = k3_feat_dim(plan);
    k3_ivector *ivx = nullptr; IvectorExtractionInfo iv_info; int iv_dim = 0;
    if (!ivector_config.empty()) {
      iv_info = ReadIvectorExtractionConfig(ivector_config);
      k3_ivector_model m; memset(&m, 0, sizeof m);
      m.feat_dim = iv_info.global_cmvn_stats.cols - 1;
      m.lda_rows = iv_info.lda_rows;
      m.lda_cols = iv_info.lda_cols;
      m.num_gauss = iv_info.ubm.num_gauss;
      m.ivector_dim = iv_info.ie.ivector_dim;
      m.lda = iv_info.lda.data();
      m.global_cmvn_stats = iv_info.global_cmvn_stats.data.data();
      m.gconsts = iv_info.ubm.gconsts.data();
      m.means_invvars = iv_info.ubm.means_invvars.data();
      m.inv_vars = iv_info.ubm.inv_vars.data();
      m.M = iv_info.ie.M.data(); m.sigma_inv = iv_info.ie.sigma_inv.data(); m.prior_offset = iv_info.ie.prior_offset;
      k3_ivector_opts o; k3_ivector_opts_default(&o);
      o.left_context = iv_info.left_context;
      o.right_context = iv_info.right_context;
      o.num_gselect = iv_info.num_gselect;
      o.min_post = iv_info.min_post;
      o.posterior_scale = iv_info.posterior_scale;
      o.max_count = iv_info.max_count;
      o.ivector_period = iv_info.ivector_period; o.num_cg_iters = iv_info.num_cg_iters; o.online_cmvn_iextractor = iv_info.online_cmvn_iextractor;
      o.cmvn.cmn_window = iv_info.cmn_window;
      o.cmvn.speaker_frames = iv_info.speaker_frames;
      o.cmvn.global_frames = iv_info.global_frames;
      o.cmvn.normalize_mean = iv_info.normalize_mean;
      o.cmvn.normalize_variance = iv_info.normalize_variance;
      if (m.feat_dim != dim) K3H_ERR << "The i-vector extractor expects features of dimension " << m.feat_dim << " but the feature config gives " << dim;
      K3H_CHECK_K3(k3_ivector_create(&m, &o, &ivx)); iv_dim = iv_info.ie.ivector_dim;
    }
    auto scp = ReadScp(po.GetArg(1)); TableWriter ivector_writer(po.GetArg(2)), feature_writer(po.GetArg(3));
    struct Utt { std::string key; Wave wave; size_t pos = 0; std::vector<float> feats; int channel = -1; bool started = false; };
    std::deque<Utt> active; size_t next = 0; std::vector<int> free_channels; for (int c = num_channels - 1; c >= 0; c--) free_channels.push_back(c);
    OnlineFeatures online(plan, fopts, num_channels);
    int num_done = 0, num_fail = 0; int64_t tot_t = 0;
    auto finish = [&](Utt &u) {
      const int nf = (int)(u.feats.size() / dim);
      std::vector<float> ivec(iv_dim, 0.0f);
      if (ivx && nf > 0) {
        float *d_f, *d_iv; std::vector<int64_t> fo{0, nf}; std::vector<int64_t> ro(2); const int64_t rows = k3_ivector_num_rows(ivx, 1, fo.data(), ro.data());
        HIPCHK(hipMalloc((void **)&d_f, u.feats.size() * 4)); HIPCHK(hipMalloc((void **)&d_iv, (size_t)rows * iv_dim * 4));
        HIPCHK(hipMemcpy(d_f, u.feats.data(), u.feats.size() * 4, hipMemcpyHostToDevice));
        K3H_CHECK_K3(k3_ivector_extract_batch(ivx, d_f, dim, fo.data(), 1, d_iv, iv_dim, nullptr));
        HIPCHK(hipMemcpy(ivec.data(), d_iv + (size_t)(rows - 1) * iv_dim, (size_t)iv_dim * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipFree(d_f)); HIPCHK(hipFree(d_iv));
      }
      feature_writer.WriteMatrix(u.key, u.feats.data(), nf, dim, dim); ivector_writer.WriteVector(u.key, ivec.data(), iv_dim);
      num_done++; tot_t += nf; free_channels.push_back(u.channel);
    };
    for (;;) {
      while (next < scp.size() && !free_channels.empty() && (int)active.size() < num_channels) {      // fill the free channels (cudafeatbin/...:262-290)
        Utt u; u.key = scp[next].first;
        if (whole) {      // (:83-113: "Processing Utterance", a failure is a warning and the next file)
          K3H_LOG << "Processing Utterance " << u.key;
          try { u.wave = ReadWave(scp[next].second); if (u.wave.samp_freq != fopts.samp_freq) K3H_ERR << "Sample frequency mismatch"; }
          catch (const std::exception &) { K3H_WARN << "Failed to compute features for utterance " << u.key; num_fail++; next++; continue; }
          next++;
        } else {
        u.wave = ReadWave(scp[next].second); next++;
        if (u.wave.samp_freq != fopts.samp_freq) K3H_ERR << "Sample frequency mismatch for " << u.key << ": " << u.wave.samp_freq << " vs " << fopts.samp_freq;
        }
        u.channel = free_channels.back(); free_channels.pop_back(); active.push_back(std::move(u));
      }
      if (active.empty()) break;
      const int n = std::min<int>(num_lanes, (int)active.size());      // one batch: the next chunk of the first n active utterances
      std::vector<int> channels(n); std::vector<std::vector<float>> chunks(n); std::vector<char> first(n);
      for (int i = 0; i < n; i++) {
        Utt &u = active[i]; const size_t len = std::min<size_t>((size_t)chunk_len, u.wave.samples.size() - u.pos);
        channels[i] = u.channel; first[i] = !u.started; u.started = true; chunks[i].assign(u.wave.samples.begin() + u.pos, u.wave.samples.begin() + u.pos + len); u.pos += len;
      }
      float *d_feats = nullptr; const std::vector<int> nf = online.ComputeFeaturesBatched(channels, chunks, first, &d_feats);
      int64_t tot = 0; for (int v : nf) tot += v;
      std::vector<float> h((size_t)tot * dim); if (tot > 0) HIPCHK(hipMemcpy(h.data(), d_feats, h.size() * 4, hipMemcpyDeviceToHost));
      int64_t off = 0;
      for (int i = 0; i < n; i++) { active[i].feats.insert(active[i].feats.end(), h.begin() + off * dim, h.begin() + (off + nf[i]) * dim); off += nf[i]; }
      std::deque<Utt> keep;      // finished utterances leave, the others rotate to the back (round robin over the channels)
      for (int i = 0; i < (int)active.size(); i++) { Utt &u = active[i]; if (i < n && u.pos >= u.wave.samples.size()) finish(u); else keep.push_back(std::move(u)); }
      for (int i = 0; i < n && !keep.empty() && (int)keep.size() > num_lanes; i++) { keep.push_back(std::move(keep.front())); keep.pop_front(); }
      active.swap(keep);
    }
    feature_writer.Flush(); ivector_writer.Flush(); k3_feat_plan_destroy(plan); if (ivx) k3_ivector_destroy(ivx);
    if (whole) { K3H_LOG << "Processed " << num_done + num_fail << " utterances with " << num_fail << " failures."; return num_done != 0 ? 0 : 1; }
    K3H_LOG << "Computed online features for " << num_done << " files, and " << tot_t << " total feature frames.";
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
