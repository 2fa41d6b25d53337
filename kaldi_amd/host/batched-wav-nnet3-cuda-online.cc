// batched-wav-nnet3-cuda-online -- the streaming counterpart of batched-wav-nnet3-cuda2 (cudadecoderbin/batched-wav-nnet3-cuda-online.cc):
//   batched-wav-nnet3-cuda-online [options] <nnet3-in> <fst-in> <wav-rspecifier> <lattice-wspecifier>
// The wav files are played as --num-channels concurrent audio streams: every round each busy channel submits its next chunk of
// --frames-per-chunk frames' worth of samples; the chunks go through the streaming drivers of k3_online.h (sample stash ->
// k3_feat_compute_batch, input-context stash -> k3_nnet_forward planned once, k3_decoder_advance_decoding); a channel whose stream
// ended is finalised (k3_decoder_finalize_channels), its lattice determinized on the host (default; see batched-wav-nnet3-cuda2.cc) and
// written, and the channel given to the next file.  Lattices are bit-identical to batched-wav-nnet3-cuda2's.
// The reference program's outputs (cudadecoderbin/cuda-bin-tools.h:33-104, batched-wav-nnet3-cuda-online.cc:150-320): --print-hypotheses / --print-partial-hypotheses /
// --print-endpoints lines "corr_id #N ...", --write-lattice (default false, like the reference) / --generate-lattice, --lattice-postprocessor-rxfilename with CTM output when
// the fourth argument is not a table wspecifier, and the latency statistics of PrintLatencyStats.  One deliberate difference: the reference ALWAYS plays its streams in real
// time (a stream's chunk is submitted when it would have been spoken); here that is --simulate-realtime-writing=true and the default submits chunks as fast as the GPU takes
// them (a throughput run; the latency of an utterance is then counted from the submission of its last chunk).
#include <chrono>
#include <cstring>
#include <cmath>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <iostream>
#include <mutex>
#include <random>
#include <set>
#include <sstream>
#include <fstream>
#include <thread>
#include "k3_feat_options.h"
#include "k3_online.h"
using namespace k3host;

// The simulation's audio source: the wav files are read and parsed by a few background threads, a bounded number of files ahead of the channel that will play them, in the order
// the main loop admits them (iteration-major).  Reading 512 files of 10 s between two iterations was ~60 ms on the main thread with the GPU idle -- a fifth of an iteration.
namespace {
struct WavePrefetcher {
  struct Item { k3host::Wave wave; bool ok = false, ready = false; };
  WavePrefetcher(const std::vector<std::pair<std::string, std::string>> &scp, int iterations, size_t window, int threads) : scp_(scp),
      total_((size_t)iterations * scp.size()), window_(std::max<size_t>(window, 1)), items_(window_) {
    for (int t = 0; t < std::max(1, threads); t++) th_.emplace_back([this] { Loop(); });
  }
  ~WavePrefetcher() { { std::lock_guard<std::mutex> l(m_); stop_ = true; } cv_.notify_all(); for (auto &t : th_) t.join(); }
  // the k-th file of the run (k = iteration * files + index): false when it could not be read
  bool Get(size_t k, k3host::Wave *out) {
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return items_[k % window_].ready && tag_[k % window_] == k; });
    Item &it = items_[k % window_]; const bool ok = it.ok; if (ok) *out = std::move(it.wave);
    it = Item(); consumed_ = k + 1; l.unlock(); cv_.notify_all();
    return ok;
  }
 private:
  void Loop() {
    for (;;) {
      size_t k;
      { std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || (next_ < total_ && next_ < consumed_ + window_); });
        if (stop_ || next_ >= total_) return;
        k = next_++; }
      Item it;
      try { it.wave = k3host::ReadWave(scp_[k % scp_.size()].second); it.ok = true; } catch (const k3host::FatalError &) { it.ok = false; }
      it.ready = true;
      { std::lock_guard<std::mutex> l(m_); items_[k % window_] = std::move(it); tag_[k % window_] = k; }
      cv_.notify_all();
    }
  }
  const std::vector<std::pair<std::string, std::string>> &scp_; size_t total_, window_; std::vector<Item> items_; std::map<size_t, size_t> tag_;
  std::mutex m_; std::condition_variable cv_; size_t next_ = 0, consumed_ = 0; bool stop_ = false; std::vector<std::thread> th_;
};
}  // namespace

int main(int argc, char **argv) {
  try {
    const char *usage =
        "Reads in wav file(s) and simulates online decoding with neural nets (nnet3 setup), the audio of several files being fed chunk by chunk.\n"
        "Usage: batched-wav-nnet3-cuda-online [options] <nnet3-in> <fst-in> <wav-rspecifier> <lattice-wspecifier>\n";
    ParseOptions po(usage);
    bool literal_order = true; float hash_ratio = 2.0f;
    bool write_compact = true, write_lattice = false, generate_lattice = false, determinize = true, minimize = false, phone_det = true, word_det = true, print_partial = false,
        print_hyp = false, print_endpoints = false, simulate_rt = false, reset_on_endpoint = false;
    // OnlineEndpointConfig (online2/online-endpoint.h:122-170): {must_contain_nonsilence, min_trailing_silence, max_relative_cost, min_utterance_length} x 5
    struct Rule { bool nonsil; float sil, rel, len; };
    const float kInfF = std::numeric_limits<float>::infinity();
    Rule rules[5] = {{false, 5.0f, kInfF, 0.0f}, {true, 0.5f, 2.0f, 0.0f}, {true, 1.0f, 8.0f, 0.0f}, {true, 2.0f, kInfF, 0.0f}, {false, 0.0f, kInfF, 20.0f}};
    std::string silence_phones, postproc;
    int32_t worker_threads = -1;
    int32_t num_todo = -1, iterations = 1, max_batch = 400, num_channels = -1, frames_per_chunk = 51, subsampling = 1, num_streaming = 2000;
    int32_t max_active = 10000, min_active = 200, main_q = -1, aux_q = -1, ntok_pre = 1000000, max_frames = 6000;
    float beam = 15.0f, lattice_beam = 10.0f, acoustic_scale = 0.1f, beam_delta = 0.5f, det_delta = 1.0f / 1024.0f; int32_t det_max_mem = 50000000;
    std::string feature_type = "mfcc", mfcc_config, fbank_config, word_syms, use_gpu = "yes", ivector_config;
    po.Register("print-hypotheses", &print_hyp, "Prints the final hypotheses");
    po.Register("print-partial-hypotheses", &print_partial, "Prints the partial hypotheses");
    po.Register("print-endpoints", &print_endpoints, "Prints the detected endpoints");
    po.Register("generate-lattice", &generate_lattice, "Generate full lattices");
    po.Register("write-lattice", &write_lattice, "Output lattice to a file");
    po.Register("lattice-postprocessor-rxfilename", &postproc, "(optional) Config file for lattice postprocessor");
    po.Register("word-symbol-table", &word_syms, "Symbol table for words [for debug output]");
    po.Register("endpoint.silence-phones", &silence_phones, "List of phones that are considered to be silence phones by the endpointing code.");
    for (int r = 0; r < 5; r++) {
      const std::string pre = "endpoint.rule" + std::to_string(r + 1) + ".";
      po.Register(pre + "must-contain-nonsilence", &rules[r].nonsil, "If true, for this endpointing rule to apply there must be nonsilence in the best-path traceback.");
      po.Register(pre + "min-trailing-silence", &rules[r].sil, "This endpointing rule requires duration of trailing silence (in seconds) to be >= this value.");
      po.Register(pre + "max-relative-cost", &rules[r].rel, "This endpointing rule requires relative-cost of final-states to be <= this value (describes how good the probability of final-states is).");
      po.Register(pre + "min-utterance-length", &rules[r].len, "This endpointing rule requires utterance-length (in seconds) to be >= this value.");
    }
    po.Register("file-limit", &num_todo, "Limits the number of files that are processed by this driver.");
    po.Register("iterations", &iterations, "Number of times to decode the corpus. Output will be written only once.");
    po.Register("max-batch-size", &max_batch, "The maximum execution batch size (chunks evaluated together)");
    po.Register("num-channels", &num_channels, "The number of parallel audio channels (-1 = max-batch-size)");
    po.Register("num-parallel-streaming-channels", &num_streaming, "(accepted; the streams are fed round-robin over --num-channels)");
    po.Register("determinize-lattice", &determinize, "Determinize the lattice before output.");
    po.Register("write-compact", &write_compact,
        "(not in the reference) with --determinize-lattice=false: true = the state-level lattice re-packed as a CompactLattice like the reference (ConvertLattice), false = written as a Lattice table");
    po.Register("cuda-worker-threads", &worker_threads,
        "The total number of CPU threads launched to process CPU tasks (here: lattice determinization). -1 = use std::hardware_concurrency().");
    po.Register("delta", &det_delta, "Tolerance used in determinization");
    po.Register("max-mem", &det_max_mem, "Maximum approximate memory usage in determinization (real usage might be many times this).");
    po.Register("phone-determinize", &phone_det, "If true, do an initial pass of determinization on both phones and words (see also --word-determinize)");
    po.Register("word-determinize", &word_det, "If true, do a second pass of determinization on words only (see also --phone-determinize)");
    po.Register("minimize", &minimize, "If true, push and minimize after determinization.");
    po.Register("simulate-realtime-writing", &simulate_rt,
        "(not in the reference, whose streams are always paced) true = a stream's chunk is submitted when it would have been spoken: streams start at a random point of the first "
        "second, latency = result time - end of speech; false = chunks are submitted as fast as the GPU takes them");
    po.Register("reset-on-endpoint", &reset_on_endpoint, "Reset a decoder channel when endpoint detected. Do not close stream (accepted; streams are decoded as one segment)");
    po.Register("literal-order", &literal_order,
        "(not in the reference) true = raw lattices identical to the CPU LatticeFasterDecoder's; false = the order-independent fast decoder");
    po.Register("hash-ratio", &hash_ratio, "LatticeFasterDecoderConfig::hash_ratio (used with --literal-order)");
    po.Register("beam", &beam, "Decoding beam. Larger->slower, more accurate."); po.Register("lattice-beam", &lattice_beam, "The width of the lattice beam");
    po.Register("max-active", &max_active, "Decoder max active states. Larger->slower; more accurate"); po.Register("min-active", &min_active, "Decoder min active states");
    po.Register("beam-delta", &beam_delta, "Increment used when the active-state limits move the beam");
    po.Register("main-q-capacity", &main_q, "Max tokens alive on one frame of one utterance (-1 = 4 * max-active, capped)");
    po.Register("aux-q-capacity", &aux_q, "Max arcs considered on one frame (-1 = 3 * main-q-capacity)");
    po.Register("ntokens-pre-allocated", &ntok_pre,
        "Advanced - Number of tokens pre-allocated in host buffers to store lattices. If this size is exceeded the buffer will reallocate (here: an utterance that outgrows it moves to bigger token / link pools inside the decoder kernel)");
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic log-likelihoods");
    po.Register("frame-subsampling-factor", &subsampling,
        "Required if the frame-rate of the output (e.g. in 'chain' models) is less than the frame-rate of the original alignment.");
    po.Register("ivector-extraction-config", &ivector_config,
        "Configuration file for online iVector extraction, see class OnlineIvectorExtractionConfig in the code.  Every chunk the network evaluates gets the extractor's "
                "latest i-vector for its stream (the estimate at the last multiple of --ivector-period among the frames seen so far), as in nnet3/decodable-online-looped.cc");
    po.Register("frames-per-chunk", &frames_per_chunk, "Number of feature frames evaluated per chunk and channel (a multiple of --frame-subsampling-factor)");
    po.Register("max-utterance-frames", &max_frames, "Upper bound on the decoded (subsampled) frames of one utterance: sizes the per-channel frame tables");
    po.Register("feature-type", &feature_type, "Base feature type [mfcc, fbank]");
    po.Register("mfcc-config", &mfcc_config, "Configuration file for MFCC features (e.g. conf/mfcc.conf)");
    po.Register("fbank-config", &fbank_config, "Configuration file for filterbank features (e.g. conf/fbank.conf)"); po.Register("use-gpu", &use_gpu, "(accepted; always the GPU)");
    po.Read(argc, argv);
    if (po.NumArgs() != 4) { po.PrintUsage(); return 1; }
    if (write_lattice) generate_lattice = true;
    DeterminizeLatticePhonePrunedOptions det_opts;
    det_opts.delta = det_delta;
    det_opts.max_mem = det_max_mem;
    det_opts.phone_determinize = phone_det;
    det_opts.word_determinize = word_det;
    det_opts.minimize = minimize;
    if (num_channels < 0) num_channels = max_batch;
    if (num_channels > max_batch) max_batch = num_channels;      // one slot per channel and round
    const std::string nnet3_rx = po.GetArg(1), fst_rx = po.GetArg(2), wav_rspec = po.GetArg(3), out_wspec = po.GetArg(4);

    const bool mfcc = feature_type == "mfcc";
    if (!mfcc && feature_type != "fbank") K3H_ERR << "Invalid feature type: " << feature_type << " (supported: mfcc, fbank)";
    FeatOptions fo(mfcc);
    { ParseOptions fpo(""); fo.Register(&fpo); const std::string &cfg = mfcc ? mfcc_config : fbank_config; if (!cfg.empty()) fpo.ReadConfigFile(cfg); }
    const k3_feat_opts &fopts = fo.Finish();
    k3_feat_plan *plan = nullptr; K3H_CHECK_K3(k3_feat_plan_create(&fopts, &plan));
    const int fdim = k3_feat_dim(plan);
    TransitionInfo ti = ReadTransitionModel(nnet3_rx);
    k3_nnet *nnet = nullptr; K3H_CHECK_K3(k3_nnet_load(nnet3_rx.c_str(), &nnet));
    k3_nnet_info ninfo; K3H_CHECK_K3(k3_nnet_get_info(nnet, &ninfo));
    if (ninfo.input_dim != fdim) K3H_ERR << "Feature dimension " << fdim << " does not match the model's input dimension " << ninfo.input_dim;
    k3_ivector *ivx = nullptr; IvectorExtractionInfo iv_info;
    if (!ivector_config.empty()) { iv_info = ReadIvectorExtractionConfig(ivector_config); ivx = CreateIvectorExtractor(iv_info, fdim); }
    if ((ninfo.ivector_dim > 0) != (ivx != nullptr) || (ivx && ninfo.ivector_dim != iv_info.ie.ivector_dim))
      K3H_ERR << "Neural net expects 'ivector' features with dimension " << ninfo.ivector_dim << " but you provided " << (ivx ? iv_info.ie.ivector_dim : 0);
    if (ninfo.output_dim != ti.num_pdfs) K3H_ERR << "Model output dimension " << ninfo.output_dim << " != number of pdfs in the transition model " << ti.num_pdfs;
    std::vector<float> log_priors;
    if (ninfo.has_priors) { log_priors.resize(ninfo.output_dim); K3H_CHECK_K3(k3_nnet_get_priors(nnet, log_priors.data())); for (float &p : log_priors) p = logf(p); }
    HostFst hfst = ReadFstKaldiGeneric(fst_rx);
    k3_fst *fst = nullptr;
    K3H_CHECK_K3(k3_fst_create(hfst.NumStates(), hfst.start, hfst.arc_offsets.data(), hfst.ilabel.data(), hfst.olabel.data(), hfst.weight.data(), hfst.nextstate.data(),
                               hfst.final_cost.data(), ti.id2pdf.data(), (int32_t)ti.id2pdf.size(), &fst));
    k3_decoder_config dc; k3_decoder_config_default(&dc);
    dc.beam = beam; dc.lattice_beam = lattice_beam; dc.max_active = max_active; dc.min_active = std::min(min_active, max_active - 1); dc.beam_delta = beam_delta;
    dc.frame_tokens_cap = main_q > 0 ? main_q : std::min(65536, std::max(4 * max_active, 4096));
    dc.frame_cands_cap = aux_q > 0 ? std::max(aux_q, dc.frame_tokens_cap) : 3 * dc.frame_tokens_cap;
    dc.lane_tokens_cap = std::max<int64_t>(ntok_pre, dc.frame_tokens_cap); dc.lane_links_cap = 2 * dc.lane_tokens_cap;
    dc.literal_order = literal_order ? 1 : 0;
    dc.hash_ratio = hash_ratio;
    if (literal_order) {
      dc.frame_tokens_cap = std::min(dc.frame_tokens_cap, 65536);
      dc.frame_cands_cap = std::max(dc.frame_cands_cap, dc.frame_tokens_cap + 1);
    }
    const int nch = num_channels, N = ninfo.output_dim, C = frames_per_chunk / subsampling * subsampling > 0 ? frames_per_chunk / subsampling * subsampling : subsampling;
    k3_decoder *dec = nullptr; K3H_CHECK_K3(k3_decoder_create(fst, &dc, nch, N, &dec));
    K3H_CHECK_K3(k3_decoder_init_decoding(dec, nch, max_frames, nullptr));
    // ONE work stream for everything a round queues -- sample copy, features, row gathers, network, token passing -- fed from page-locked staging rings (DevBuf::upload_async,
    // k3_decoder_advance_decoding's argument ring): the host never waits for the device inside a round, so it prepares round k + 1 while the GPU runs round k; it only
    // synchronises where a stream ends and its lattice is fetched.  (A blocking stream: ordered with the few null-stream calls the C ABI still makes when lattices are fetched.)
    hipStream_t ws = nullptr; K3O_HIP(hipStreamCreate(&ws));
    // ... and a second stream for token passing.  A chunk's token-passing launch lasts as long as its SLOWEST lane (5 - 9 ms for 17 frames of 512 lanes whose mean is ~2 ms:
    // most CUs idle in its tail), so the next pass's features / gathers / network run beside it on `ws`, the way the offline pipeline hides its front end behind the decoder.
    // The only buffer the two streams share is the gathered log-likelihood block: two of them, guarded by events (ev_ll: filled, ev_tp: consumed).
    // (the decoder's stream at the highest priority the device offers: a chunk's launches are the critical path of a round -- consecutive launches of the same lanes --, the
    // network's GEMMs beside them are throughput work; K3_ONLINE_FLAT_PRIORITY=1: both streams at the default priority, for A/B)
    hipStream_t ds = nullptr;
    { int lo = 0, hi = 0; K3O_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      if (getenv("K3_ONLINE_FLAT_PRIORITY") || lo == hi) K3O_HIP(hipStreamCreate(&ds)); else K3O_HIP(hipStreamCreateWithPriority(&ds, hipStreamDefault, hi)); }
    hipEvent_t ev_ll[2], ev_tp[2];
    for (int k = 0; k < 2; k++) {
      K3O_HIP(hipEventCreateWithFlags(&ev_ll[k], hipEventDisableTiming));
      K3O_HIP(hipEventCreateWithFlags(&ev_tp[k], hipEventDisableTiming));
    }
    bool tp_used[2] = {false, false}; unsigned pass_no = 0;
    OnlineFeatures features(plan, fopts, nch, ws);
    StaticNnet3 net(nnet, nch, nch, C, subsampling, log_priors.empty() ? nullptr : log_priors.data(), acoustic_scale, ws);
    std::unique_ptr<OnlineIvectors> ivs; if (ivx) ivs.reset(new OnlineIvectors(ivx, iv_info.right_context, nch, ws));
    const int shift = (int)(fopts.samp_freq * 0.001 * fopts.frame_shift_ms), chunk_samples = C * shift;

    auto scp = ReadScp(wav_rspec);
    if (num_todo >= 0 && (size_t)num_todo < scp.size()) scp.resize(num_todo);
    // OpenOutputHandles (cudadecoderbin/cuda-bin-tools.h:181-195): an output argument that is not a table wspecifier names a .ctm file
    const bool ctm_mode = !(out_wspec.compare(0, 3, "ark") == 0 || out_wspec.compare(0, 3, "scp") == 0) || out_wspec.find(':') == std::string::npos;
    if (!write_lattice && !ctm_mode) K3H_LOG << "If you want to write lattices to disk, please set --write-lattice=true";
    std::shared_ptr<LatticePostprocessor> postprocessor; std::vector<std::string> syms;
    if (postproc.empty()) { if (ctm_mode) K3H_ERR << "You must configure the lattice postprocessor with --lattice-postprocessor-rxfilename to use CTM output"; }
    else {
      postprocessor = LoadLatticePostprocessor(postproc);
      postprocessor->SetDecoderFrameShift(fopts.frame_shift_ms * 1.0e-3f * subsampling);
      postprocessor->SetTransitionInformation(&ti);
    }
    if (!word_syms.empty()) {      // fst::SymbolTable::ReadText: lines "symbol id"
      std::istringstream in(ReadWholeInput(word_syms)); std::string sym; long id;
      while (in >> sym >> id) { if (id >= 0) { if ((size_t)id >= syms.size()) syms.resize((size_t)id + 1); syms[(size_t)id] = sym; } }
      if (syms.empty()) K3H_ERR << "Could not read symbol table from file " << word_syms;
    }
    std::unique_ptr<std::ofstream> ctm_file; if (ctm_mode) { ctm_file.reset(new std::ofstream(out_wspec)); if (!*ctm_file) K3H_ERR << "cannot open " << out_wspec; }
    std::unique_ptr<TableWriter> writer; if (write_lattice && !ctm_mode) writer.reset(new TableWriter(out_wspec));
    const bool want_lattice = writer || ctm_mode;      // result_type != 0 of the reference: a lattice (or the CTM made from it) is what the stream's latency waits for
    // latency of stream k: its result's time - the time its speaker stopped (paced streams) / its last chunk was submitted (unpaced); the reference's PrintLatencyStats
    std::vector<double> latencies; std::mutex lat_m; std::map<std::string, std::pair<size_t, double>> lat_pending;      // key -> (stream, stop time): closed by the determinization pool
    const auto t_clock0 = std::chrono::steady_clock::now();
    auto now_s = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_clock0).count(); };
    std::unique_ptr<DeterminizeSequencer> det_pool;          // lattices are determinized on worker threads while the streams go on
    if ((writer && (determinize || postprocessor)) || ctm_mode) {
      DeterminizeSequencer::Config pc; pc.num_threads = worker_threads > 0 ? worker_threads : std::max(1, (int)std::thread::hardware_concurrency());
      pc.beam = lattice_beam; pc.trans = &ti; pc.phone_det = det_opts; pc.determinize = determinize; pc.postprocessor = postprocessor; pc.ctm_out = ctm_file.get();
      pc.word_syms = syms.empty() ? nullptr : &syms;
      pc.on_done = [&](const std::string &key) {      // (a worker thread: the lattice / CTM of `key` exists now)
        std::lock_guard<std::mutex> l(lat_m); auto it = lat_pending.find(key);
        if (it != lat_pending.end()) { latencies[it->second.first] = now_s() - it->second.second; lat_pending.erase(it); }
      };
      det_pool.reset(new DeterminizeSequencer(pc, writer.get()));
    }
    // end-pointing (kaldi::EndpointDetected, online2/online-endpoint.cc:26-72) on the partial best path of a channel
    std::set<int32_t> sil_phones; { std::string tok; std::istringstream in(silence_phones); while (std::getline(in, tok, ':')) if (!tok.empty()) sil_phones.insert(std::stoi(tok)); }
    const float out_shift_s = fopts.frame_shift_ms * 1.0e-3f * subsampling;
    auto words_of = [&](const int32_t *ol, int64_t n) {
      std::string str;
      for (int64_t k = 0; k < n; k++) {
        if (ol[k] == 0) continue;
        if (!str.empty()) str += " ";
        str += ((size_t)ol[k] < syms.size() && !syms[ol[k]].empty()) ? syms[ol[k]] : std::to_string(ol[k]);
      }
      return str;
    };
    struct Paths { std::vector<int64_t> off; std::vector<int32_t> il, ol, rf; std::vector<float> g, ac, fc, rel; };
    auto best_paths = [&](const std::vector<int32_t> &channels, bool use_final, Paths *bp) {      // (synchronises with the decoder's stream)
      const int32_t n = (int32_t)channels.size(); bp->off.assign(n + 1, 0); bp->fc.assign(n, 0.f); bp->rel.assign(n, 0.f); bp->rf.assign(n, 0);
      int64_t cap = 0; for (int32_t c : channels) cap += 2 * (int64_t)k3_decoder_num_frames_decoded(dec, c) + 64;
      for (;;) {
        bp->il.resize(cap); bp->ol.resize(cap); bp->g.resize(cap); bp->ac.resize(cap);
        const int rc = k3_decoder_get_best_path(dec, channels.data(), n, use_final ? 1 : 0, bp->off.data(), cap, bp->il.data(), bp->ol.data(), bp->g.data(), bp->ac.data(),
            bp->fc.data(), bp->rel.data(), bp->rf.data());
        if (rc == K3_ERR_OVERFLOW) { cap *= 2; continue; }
        K3H_CHECK_K3(rc); break;
      }
    };
    int num_task = 0, num_err = 0; double total_audio = 0.0; size_t corr_cnt = 0;
    // per-channel state of the simulation
    // Feature rows a channel has computed but not yet fed to the network (less than a chunk, or waiting behind one) live COMPACTLY in one device buffer,
    // channel after channel, with
    // their (offset, count) on the host -- up to two segments per channel: what was left over + this round's new rows.  A pass takes its rows out with ONE row gather
    // (k3_mat_copy_rows) and the leftovers of all channels are gathered into the other buffer once per round: three launches per round where per-channel buffers cost ~2000
    // synchronous device copies per round (40 k copyBuffer calls = 46 % of the GPU time of a 512-channel run, 27 ms per round of 512 chunks).
    struct Chan { int utt = -1; Wave wav; size_t pos = 0; int pend = 0; bool started = false; int64_t seg_off[2] = {0, 0}; int seg_cnt[2] = {0, 0};
                  size_t corr_id = 0; double next_at = 0.0, stop_at = 0.0; };      // stream id (admission order); when its next chunk is spoken / its speaker stops (paced runs)
    std::vector<Chan> chan(nch);
    DevBuf<float> held[2], newbuf; DevBuf<int32_t> gidx; int held_cur = 0; int64_t held_rows = 0;
    const size_t pend_cap = (size_t)(2 * C + 8);
    for (auto &h : held) h.need((size_t)nch * (pend_cap + (size_t)C + 16) * fdim);
    auto take_rows = [](Chan &c, int n, std::vector<int32_t> *idx) {      // the first n pending rows of the channel, in order
      for (int sgm = 0; sgm < 2 && n > 0; sgm++) {
        const int k = std::min(n, c.seg_cnt[sgm]);
        for (int j = 0; j < k; j++) idx->push_back((int32_t)(c.seg_off[sgm] + j));
        c.seg_off[sgm] += k;
        c.seg_cnt[sgm] -= k;
        n -= k;
        c.pend -= k;
      }
    };
    WavePrefetcher prefetch(scp, iterations, (size_t)2 * nch + 8, 8);
    std::mt19937 rng(std::random_device{}()); std::uniform_real_distribution<double> start_jitter(0.0, 1.0); bool add_random_offset = simulate_rt;
    const auto t_start = std::chrono::steady_clock::now();
    for (int iter = 0; iter < iterations; iter++) {
      std::deque<int> queue; for (size_t i = 0; i < scp.size(); i++) queue.push_back((int)i);
      int busy = 0;
      while (!queue.empty() || busy > 0) {
        // admit files to free channels
        for (int ch = 0; ch < nch && !queue.empty(); ch++) {
          if (chan[ch].utt >= 0) continue;
          const int u = queue.front(); queue.pop_front();
          Wave w;
          if (!prefetch.Get((size_t)iter * scp.size() + (size_t)u, &w)) { num_err++; ch--; continue; }
          if (w.samp_freq != fopts.samp_freq) { K3H_WARN << "Sample frequency mismatch for " << scp[u].first; num_err++; ch--; continue; }
          if (k3_feat_num_frames(plan, (int64_t)w.samples.size()) == 0) { K3H_WARN << "Utterance " << scp[u].first << " is too short to decode"; num_err++; ch--; continue; }
          total_audio += w.samples.size() / (double)w.samp_freq; num_task++;
          chan[ch] = Chan(); chan[ch].utt = u; chan[ch].wav = std::move(w); busy++;
          chan[ch].corr_id = corr_cnt++; latencies.push_back(0.0);
          {      // the stream starts now (the first ones: somewhere in the first second, batched-wav-nnet3-cuda-online.cc:150-177); its first chunk exists once it has been spoken
            const double dur = chan[ch].wav.samples.size() / (double)chan[ch].wav.samp_freq, start = now_s() + (add_random_offset ? start_jitter(rng) : 0.0);
            chan[ch].next_at = start + std::min(dur, chunk_samples / (double)fopts.samp_freq); chan[ch].stop_at = start + dur;
          }
        }
        add_random_offset = false;
        if (busy == 0) break;
        if (simulate_rt) {      // nothing to submit before the earliest chunk has been spoken
          double first = 1e300; for (int ch = 0; ch < nch; ch++) if (chan[ch].utt >= 0) first = std::min(first, chan[ch].next_at);
          const double wait = first - now_s(); if (wait > 0) std::this_thread::sleep_for(std::chrono::duration<double>(wait));
        }
        const double t_round = now_s();
        // one chunk of audio per busy channel
        std::vector<int> chs; std::vector<const float *> chunk_ptr; std::vector<size_t> chunk_len; std::vector<char> first, last;
        for (int ch = 0; ch < nch; ch++) {
          Chan &c = chan[ch]; if (c.utt < 0) continue;
          if (simulate_rt && c.next_at > t_round) continue;      // (its next chunk has not been spoken yet)
          const size_t n = std::min<size_t>(chunk_samples, c.wav.samples.size() - c.pos);
          chs.push_back(ch); chunk_ptr.push_back(c.wav.samples.data() + c.pos); chunk_len.push_back(n);      // (the chunk is read where it lies: no copy out of the waveform)
          first.push_back(c.pos == 0); c.pos += n; last.push_back(c.pos == c.wav.samples.size());
          c.next_at += std::min<size_t>(chunk_samples, c.wav.samples.size() - c.pos) / (double)fopts.samp_freq;
          if (!simulate_rt && last.back()) c.stop_at = t_round;
        }
        if (chs.empty()) continue;
        std::vector<int32_t> fresh; for (size_t i = 0; i < chs.size(); i++) if (first[i]) { fresh.push_back(chs[i]); net.Reset(chs[i]); }
        if (!fresh.empty()) K3H_CHECK_K3(k3_decoder_init_channels(dec, fresh.data(), (int32_t)fresh.size(), ds));
        float *d_feats = nullptr;
        const std::vector<int> nf = features.ComputeFeaturesBatched(chs, chunk_ptr.data(), chunk_len.data(), first, &d_feats);
        { int64_t off = 0, tot = 0; for (int n : nf) tot += n;
          if ((size_t)(held_rows + tot) * fdim > held[held_cur].cap) K3H_ERR << "internal: pending-frame buffer";
          // the round's new rows behind the leftovers
          if (tot > 0) K3O_HIP(hipMemcpyAsync(held[held_cur].p + (size_t)held_rows * fdim, d_feats, (size_t)tot * fdim * 4, hipMemcpyDeviceToDevice, ws));
          for (size_t i = 0; i < chs.size(); i++) {
            Chan &c = chan[chs[i]];
            if ((size_t)(c.pend + nf[i]) > pend_cap) K3H_ERR << "internal: pending-frame buffer";
            if (c.seg_cnt[1] != 0) K3H_ERR << "internal: pending rows not compacted";
            c.seg_off[1] = held_rows + off; c.seg_cnt[1] = nf[i]; c.pend += nf[i]; off += nf[i];
          } }
        if (ivs) ivs->AcceptBatch(chs, d_feats, nf, first, last);      // the extractor sees every frame as soon as it exists: all channels of the batch in one launch per stage
        std::vector<char> is_last(nch, 0), closed(nch, 0); for (size_t i = 0; i < chs.size(); i++) is_last[chs[i]] = last[i];
        bool need_advance = !fresh.empty();
        while (true) {
          std::vector<int> run; std::vector<int> n_new; std::vector<char> lasts;
          for (int ch : chs) if (!closed[ch] && (chan[ch].pend >= C || is_last[ch])) run.push_back(ch);
          if (run.empty() && !need_advance) break;
          int64_t tot_new = 0;
          for (int ch : run) { const int n = std::min(C, chan[ch].pend); n_new.push_back(n); tot_new += n; }
          newbuf.need((size_t)std::max<int64_t>(tot_new, 1) * fdim);
          { std::vector<int32_t> take; take.reserve((size_t)tot_new);
            for (size_t i = 0; i < run.size(); i++) {
              Chan &c = chan[run[i]]; const int n = n_new[i];
              take_rows(c, n, &take);
              const bool end = is_last[run[i]] && c.pend == 0; lasts.push_back(end); if (end) closed[run[i]] = 1;
              c.started = true;
            }
            if (!take.empty()) { gidx.upload_async(take, ws); K3H_CHECK_K3(k3_mat_copy_rows(newbuf.p, fdim, (int32_t)take.size(), fdim, held[held_cur].p, fdim, gidx.p, ws)); } }
          // The pass's log-likelihoods are decoded WHERE THE NETWORK LEFT THEM (k3_decoder_advance_decoding_strided: a first row, a count and a row distance per channel) -- no
          // gather into one block (210 MB per pass of 512 channels).  The network writes two output buffers in turn: pass k + 1 is queued behind the token-passing launch
          // that read the buffer two passes ago (ev_tp), the launch behind the pass that fills its buffer (ev_ll).
          const int lb = (int)(pass_no++ & 1);
          std::vector<const float *> lane_first(nch, nullptr); std::vector<int32_t> lane_frames(nch, 0); int64_t ld_rows = N;
          if (!run.empty()) {
            if (tp_used[lb]) K3O_HIP(hipStreamWaitEvent(ws, ev_tp[lb], 0));
            net.SelectOut(lb);
            auto res = net.Pass(run, newbuf.p, n_new, lasts, ivs ? ivs->Gather(run) : nullptr);
            // end of stream: frames still waiting for right context may take more passes
            for (size_t i = 0; i < run.size(); i++) if (res[i].count > 0) {
              lane_first[run[i]] = net.Out() + (size_t)res[i].first * N;
              lane_frames[run[i]] = res[i].count;
              ld_rows = (int64_t)res[i].stride * N;
            }
          }
          K3O_HIP(hipEventRecord(ev_ll[lb], ws)); K3O_HIP(hipStreamWaitEvent(ds, ev_ll[lb], 0));
          K3H_CHECK_K3(k3_decoder_advance_decoding_strided(dec, nch, lane_first.data(), lane_frames.data(), ld_rows, ds));
          K3O_HIP(hipEventRecord(ev_tp[lb], ds)); tp_used[lb] = true;
          need_advance = false;
          // flush passes for closed channels whose last outputs did not fit one pass
          for (int ch : run) if (closed[ch] && net.Pending(ch)) { closed[ch] = 0; }      // stays in `run` candidates: is_last and pend == 0 -> another (empty-input) pass
        }
        {      // the rows still waiting, of all channels, into the other buffer (one gather); every channel is back to one segment
          std::vector<int32_t> keep; int64_t at = 0;
          for (int ch = 0; ch < nch; ch++) {
            Chan &c = chan[ch]; if (c.utt < 0 || c.pend == 0) { c.seg_cnt[0] = c.seg_cnt[1] = 0; c.pend = c.utt < 0 ? 0 : c.pend; continue; }
            const int n = c.pend; for (int sgm = 0; sgm < 2; sgm++) for (int j = 0; j < c.seg_cnt[sgm]; j++) keep.push_back((int32_t)(c.seg_off[sgm] + j));
            c.seg_off[0] = at; c.seg_cnt[0] = n; c.seg_off[1] = 0; c.seg_cnt[1] = 0; at += n;
          }
          if (!keep.empty()) {
            gidx.upload_async(keep, ws);
            K3H_CHECK_K3(k3_mat_copy_rows(held[held_cur ^ 1].p, fdim, (int32_t)keep.size(), fdim, held[held_cur].p, fdim, gidx.p, ws));
          }
          held_cur ^= 1; held_rows = at;
        }
        // the BestPathCallback of the reference (batched-threaded-nnet3-cuda-online-pipeline.cc:455-475), for the streams that go on: partial hypothesis and end-point
        if (print_partial || print_endpoints) {
          std::vector<int32_t> cont; for (size_t i = 0; i < chs.size(); i++) if (!last[i] && k3_decoder_num_frames_decoded(dec, chs[i]) > 0) cont.push_back(chs[i]);
          if (!cont.empty()) {
            Paths bp; best_paths(cont, false, &bp);
            for (size_t u = 0; u < cont.size(); u++) {
              const Chan &c = chan[cont[u]];
              if (print_partial) K3H_LOG << "corr_id #" << c.corr_id << " [partial] : " << words_of(bp.ol.data() + bp.off[u], bp.off[u + 1] - bp.off[u]);
              if (print_endpoints) {
                int32_t sil = 0, frames = 0;
                for (int64_t k = bp.off[u + 1] - 1; k >= bp.off[u]; k--) {
                  const int32_t t = bp.il[k]; if (t == 0) continue;
                  if (!((size_t)t < ti.id2phone.size() && sil_phones.count(ti.id2phone[t]))) break;
                  sil++;
                }
                for (int64_t k = bp.off[u]; k < bp.off[u + 1]; k++) frames += bp.il[k] != 0;
                const float len = frames * out_shift_s, ts = sil * out_shift_s; bool ep = false;
                for (const Rule &r : rules) ep = ep || (((len > ts) || !r.nonsil) && ts >= r.sil && bp.rel[u] <= r.rel && len >= r.len);
                if (ep) K3H_LOG << "corr_id #" << c.corr_id << " [endpoint detected]";
              }
            }
          }
        }
        // finalise the channels whose stream ended, write their lattices, free the channels
        std::vector<int32_t> ended; for (size_t i = 0; i < chs.size(); i++) if (last[i]) ended.push_back(chs[i]);
        if (!ended.empty()) {
          K3H_CHECK_K3(k3_decoder_finalize_channels(dec, ended.data(), (int32_t)ended.size(), ds));
          if (print_hyp || !want_lattice) {      // the final best path (with final-probs): the result a stream without a lattice callback waits for
            Paths bp; best_paths(ended, true, &bp); const double t_res = now_s();
            for (size_t u = 0; u < ended.size(); u++) {
              const Chan &c = chan[ended[u]];
              if (!want_lattice) latencies[c.corr_id] = t_res - c.stop_at;
              if (print_hyp) K3H_LOG << "corr_id #" << c.corr_id << " : " << words_of(bp.ol.data() + bp.off[u], bp.off[u + 1] - bp.off[u]);
            }
          }
          const int U = (int)ended.size();
          std::vector<int64_t> info(10 * (size_t)U); K3H_LATTICE_INFO(dec, info.data());
          int64_t NS = 0, NA = 0; for (int u = 0; u < U; u++) { NS += info[10 * u]; NA += info[10 * u + 1]; }
          std::vector<int32_t> sf(NS + 1), ss(NS + 1), as(NA + 1), ad(NA + 1), ai(NA + 1), ao(NA + 1); std::vector<float> sc(NS + 1), sfin(NS + 1), ag(NA + 1), aa(NA + 1);
          if (NS > 0 && want_lattice) K3H_CHECK_K3(k3_decoder_get_raw_lattices(dec, sf.data(), ss.data(), sc.data(), sfin.data(), as.data(), ad.data(), ai.data(), ao.data(),
              ag.data(), aa.data()));
          int64_t s0 = 0, a0 = 0;
          for (int u = 0; u < U; u++) {
            const int64_t ns = info[10 * u], na = info[10 * u + 1]; const std::string &key = scp[chan[ended[u]].utt].first;
            if (info[10 * u + 2] != 0 || ns == 0) { K3H_WARN << "Failed to decode utterance with id " << key; num_err++; }
            else if (want_lattice && !(iter == 0) ) { latencies[chan[ended[u]].corr_id] = now_s() - chan[ended[u]].stop_at; }      // (later iterations: decoded, not written)
            else if (iter == 0 && want_lattice) {
              if (!info[10 * u + 3]) K3H_WARN << "Outputting partial output for utterance " << key << " since no final-state reached";
              Lattice lat;
              lat.st_frame.assign(sf.begin() + s0, sf.begin() + s0 + ns);
              lat.st_state.assign(ss.begin() + s0, ss.begin() + s0 + ns);
              lat.st_final.assign(sfin.begin() + s0, sfin.begin() + s0 + ns);
              lat.arc_src.assign(as.begin() + a0, as.begin() + a0 + na);
              lat.arc_dst.assign(ad.begin() + a0, ad.begin() + a0 + na);
              lat.arc_ilabel.assign(ai.begin() + a0, ai.begin() + a0 + na);
              lat.arc_olabel.assign(ao.begin() + a0, ao.begin() + a0 + na);
              lat.arc_graph.assign(ag.begin() + a0, ag.begin() + a0 + na);
              lat.arc_ac.assign(aa.begin() + a0, aa.begin() + a0 + na);
              for (int64_t s = 0; s < ns; s++) if (lat.st_frame[s] == 0 && lat.st_state[s] == hfst.start) lat.start = (int32_t)s;
              Connect(&lat);
              if (det_pool) { { std::lock_guard<std::mutex> l(lat_m); lat_pending[key] = std::make_pair(chan[ended[u]].corr_id, chan[ended[u]].stop_at); } det_pool->Run(key, std::move(lat)); }
              else {
                if (write_compact) { CompactLattice clat; ConvertLattice(lat, &clat); writer->WriteCompactLattice(key, clat); }
                else writer->WriteLattice(key, lat);
                latencies[chan[ended[u]].corr_id] = now_s() - chan[ended[u]].stop_at;
              }
            }
            s0 += ns; a0 += na;
            chan[ended[u]] = Chan(); busy--;
          }
        }
      }
    }
    K3O_HIP(hipDeviceSynchronize());
    if (det_pool) { det_pool->Wait(); det_pool.reset(); }
    if (writer) writer->Flush();
    const double total_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    {      // PrintLatencyStats (cudadecoderbin/cuda-bin-tools.h:33-55)
      K3H_LOG << "Latency stats:";
      std::vector<double> lat; { std::lock_guard<std::mutex> l(lat_m); lat = latencies; }
      if (!lat.empty()) {
        double total = 0.0; for (double x : lat) total += x;
        std::sort(lat.begin(), lat.end()); const double n = (double)lat.size();
        const double l90 = lat[(size_t)std::floor(90. * n / 100.)], l95 = lat[(size_t)std::floor(95. * n / 100.)], l99 = lat[(size_t)std::floor(99. * n / 100.)];
        K3H_LOG << "Latencies (s):\tAvg\t\t90%\t\t95%\t\t99%";
        std::ostringstream os; os << std::fixed; os.precision(3); os << "\t\t\t" << total / n << "\t\t" << l90 << "\t\t" << l95 << "\t\t" << l99;
        K3H_LOG << os.str();
      }
    }
    K3H_LOG << "Decoded " << num_task << " utterances, " << num_err << " with errors.";
    K3H_LOG << "Overall: " << " Aggregate Total Time: " << total_time << " Total Audio: " << total_audio << " RealTimeX: " << total_audio / total_time;
    ivs.reset(); if (ivx) k3_ivector_destroy(ivx);
    (void)hipStreamDestroy(ws); (void)hipStreamDestroy(ds); for (int k = 0; k < 2; k++) { (void)hipEventDestroy(ev_ll[k]); (void)hipEventDestroy(ev_tp[k]); }
    k3_decoder_destroy(dec); k3_fst_destroy(fst); k3_nnet_destroy(nnet); k3_feat_plan_destroy(plan);
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
