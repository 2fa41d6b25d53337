// batched-wav-nnet3-cuda2 -- drop-in for cudadecoderbin/batched-wav-nnet3-cuda2.cc:51-260 on MI355X:
//   batched-wav-nnet3-cuda2 [options] <nnet3-in> <fst-in> <wav-rspecifier> <lattice-wspecifier>
// Same positional arguments, same option names (BatchedThreadedNnet3CudaPipeline2Config and its nested configs), same exit codes
// (1 usage, -1 exception) and the same closing log line "Overall:  Aggregate Total Time: .. Total Audio: .. RealTimeX: ..".
// The pipeline behind it is the whole-utterance batch path of libk3hip.so: waveforms -> k3_feat_compute_batch -> k3_nnet_forward ->
// k3_decoder_decode_batch -> raw lattices -> (default, like BatchedThreadedNnet3CudaOnlinePipelineConfig::determinize_lattice) phone- then word-level
// pruned determinization on the host (k3_lattice.cc; beam = --lattice-beam, batched-threaded-nnet3-cuda-online-pipeline.cc:759-765) ->
// CompactLattice table.  Like the reference's CUDA pipeline the lattice keeps the acoustic scale it was decoded with (only the CPU
// decoders' wrappers undo it).  --determinize-lattice=false writes the trimmed state-level lattice re-packed by ConvertLattice as a
// CompactLattice, as the reference does; with --write-compact=false (an addition) it is written as a Lattice table, arc for arc.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <chrono>
#include <fstream>
#include <sstream>
#include <cmath>
#include <cstring>
#include <array>
#include <functional>
#include <future>
#include <iostream>
#include <thread>
#include "k3_feat_options.h"
#include "k3_online.h"
using namespace k3host;
#define HIPCHK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)

int main(int argc, char **argv) {
  try {
    const char *slash = strrchr(argv[0], '/'); const std::string prog = slash ? slash + 1 : argv[0];
    const bool v1 = prog.size() >= 22 && prog.compare(prog.size() - 22, 22, "batched-wav-nnet3-cuda") == 0;      // the first-generation driver's name (no trailing 2)
    const char *usage = v1 ?
        "Reads in wav file(s) and simulates online decoding with neural nets\n(nnet3 setup), with optional iVector-based speaker adaptation and\noptional endpointing.  Note: some configuration values and inputs are\n"
        "set via config files whose filenames are passed as options\n\nUsage: batched-wav-nnet3-cuda [options] <nnet3-in> <fst-in> <wav-rspecifier> <lattice-wspecifier>\n" :
        "Reads in wav file(s) and decodes them with neural nets\n(nnet3 setup).  Note: some configuration values and inputs are\n"
        "set via config files whose filenames are passed as options\nOutput can either be a lattice wspecifier or a ctm filename\n"
        "Usage: batched-wav-nnet3-cuda2 [options] <nnet3-in> <fst-in> <wav-rspecifier> <lattice-wspecifier|ctm-wxfilename>\n";
    ParseOptions po(usage);
    int32_t v1_drain = 10, v1_control = 2, v1_pending = 4000;
    if (v1) {      // BatchedThreadedNnet3CudaPipelineConfig::Register (batched-threaded-nnet3-cuda-pipeline.h:65-110): the knobs of the v1 class's task queue
      po.Register("batch-drain-size", &v1_drain, "How far to drain the batch before refilling work. (accepted: batches here are whole --max-batch-size groups of utterances)");
      po.Register("cuda-control-threads", &v1_control,
          "The number of pipeline control threads for the CUDA work. (accepted: one control thread drives the front-end and decoder streams here)");
      po.Register("max-outstanding-queue-length", &v1_pending, "Number of files to allow to be outstanding at a time. (accepted: two batches are in flight)");
    }
    bool write_compact = true, write_lattice = true, segmentation = false, determinize = true, minimize = false, phone_det = true, word_det = true,
        gpu_feat = true, use_online = false, reset_on_endpoint = false, tensor_cores = false, tf32 = false, add_pitch = false, debug_comp = false,
        cache_mem = true;
    bool alternate_decoders = true;
    int32_t num_todo = -1, iterations = 1, max_batch = 400, num_channels = -1, worker_threads = -1, copy_threads = 2, frames_per_chunk = 50, subsampling = 1;
    int32_t max_active = 10000, min_active = 200, main_q = -1, aux_q = -1, ntok_pre = 1000000, elc = 0, erc = 0, elci = -1, ercf = -1;
    float beam = 15.0f, lattice_beam = 10.0f, acoustic_scale = 0.1f, beam_delta = 0.5f, det_delta = 1.0f / 1024.0f; double mem_prop = 0.5; int32_t det_max_mem = 50000000;
    bool literal_order = true;
    float hash_ratio = 2.0f;
    bool pin_cores = true;
    int32_t rank = getenv("RANK") ? atoi(getenv("RANK")) : 0, world_size = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1, device = -1;
    std::string nccl_id_file;
    std::string word_syms, postproc, feature_type = "mfcc", mfcc_config, fbank_config, plp_config, pitch_config, cmvn_config, global_cmvn, ivector_config, use_gpu = "yes";
    po.Register("write-lattice", &write_lattice, "Output lattice to a file. Setting to false is useful when benchmarking");
    po.Register("word-symbol-table", &word_syms, "Symbol table for words [the words of the CTM output]");
    po.Register("file-limit", &num_todo, "Limits the number of files that are processed by this driver.");
    po.Register("iterations", &iterations, "Number of times to decode the corpus. Output will be written only once.");
    po.Register("segmentation", &segmentation, "Split audio files into segments");
    // CudaPipelineSegmentationConfig (cudadecoder/cuda-pipeline-common.h:39-60)
    double segment_length_s = 20, segment_overlap_s = 1, min_segment_length_s = 1;
    po.Register("segment-length", &segment_length_s, "Segment length (s)"); po.Register("segment-overlap", &segment_overlap_s, "Overlap between segments (s)");
    po.Register("min-segment-length", &min_segment_length_s, "Min segment length (s, >=1)");
    po.Register("lattice-postprocessor-rxfilename", &postproc,
        "(optional) Config file for lattice postprocessor (scales, word insertion penalty, MBR options; needed for CTM output)");
    po.Register("max-batch-size", &max_batch, "The maximum execution batch size (utterances decoded together)");
    po.Register("alternate-decoders", &alternate_decoders,
        "Keep two decoder objects (twice the lane pools in HBM) and alternate them batch by batch: a batch's token passing then starts while the previous batch's lattice pruning, compaction and copy to the host are still running");
    po.Register("num-channels", &num_channels, "(accepted; whole-utterance batching needs no separate channel pool)");
    po.Register("cuda-worker-threads", &worker_threads,
        "The total number of CPU threads launched to process CPU tasks (here: lattice determinization). -1 = use std::hardware_concurrency().");
    po.Register("cuda-decoder-copy-threads", &copy_threads, "Number of worker threads that read the wave files and fill the pinned staging buffers.");
    po.Register("determinize-lattice", &determinize, "Determinize the lattice before output.");
    po.Register("write-compact", &write_compact,
        "(not in the reference) with --determinize-lattice=false: true = the state-level lattice re-packed as a CompactLattice like the reference (ConvertLattice), false = written as a Lattice table");
    po.Register("delta", &det_delta, "Tolerance used in determinization");
    po.Register("max-mem", &det_max_mem, "Maximum approximate memory usage in determinization (real usage might be many times this).");
    po.Register("phone-determinize", &phone_det, "If true, do an initial pass of determinization on both phones and words (see also --word-determinize)");
    po.Register("word-determinize", &word_det, "If true, do a second pass of determinization on words only (see also --phone-determinize)");
    po.Register("minimize", &minimize, "If true, push and minimize after determinization.");
    po.Register("gpu-feature-extract", &gpu_feat, "Use GPU feature extraction (always true)"); po.Register("use-online-features", &use_online, "(only false is supported)");
    po.Register("reset-on-endpoint", &reset_on_endpoint, "(accepted, unused: offline decoding)");
    po.Register("beam", &beam, "Decoding beam. Larger->slower, more accurate."); po.Register("lattice-beam", &lattice_beam, "The width of the lattice beam");
    po.Register("max-active", &max_active, "Decoder max active states. Larger->slower; more accurate");
    po.Register("min-active", &min_active, "Decoder min active states (LatticeFasterDecoderConfig)");
    po.Register("beam-delta", &beam_delta, "Increment used when the active-state limits move the beam (LatticeFasterDecoderConfig)");
    po.Register("main-q-capacity", &main_q, "Max tokens alive on one frame of one utterance (-1 = 4 * max-active, capped)");
    po.Register("aux-q-capacity", &aux_q, "Max arcs considered on one frame (-1 = 3 * main-q-capacity)");
    po.Register("ntokens-pre-allocated", &ntok_pre,
        "Advanced - Number of tokens pre-allocated in host buffers to store lattices. If this size is exceeded the buffer will reallocate (here: an utterance that outgrows it moves to bigger token / link pools inside the decoder kernel)");
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic log-likelihoods");
    po.Register("frame-subsampling-factor", &subsampling,
        "Required if the frame-rate of the output (e.g. in 'chain' models) is less than the frame-rate of the original alignment.");
    po.Register("frames-per-chunk", &frames_per_chunk,
        "Number of frames in each chunk that is separately evaluated by the neural net (matters with i-vectors: one i-vector per chunk; without them chunking does not change the outputs of a feed-forward model and utterances are evaluated whole)");
    po.Register("extra-left-context", &elc, "(accepted; only 0 is supported)"); po.Register("extra-right-context", &erc, "(accepted; only 0 is supported)");
    po.Register("extra-left-context-initial", &elci, "(accepted)");
    po.Register("extra-right-context-final", &ercf, "(accepted)");
    po.Register("debug-computation", &debug_comp, "(accepted, unused)");
    po.Register("feature-type", &feature_type, "Base feature type [mfcc, fbank]");
    po.Register("mfcc-config", &mfcc_config, "Configuration file for MFCC features (e.g. conf/mfcc.conf)");
    po.Register("fbank-config", &fbank_config, "Configuration file for filterbank features (e.g. conf/fbank.conf)");
    po.Register("plp-config", &plp_config, "(PLP features are not supported)");
    po.Register("add-pitch", &add_pitch, "(pitch features are not supported)"); po.Register("online-pitch-config", &pitch_config, "(not supported)");
    po.Register("cmvn-config", &cmvn_config, "(online CMVN is not supported; chain recipes use --norm-means=false)");
    po.Register("global-cmvn-stats", &global_cmvn, "(not supported)");
    po.Register("ivector-extraction-config", &ivector_config,
        "Configuration file for online iVector extraction, see class OnlineIvectorExtractionConfig in the code.  The i-vectors are extracted on the GPU, one per --ivector-period frames, and every "
                "--frames-per-chunk chunk of the network sees the one at its middle frame: the results of ivector-extract-online2 | nnet3-latgen-faster --online-ivectors (steps/online/nnet2/extract_ivectors_online.sh + steps/nnet3/decode.sh)");
    po.Register("literal-order", &literal_order,
        "(not in the reference) true = raw lattices identical to the CPU LatticeFasterDecoder's, bit for bit (serial cutoff tightening in hash-list order reproduced on the GPU); false = the order-independent fast decoder");
    po.Register("hash-ratio", &hash_ratio, "LatticeFasterDecoderConfig::hash_ratio (it decides the reference's token visit order; used with --literal-order)");
    po.Register("rank", &rank,
        "(not in the reference) this process's rank in a multi-GPU job: it takes the utterances i with i % world-size == rank, uses GPU <rank> of the node unless LOCAL_RANK / --device says otherwise, and writes the lattice wspecifier with JOB replaced by rank + 1 (lat.JOB.gz of decode.sh).  Default: $RANK or 0");
    po.Register("pin-cores", &pin_cores,
        "(not in the reference) with world-size > 1: this rank's threads run on its own contiguous share of the host's cores (cores / world-size, by LOCAL_RANK), "
                "and --cuda-worker-threads defaults to that share instead of every core -- eight ranks' determinizers do not oversubscribe a 256-core host");
    po.Register("world-size", &world_size, "(not in the reference) number of ranks (one process per GPU).  Default: $WORLD_SIZE or 1");
    po.Register("device", &device, "(not in the reference) HIP device of this process (-1: $LOCAL_RANK, else rank modulo the number of devices)");
    po.Register("nccl-id-file", &nccl_id_file,
        "(not in the reference) with world-size > 1: rank 0 reads the graph and broadcasts it once over RCCL/xGMI to the other ranks; the communicator's id travels through this file (shared directory).  Empty: every rank reads the graph itself");
    po.Register("use-gpu", &use_gpu, "(accepted; always the GPU)"); po.Register("cuda-use-tensor-cores", &tensor_cores, "(accepted, unused: FP32 matrix cores are always used)");
    po.Register("cuda-use-tf32-compute", &tf32, "(accepted, unused: gfx950 has no tf32/xf32)");
    po.Register("cuda-cache-memory", &cache_mem, "(accepted, unused)");
    po.Register("cuda-memory-proportion", &mem_prop, "(accepted, unused)");
    po.Read(argc, argv);
    if (po.NumArgs() != 4) { po.PrintUsage(); return 1; }
    DeterminizeLatticePhonePrunedOptions det_opts;
    det_opts.delta = det_delta;
    det_opts.max_mem = det_max_mem;
    det_opts.phone_determinize = phone_det;
    det_opts.word_determinize = word_det;
    det_opts.minimize = minimize;
    if (use_online || add_pitch || !plp_config.empty() || !cmvn_config.empty() || !global_cmvn.empty() || elc || erc)
      K3H_ERR << "an option that needs a component outside the accelerated path was given (online features / pitch / PLP / CMVN / extra context)";
    const std::string nnet3_rx = po.GetArg(1), fst_rx = po.GetArg(2), wav_rspec = po.GetArg(3); std::string out_wspec = po.GetArg(4);
    if (world_size < 1 || rank < 0 || rank >= world_size) K3H_ERR << "--rank=" << rank << " is not in [0, --world-size=" << world_size << ")";
    int cores_of_rank = std::max(1, (int)std::thread::hardware_concurrency());
    if (world_size > 1 && pin_cores) {      // before any worker thread exists: they inherit the mask
      const int ncpu = cores_of_rank, lr = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank,
          lw = getenv("LOCAL_WORLD_SIZE") ? std::max(1, atoi(getenv("LOCAL_WORLD_SIZE"))) : world_size;
      const int share = std::max(1, ncpu / lw), first = (lr % lw) * share;
      cpu_set_t set; CPU_ZERO(&set); for (int c = first; c < first + share && c < ncpu; c++) CPU_SET(c, &set);
      if (sched_setaffinity(0, sizeof set, &set) == 0) { cores_of_rank = share; K3H_LOG << "rank " << rank << ": host threads on cores " << first << " .. " << first + share - 1; }
      else K3H_WARN << "sched_setaffinity failed; the rank's threads stay on all cores";
    }
    { int ndev = 0; HIPCHK(hipGetDeviceCount(&ndev)); if (ndev < 1) K3H_ERR << "no HIP device";
      if (device < 0) device = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) % ndev : rank % ndev;
      if (device >= ndev) K3H_ERR << "--device=" << device << " but the node has " << ndev << " devices";
      HIPCHK(hipSetDevice(device)); }
    for (size_t q; (q = out_wspec.find("JOB")) != std::string::npos;) out_wspec.replace(q, 3, std::to_string(rank + 1));      // utils/run.pl convention

    // feature options come from the config file named for the selected feature type, like OnlineNnet2FeaturePipelineInfo
    const bool mfcc = feature_type == "mfcc";
    if (!mfcc && feature_type != "fbank") K3H_ERR << "Invalid feature type: " << feature_type << " (supported: mfcc, fbank)";
    FeatOptions fo(mfcc);
    { ParseOptions fpo(""); fo.Register(&fpo); const std::string &cfg = mfcc ? mfcc_config : fbank_config; if (!cfg.empty()) fpo.ReadConfigFile(cfg); }
    const k3_feat_opts &fopts = fo.Finish();
    k3_feat_plan *plan = nullptr; K3H_CHECK_K3(k3_feat_plan_create(&fopts, &plan));
    const int fdim = k3_feat_dim(plan);

    // model: TransitionModel (tid -> pdf) + AmNnetSimple
    TransitionInfo ti = ReadTransitionModel(nnet3_rx);
    k3_nnet *nnet = nullptr; K3H_CHECK_K3(k3_nnet_load(nnet3_rx.c_str(), &nnet));
    k3_nnet_info ninfo; K3H_CHECK_K3(k3_nnet_get_info(nnet, &ninfo));
    if (ninfo.input_dim != fdim) K3H_ERR << "Feature dimension " << fdim << " does not match the model's input dimension " << ninfo.input_dim;
    // i-vector extractor (optional): OnlineNnet2FeaturePipelineInfo's ivector_extractor_info (online2/online-nnet2-feature-pipeline.cc:70-80)
    k3_ivector *ivx = nullptr; IvectorExtractionInfo iv_info; int32_t iv_period = 0;
    if (!ivector_config.empty()) {
      iv_info = ReadIvectorExtractionConfig(ivector_config);
      ivx = CreateIvectorExtractor(iv_info, fdim); iv_period = iv_info.ivector_period;
    }
    if ((ninfo.ivector_dim > 0) != (ivx != nullptr) || (ivx && ninfo.ivector_dim != iv_info.ie.ivector_dim))
      K3H_ERR << "Neural net expects 'ivector' features with dimension " << ninfo.ivector_dim << " but you provided " << (ivx ? iv_info.ie.ivector_dim : 0);
    if (ninfo.output_dim != ti.num_pdfs) K3H_ERR << "Model output dimension " << ninfo.output_dim << " != number of pdfs in the transition model " << ti.num_pdfs;
    std::vector<float> log_priors;
    if (ninfo.has_priors) { log_priors.resize(ninfo.output_dim); K3H_CHECK_K3(k3_nnet_get_priors(nnet, log_priors.data())); for (float &p : log_priors) p = logf(p); }

    // decoding graph
    // (multi-GPU: read and converted once, on rank 0, then one RCCL broadcast of the device image; the start state travels with it)
    HostFst hfst; k3_fst *fst = nullptr; const bool bcast = world_size > 1 && !nccl_id_file.empty();
    if (!bcast || rank == 0) {
      hfst = ReadFstKaldiGeneric(fst_rx);
      K3H_CHECK_K3(k3_fst_create(hfst.NumStates(), hfst.start, hfst.arc_offsets.data(), hfst.ilabel.data(), hfst.olabel.data(), hfst.weight.data(), hfst.nextstate.data(),
                                 hfst.final_cost.data(), ti.id2pdf.data(), (int32_t)ti.id2pdf.size(), &fst));
    }
    if (bcast) {
      void *comm = nullptr; K3H_CHECK_K3(k3_comm_create(nccl_id_file.c_str(), rank, world_size, 600, &comm));
      K3H_CHECK_K3(k3_fst_bcast(&fst, comm, 0, rank, nullptr)); k3_comm_destroy(comm);
      K3H_LOG << "rank " << rank << ": decoding graph " << (rank == 0 ? "sent" : "received") << " over RCCL (" << k3_fst_num_states(fst) << " states, " <<
          k3_fst_num_arcs(fst) << " arcs)";
    }
    const int32_t graph_start = k3_fst_start(fst);
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
    k3_decoder *dec = nullptr; K3H_CHECK_K3(k3_decoder_create(fst, &dc, max_batch, ninfo.output_dim, &dec));
    k3_decoder *dec_b = nullptr; if (alternate_decoders) K3H_CHECK_K3(k3_decoder_create(fst, &dc, max_batch, ninfo.output_dim, &dec_b));
    k3_decoder *decs2[2] = {dec, alternate_decoders ? dec_b : dec};

    auto scp = ReadScp(wav_rspec);
    if (num_todo >= 0 && (size_t)num_todo < scp.size()) scp.resize(num_todo);
    // static round-robin shard
    if (world_size > 1) {
      decltype(scp) mine;
      for (size_t i = 0; i < scp.size(); i++) if ((int32_t)(i % (size_t)world_size) == rank) mine.push_back(scp[i]);
      scp.swap(mine);
    }
    // --segmentation (BatchedThreadedNnet3CudaPipeline2::SegmentedDecodeWithCallback, batched-threaded-nnet3-cuda-pipeline2.cc:265-337): every file is cut into segments of
    // --segment-length seconds that overlap by --segment-overlap, a last piece shorter than --min-segment-length is dropped, every segment is decoded as an utterance of its own
    // and written under the key [utt]-[offset in whole seconds] (WriteLattices with print_offsets, cuda-pipeline-common.cc:38-62).  One pass over the files for their lengths.
    struct Segment { int64_t off = 0, len = -1; };
    std::vector<Segment> segs(scp.size());
    if (segmentation) {
      if (min_segment_length_s < 0.5) K3H_ERR << "Min segment length must be at least 0.5 second";
      if (segment_overlap_s > segment_length_s) K3H_ERR << "The segments overlap cannot be larger than segment length";
      if (segment_length_s < min_segment_length_s) K3H_ERR << "Segment length cannot be smaller than min segment length";
      if (segment_overlap_s >= segment_length_s) K3H_ERR << "The segments overlap must be smaller than the segment length";
      const int seg_len = (int)(segment_length_s * fopts.samp_freq), seg_shift = (int)((segment_length_s - segment_overlap_s) * fopts.samp_freq),
          seg_min = (int)(min_segment_length_s * fopts.samp_freq);
      decltype(scp) cut; std::vector<Segment> cut_segs;
      for (size_t i = 0; i < scp.size(); i++) {
        int64_t total = -1;
        try { const Wave w = ReadWave(scp[i].second); if (w.samp_freq == fopts.samp_freq) total = (int64_t)w.samples.size(); } catch (const std::exception &) {}
        if (total < 0) { cut.push_back(scp[i]); cut_segs.push_back(Segment()); continue; }      // unreadable / wrong rate: the batch loader reports it, once
        for (int64_t offset = 0; total > 0; offset += seg_shift) {
          const int64_t n = std::min<int64_t>(total - offset, seg_len);
          if (n >= seg_min) {
            std::ostringstream key; key << scp[i].first << "-" << (double)std::floor((float)offset / fopts.samp_freq);
            cut.push_back({key.str(), scp[i].second}); Segment sg; sg.off = offset; sg.len = n; cut_segs.push_back(sg);
          }
          if (offset + n >= total) break;
        }
      }
      scp.swap(cut); segs.swap(cut_segs);
    }
    // OpenOutputHandles (cudadecoderbin/cuda-bin-tools.h:181-195): an output argument that is not a table wspecifier names a .ctm file
    const bool ctm_mode = !(out_wspec.compare(0, 3, "ark") == 0 || out_wspec.compare(0, 3, "scp") == 0) || out_wspec.find(':') == std::string::npos;
    std::shared_ptr<LatticePostprocessor> postprocessor; std::vector<std::string> syms;
    if (postproc.empty()) { if (ctm_mode) K3H_ERR << "You must configure the lattice postprocessor with --lattice-postprocessor-rxfilename to use CTM output"; }
    // (the model: --word-boundary-rxfilename aligns the lattice on word boundaries first)
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
    // determinization runs on worker threads while the GPU works on the next batch; records come out in submission order
    std::unique_ptr<DeterminizeSequencer> det_pool;
    if ((writer && (determinize || postprocessor)) || ctm_mode) {
      DeterminizeSequencer::Config pc; pc.num_threads = worker_threads > 0 ? worker_threads : cores_of_rank;
      pc.beam = lattice_beam;
      pc.trans = &ti;
      pc.phone_det = det_opts;
      pc.determinize = determinize;
      pc.postprocessor = postprocessor;
      pc.ctm_out = ctm_file.get();
      pc.word_syms = syms.empty() ? nullptr : &syms;
      det_pool.reset(new DeterminizeSequencer(pc, writer.get()));
    }
    int num_task = 0, num_err = 0; double total_audio = 0.0;

    // Three overlapped stages, one batch apart (the reference overlaps the same work with its thread pool and copy threads):
    //   reader  (background): ReadWave on --cuda-decoder-copy-threads threads into one of two pinned staging buffers
    //   GPU     (this thread): async H2D from the pinned buffer, features, network, decoder, raw lattices back to the host
    //   post    (background): per-utterance Lattice objects, fst::Connect, then the determinization pool or the raw writer
    struct Batch { std::vector<std::string> keys; std::vector<int64_t> woff, foff; std::vector<int32_t> nframes; int slot = 0, iter = 0, num_err = 0; double audio = 0.0; };
    struct Pinned { float *p = nullptr; size_t cap = 0; };
    Pinned pinned[2];
    const int n_read_threads = std::max(1, copy_threads);
    auto load_batch = [&](size_t b0, size_t b1, int slot, int iter) {
      Batch b; b.slot = slot; b.iter = iter; b.woff.assign(1, 0); b.foff.assign(1, 0);
      std::vector<Wave> waves(b1 - b0); std::vector<char> bad(b1 - b0, 0);
      auto parallel = [&](const std::function<void(size_t)> &fn) {
        std::vector<std::thread> th;
        for (int t = 0; t < n_read_threads; t++) th.emplace_back([&, t] { for (size_t i = t; i < waves.size(); i += n_read_threads) fn(i); });
        for (auto &x : th) x.join();
      };
      parallel([&](size_t i) {
        try {
          waves[i] = ReadWave(scp[b0 + i].second); const Segment &sg = segs[b0 + i];
          if (sg.len >= 0) { std::vector<float> piece(waves[i].samples.begin() + sg.off, waves[i].samples.begin() + sg.off + sg.len); waves[i].samples.swap(piece); }
        } catch (const std::exception &) { bad[i] = 1; }
      });
      std::vector<size_t> src;
      for (size_t i = 0; i < waves.size(); i++) {
        const Wave &w = waves[i];
        if (bad[i]) { b.num_err++; continue; }
        if (w.samp_freq != fopts.samp_freq) { K3H_WARN << "Sample frequency mismatch for " << scp[b0 + i].first; b.num_err++; continue; }
        const int nf = k3_feat_num_frames(plan, (int64_t)w.samples.size());
        if (nf == 0) { K3H_WARN << "Utterance " << scp[b0 + i].first << " is too short to decode"; b.num_err++; continue; }
        b.keys.push_back(scp[b0 + i].first);
        src.push_back(i);
        b.woff.push_back(b.woff.back() + (int64_t)w.samples.size());
        b.foff.push_back(b.foff.back() + nf);
        b.nframes.push_back(nf);
        b.audio += w.samples.size() / (double)w.samp_freq;
      }
      Pinned &pb = pinned[slot]; const size_t need = (size_t)b.woff.back();
      if (need > pb.cap) {
        if (pb.p) HIPCHK(hipHostFree(pb.p));
        pb.cap = need + need / 4 + 1024;
        HIPCHK(hipHostMalloc((void **)&pb.p, pb.cap * sizeof(float), hipHostMallocDefault));
      }
      std::vector<std::thread> th;
      for (int t = 0; t < n_read_threads; t++) th.emplace_back([&,
          t] { for (size_t k = t; k < src.size(); k += n_read_threads) memcpy(pb.p + b.woff[k], waves[src[k]].samples.data(),
          waves[src[k]].samples.size() * sizeof(float)); });
      for (auto &x : th) x.join();
      return b;
    };
    struct Raw { std::vector<std::string> keys; std::vector<int64_t> info; std::vector<int32_t> sf, ss, as, ad, ai, ao; std::vector<float> sc, sfin, ag, aa; };
    int post_err = 0; double tot_like = 0.0; int64_t like_frames = 0;      // (v1) GetDiagnosticsAndPrintOutput: likelihood of the best path, per frame
    auto post_process = [&](std::shared_ptr<Raw> r) {
      int64_t s0 = 0, a0 = 0; const int U = (int)r->keys.size();
      for (int u = 0; u < U; u++) {
        const int64_t ns = r->info[10 * u], na = r->info[10 * u + 1]; const std::string &key = r->keys[u];
        if (r->info[10 * u + 2] != 0 || ns == 0) { K3H_WARN << "Failed to decode utterance with id " << key; post_err++; s0 += ns; a0 += na; continue; }
        if (!r->info[10 * u + 3]) K3H_WARN << "Outputting partial output for utterance " << key << " since no final-state reached";
        Lattice lat;
        lat.st_frame.assign(r->sf.begin() + s0, r->sf.begin() + s0 + ns);
        lat.st_state.assign(r->ss.begin() + s0, r->ss.begin() + s0 + ns);
        lat.st_final.assign(r->sfin.begin() + s0, r->sfin.begin() + s0 + ns);
        lat.arc_src.assign(r->as.begin() + a0, r->as.begin() + a0 + na);
        lat.arc_dst.assign(r->ad.begin() + a0, r->ad.begin() + a0 + na);
        lat.arc_ilabel.assign(r->ai.begin() + a0, r->ai.begin() + a0 + na);
        lat.arc_olabel.assign(r->ao.begin() + a0, r->ao.begin() + a0 + na);
        lat.arc_graph.assign(r->ag.begin() + a0, r->ag.begin() + a0 + na);
        lat.arc_ac.assign(r->aa.begin() + a0, r->aa.begin() + a0 + na);
        for (int64_t s = 0; s < ns; s++) if (lat.st_frame[s] == 0 && lat.st_state[s] == graph_start) lat.start = (int32_t)s;
        Connect(&lat);
        if (v1) {
          std::vector<int32_t> ali, words;
          double gc = 0.0, ac = 0.0;
          if (BestPath(lat, &ali, &words, &gc, &ac)) {
            tot_like += -(gc + ac);
            like_frames += (int64_t)ali.size();
          } else K3H_WARN << "Empty lattice.";
        }
        if (det_pool) det_pool->Run(key, std::move(lat));
        else if (write_compact) { CompactLattice clat; ConvertLattice(lat, &clat); writer->WriteCompactLattice(key, clat); }
        else writer->WriteLattice(key, lat);
        s0 += ns; a0 += na;
      }
    };
    // the list of (iteration, first file, last file) batches
    std::vector<std::array<size_t, 3>> plan_batches;
    for (int iter = 0; iter < iterations; iter++) for (size_t b0 = 0; b0 < scp.size(); b0 +=
        max_batch) plan_batches.push_back({(size_t)iter, b0, std::min(scp.size(), b0 + (size_t)max_batch)});
    // GPU stage, two streams one batch apart: the FRONT END of batch k+1 (upload, features, i-vectors, network) is issued on its own stream right behind the
    // decoder kernels of batch k
    // (log-likelihoods double-buffered), so that the copy and the first network layers run while the decoder's last lanes finish -- as bench.py does.
    DevBuf<float> d_w, d_f, d_ll[2], d_iv; DevBuf<int64_t> d_wo, d_fo; PinnedBuf<int64_t> h_off[2];
    std::vector<std::pair<std::vector<int32_t>, k3_nnet_batch *>> plan_cache;
    std::future<Batch> next; std::future<void> post;
    hipStream_t s_front, s_dec, s_dec_b;
    HIPCHK(hipStreamCreateWithFlags(&s_front, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&s_dec, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&s_dec_b, hipStreamNonBlocking));
    hipStream_t s_decs[2] = {s_dec, alternate_decoders ? s_dec_b : s_dec};
    // decoder of the batch that last read log-likelihood buffer k & 1 is through
    hipEvent_t ev_dec[2];
    for (auto &e : ev_dec) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t ev_front[2]; for (auto &e : ev_front) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // the copy out of staging slot 0 / 1 is through
    hipEvent_t ev_h2d[2];
    for (auto &e : ev_h2d) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    bool h2d_recorded[2] = {false, false};
    struct Front { Batch b; std::vector<int64_t> ro; bool valid = false; double wait_ms = 0.0; } fr[2];
    auto tick = [] {
      return std::chrono::steady_clock::now();
    };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_start = std::chrono::steady_clock::now();
    if (!plan_batches.empty()) next = std::async(std::launch::async, load_batch, plan_batches[0][1], plan_batches[0][2], 0, (int)plan_batches[0][0]);
    // everything of batch k up to its log-likelihoods, on s_front; the shared device buffers and the network workspace are free: front end k-1 has completed
    // (caller)
    auto front_end = [&](size_t k) {
      Front &f = fr[k & 1]; f.valid = false;
      const auto t_a = tick();
      f.b = next.get(); f.wait_ms = ms(t_a, tick());
      if (k + 1 < plan_batches.size()) {
        // the reader fills the staging slot batch k - 1 was copied from: that copy must be through
        if (h2d_recorded[(k + 1) & 1]) HIPCHK(hipEventSynchronize(ev_h2d[(k + 1) & 1]));
        next = std::async(std::launch::async, load_batch, plan_batches[k + 1][1], plan_batches[k + 1][2], (int)((k + 1) & 1), (int)plan_batches[k + 1][0]);
      }
      Batch &b = f.b;
      num_err += b.num_err; if (b.iter == 0) { total_audio += b.audio; num_task += (int)b.keys.size(); }      // per iteration, like the reference's counters
      if (b.keys.empty()) return;
      const int U = (int)b.keys.size(); const int64_t tot = b.foff.back(), nsamp = b.woff.back();
      // (alternating decoders: batch k - 2 may still be decoding from the log-likelihood buffer this front end writes.  The buffer's last reader is that decoder object's token-passing
      // launch; the pruning / output kernels behind it cannot start beside the other object's resident launch and end tens of milliseconds into it)
      if (alternate_decoders) K3H_CHECK_K3(k3_decoder_stream_wait_token_passing(decs2[k & 1], s_front)); else HIPCHK(hipStreamWaitEvent(s_front, ev_dec[k & 1], 0));
      HIPCHK(hipMemcpyAsync(d_w.need((size_t)nsamp), pinned[b.slot].p, (size_t)nsamp * sizeof(float), hipMemcpyHostToDevice, s_front));
      // the offsets go through the stream as well (a synchronous copy on the null stream would overtake a previous front end that is still queued: this one is
      // issued without waiting for it)
      {      // (from page-locked memory: an asynchronous copy out of pageable memory may wait for the stream's earlier work on the host)
        // (this slot's previous copy is through: ev_h2d was waited for before the reader refilled the slot)
        int64_t *ho = h_off[b.slot & 1].need(b.woff.size() + b.foff.size());
        memcpy(ho, b.woff.data(), b.woff.size() * sizeof(int64_t)); memcpy(ho + b.woff.size(), b.foff.data(), b.foff.size() * sizeof(int64_t));
        HIPCHK(hipMemcpyAsync(d_wo.need(b.woff.size()), ho, b.woff.size() * sizeof(int64_t), hipMemcpyHostToDevice, s_front));
        HIPCHK(hipMemcpyAsync(d_fo.need(b.foff.size()), ho + b.woff.size(), b.foff.size() * sizeof(int64_t), hipMemcpyHostToDevice, s_front));
      }
      HIPCHK(hipEventRecord(ev_h2d[b.slot & 1], s_front)); h2d_recorded[b.slot & 1] = true;
      K3H_CHECK_K3(k3_feat_compute_batch(plan, d_w.p, d_wo.p, d_fo.p, U, tot, d_f.need((size_t)tot * fdim), fdim, s_front));
      // the network plan of a batch (row bookkeeping, tile tables, activation workspace in HBM) depends only on the utterances' frame counts: kept for
      // the batches that come back (--iterations, equal-length test sets) instead of being rebuilt per batch
      k3_nnet_batch *nb = nullptr; f.ro.assign(U + 1, 0);
      for (auto &c : plan_cache) if (c.first == b.nframes) { nb = c.second; break; }
      const bool cached = nb != nullptr;
      DevBuf<float> &ll = d_ll[k & 1];
      // (growing the buffer frees it: only once the decoder that read it last is through)
      auto ll_need = [&](size_t n) {
        if (n > ll.cap) HIPCHK(hipEventSynchronize(ev_dec[k & 1]));
        return ll.need(n);
      };
      if (!ivx) {
        if (!cached) K3H_CHECK_K3(k3_nnet_batch_create(nnet, U, b.nframes.data(), subsampling, log_priors.empty() ? nullptr : log_priors.data(), acoustic_scale, &nb));
        const int64_t rows = k3_nnet_batch_output_rows(nb, f.ro.data());
        K3H_CHECK_K3(k3_nnet_forward(nb, d_f.p, fdim, ll_need((size_t)rows * ninfo.output_dim), ninfo.output_dim, s_front));
      } else {
        std::vector<int32_t> iv_rows(U); for (int u = 0; u < U; u++) iv_rows[u] = (b.nframes[u] + iv_period - 1) / iv_period;
        const int64_t n_iv = k3_ivector_num_rows(ivx, U, b.foff.data(), nullptr);
        K3H_CHECK_K3(k3_ivector_extract_batch(ivx, d_f.p, fdim, b.foff.data(), U, d_iv.need((size_t)n_iv * ninfo.ivector_dim), ninfo.ivector_dim, s_front));
        if (!cached) K3H_CHECK_K3(k3_nnet_batch_create_ivector(nnet, U, b.nframes.data(), subsampling, log_priors.empty() ? nullptr : log_priors.data(),
            acoustic_scale, frames_per_chunk, iv_period, iv_rows.data(), &nb));
        const int64_t rows = k3_nnet_batch_output_rows(nb, f.ro.data());
        K3H_CHECK_K3(k3_nnet_forward_ivector(nb, d_f.p, fdim, d_iv.p, ninfo.ivector_dim, ll_need((size_t)rows * ninfo.output_dim), ninfo.output_dim, s_front));
      }
      if (!cached) {      // (an evicted plan's workspace is not in use: only this front end's network is in flight and it uses `nb`)
        plan_cache.push_back({b.nframes, nb});
        if (plan_cache.size() > 4) { HIPCHK(hipStreamSynchronize(s_front)); k3_nnet_batch_destroy(plan_cache.front().second); plan_cache.erase(plan_cache.begin()); }
      }
      HIPCHK(hipEventRecord(ev_front[k & 1], s_front)); f.valid = true;
    };
    if (!plan_batches.empty()) front_end(0);
    // lattices of a decoded batch: sizes, the ten arrays, hand-over to the post stage
    auto fetch = [&](k3_decoder *d, Batch &b, size_t k, double waited, std::chrono::steady_clock::time_point t_b) {
      const int U = (int)b.keys.size();
      auto r = std::make_shared<Raw>(); r->info.resize(10 * (size_t)U);
      K3H_LATTICE_INFO(d, r->info.data());
      const auto t_c = tick();
      if ((b.iter == 0 || v1) && (writer || ctm_mode)) {
        int64_t NS = 0, NA = 0; for (int u = 0; u < U; u++) { NS += r->info[10 * u]; NA += r->info[10 * u + 1]; }
        r->sf.resize(NS + 1); r->ss.resize(NS + 1); r->sc.resize(NS + 1); r->sfin.resize(NS + 1);
        r->as.resize(NA + 1); r->ad.resize(NA + 1); r->ai.resize(NA + 1); r->ao.resize(NA + 1); r->ag.resize(NA + 1); r->aa.resize(NA + 1);
        K3H_CHECK_K3(k3_decoder_get_raw_lattices(d, r->sf.data(), r->ss.data(), r->sc.data(), r->sfin.data(), r->as.data(), r->ad.data(), r->ai.data(),
            r->ao.data(), r->ag.data(), r->aa.data()));
        r->keys = std::move(b.keys);
        if (v1 && b.iter > 0) for (auto &key : r->keys) key = std::to_string(b.iter) + "-" + key;      // "make key unique for each iteration" (batched-wav-nnet3-cuda.cc:211-214)
        if (post.valid()) post.get();          // keeps the records in order; an error in the previous batch's post stage surfaces here
        post = std::async(std::launch::async, post_process, r);
      } else {
        for (int u = 0; u < U; u++) if (r->info[10 * u + 2] != 0 || r->info[10 * u] == 0) { K3H_WARN << "Failed to decode utterance with id " << b.keys[u]; num_err++; }
      }
      if (v1 && (k + 1 == plan_batches.size() || plan_batches[k + 1][0] != plan_batches[k][0])) {      // the iteration's last batch: its group is complete (:292-299)
        const double tt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); const int it = (int)plan_batches[k][0];
        K3H_LOG << "~Group " << it << " completed Aggregate Total Time: " << tt << " Audio: " << total_audio * (it + 1) << " RealTimeX: " << total_audio * (it + 1) / tt;
      }
      K3H_VLOG(1) << "batch " << k << ": waited " << waited << " ms for the reader, " << ms(t_b, t_c) <<
          " ms decoder of this batch (+ front end of the next one queued behind it), " << ms(t_c, tick()) << " ms lattices to the host + hand-over";
    };
    // alternating decoders: the batch whose lattices are fetched one step later
    struct Held {
      Batch b;
      bool valid = false;
      double waited = 0.0;
      std::chrono::steady_clock::time_point t_b;
      size_t k = 0;
    }
    held;
    for (size_t k = 0; k < plan_batches.size(); k++) {
      const auto t_b = tick();
      Front &f = fr[k & 1];
      if (f.valid) {
        HIPCHK(hipStreamWaitEvent(s_decs[k & 1], ev_front[k & 1], 0));
        K3H_CHECK_K3(k3_decoder_decode_batch(decs2[k & 1], (int)f.b.keys.size(), d_ll[k & 1].p, ninfo.output_dim, f.ro.data(), s_decs[k & 1]));
        HIPCHK(hipEventRecord(ev_dec[k & 1], s_decs[k & 1]));
        // (no host wait for front end k here: batch k+1's front end is queued behind it on the same stream, its staging slot is guarded by ev_h2d; with an i-vector extractor the
        // extraction call reads host-side offsets of the batch, so that configuration keeps the wait)
        if (ivx) HIPCHK(hipEventSynchronize(ev_front[k & 1]));
      }
      // batch k+1's front end is queued NOW, behind the decoder kernels of batch k; then this thread waits for a decoder: batch k's, or -- with two decoder objects -- batch k-1's,
      // whose pruning kernel, compaction and copy run while batch k's token passing has already started on the other object's stream
      const bool had = f.valid; Batch bk; const double waited = f.wait_ms;
      if (had) bk = std::move(f.b);
      if (k + 1 < plan_batches.size()) front_end(k + 1);
      if (alternate_decoders) {
        if (held.valid) { fetch(decs2[held.k & 1], held.b, held.k, held.waited, held.t_b); held.valid = false; }
        if (had) { held.b = std::move(bk); held.valid = true; held.waited = waited; held.t_b = t_b; held.k = k; }
      } else if (had) fetch(dec, bk, k, waited, t_b);
    }
    if (held.valid) fetch(decs2[held.k & 1], held.b, held.k, held.waited, held.t_b);
    if (post.valid()) post.get();
    num_err += post_err;
    HIPCHK(hipDeviceSynchronize());
    for (auto &c : plan_cache) k3_nnet_batch_destroy(c.second);
    for (auto &e : ev_front) (void)hipEventDestroy(e);
    for (auto &e : ev_dec) (void)hipEventDestroy(e);
    for (auto &e : ev_h2d) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(s_front);
    (void)hipStreamDestroy(s_dec);
    (void)hipStreamDestroy(s_dec_b);
    { const auto t_w = std::chrono::steady_clock::now(); if (det_pool) { det_pool->Wait(); det_pool.reset(); }
      K3H_VLOG(1) << "waited " << std::chrono::duration<double,
          std::milli>(std::chrono::steady_clock::now() - t_w).count() << " ms for the determinization pool after the last batch";
      }
    if (writer) writer->Flush();
    const double total_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    K3H_LOG << "Decoded " << num_task << " utterances, " << num_err << " with errors.";
    if (v1) K3H_LOG << "Overall likelihood per frame was " << (like_frames ? tot_like / like_frames : 0.0) << " per frame over " << like_frames << " frames.";
    K3H_LOG << "Overall: " << " Aggregate Total Time: " << total_time << " Total Audio: " << total_audio * iterations << " RealTimeX: " << total_audio * iterations / total_time;
    k3_decoder_destroy(dec); if (dec_b) k3_decoder_destroy(dec_b); k3_fst_destroy(fst); k3_nnet_destroy(nnet); k3_feat_plan_destroy(plan); if (ivx) k3_ivector_destroy(ivx);
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
