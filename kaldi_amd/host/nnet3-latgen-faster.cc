// nnet3-latgen-faster -- drop-in for nnet3bin/nnet3-latgen-faster.cc:33-260 with the forward pass and the decoder on MI355X:
//   nnet3-latgen-faster [options] <nnet-in> <fst-in> <features-rspecifier> <lattice-wspecifier> [ <words-wspecifier> [<alignments-wspecifier>] ]
// = DecodableAmNnetSimple + LatticeFasterDecoder + DecodeUtteranceLatticeFaster (decoder/decoder-wrappers.cc:287-382) per utterance in
// the reference; here all utterances of a batch run through k3_nnet_forward and k3_decoder_decode_batch.  Like the reference it
// determinizes by default and writes CompactLattices (decoder-wrappers.cc:354-368); the determinizer is the host-side restatement
// of DeterminizeLatticePhonePrunedWrapper in k3_lattice.cc.
// --determinize-lattice=false writes the raw state-level lattice.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <iostream>
#include <thread>
#include "k3_host.h"
#include "k3_nnet_ivector_cli.h"
#include "../../include/k3hip.h"
using namespace k3host;
#define HIPCHK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) K3H_ERR << "HIP error " << hipGetErrorName(e__) << " in " << #e; } while (0)
int main(int argc, char **argv) {
  try {
    const char *usage = "Generate lattices using nnet3 neural net model.\n"
                        "Usage: nnet3-latgen-faster [options] <nnet-in> <fst-in> <features-rspecifier> <lattice-wspecifier> [ <words-wspecifier> [<alignments-wspecifier>] ]\n"
                        "See also: nnet3-latgen-faster-parallel, nnet3-latgen-faster-batch\n";
    ParseOptions po(usage);
    bool allow_partial = false, determinize = true, debug_comp = false, literal_order = true, phone_det = true, word_det = true, minimize = false;
    int32_t det_threads = 0;
    std::string word_syms, use_gpu = "yes", ivector_rspecifier, online_ivector_rspecifier, utt2spk;
    int32_t subsampling = 1, frames_per_chunk = 50, elc = 0, erc = 0, elci = -1, ercf = -1, online_ivector_period = 0, max_batch = 256,
        max_active = 2147483647, min_active = 200, prune_interval = 25, max_mem = 50000000, frame_tokens_cap = 65536, lane_tokens_cap = 4000000,
        lane_links_cap = 8000000;
    float beam = 16.0f, lattice_beam = 10.0f, acoustic_scale = 0.1f, beam_delta = 0.5f, hash_ratio = 2.0f, prune_scale = 0.1f, delta = 0.000976562f;
    po.Register("word-symbol-table", &word_syms, "Symbol table for words [for debug output] (accepted, unused)");
    po.Register("allow-partial", &allow_partial, "If true, produce output even if end state was not reached.");
    po.Register("beam", &beam, "Decoding beam.  Larger->slower, more accurate.");
    po.Register("max-active", &max_active, "Decoder max active states.  Larger->slower; more accurate");
    po.Register("min-active", &min_active, "Decoder minimum #active states.");
    po.Register("lattice-beam", &lattice_beam, "Lattice generation beam.  Larger->slower, and deeper lattices");
    po.Register("prune-interval", &prune_interval, "(accepted; pruning runs once after the last frame and gives the same lattice)");
    po.Register("determinize-threads", &det_threads,
        "Host threads that determinize lattices while the GPU decodes the next batch (0: all cores; an addition -- the reference determinizes inline)");
    po.Register("determinize-lattice", &determinize,
        "If true, determinize the lattice (lattice-determinization, keeping only best pdf-sequence for each word-sequence).");
    po.Register("beam-delta", &beam_delta, "Increment used in decoding-- this parameter is obscure and relates to a speedup in the way the max-active constraint is applied.");
    po.Register("hash-ratio", &hash_ratio,
        "Setting used in decoder to control hash behavior (it decides the token visit order, hence which tokens the running cutoff keeps; honoured with --literal-order)");
    po.Register("literal-order", &literal_order,
        "(not in the reference) true = raw lattices identical to the CPU LatticeFasterDecoder's, bit for bit; false = the order-independent fast decoder");
    po.Register("prune-scale", &prune_scale, "(accepted, unused)");
    po.Register("max-mem", &max_mem, "Maximum approximate memory usage in determinization (real usage might be many times this).");
    po.Register("phone-determinize", &phone_det, "If true, do an initial pass of determinization on both phones and words (see also --word-determinize)");
    po.Register("word-determinize", &word_det, "If true, do a second pass of determinization on words only (see also --phone-determinize)");
    po.Register("minimize", &minimize, "If true, push and minimize after determinization."); po.Register("delta", &delta, "Tolerance used in determinization");
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic log-likelihoods");
    po.Register("frame-subsampling-factor", &subsampling,
        "Required if the frame-rate of the output (e.g. in 'chain' models) is less than the frame-rate of the original alignment.");
    po.Register("frames-per-chunk", &frames_per_chunk,
        "Number of frames in each chunk that is separately evaluated by the neural net (matters with --online-ivectors: one i-vector per chunk)");
    po.Register("extra-left-context", &elc, "(only 0 is supported)");
    po.Register("extra-right-context", &erc, "(only 0 is supported)");
    po.Register("extra-left-context-initial", &elci, "(accepted)");
    po.Register("extra-right-context-final", &ercf, "(accepted)");
    po.Register("debug-computation", &debug_comp, "(accepted, unused)");
    po.Register("ivectors", &ivector_rspecifier,
        "Rspecifier for iVectors as vectors (i.e. not estimated online); per utterance by default, or per speaker if you provide the --utt2spk option.");
    po.Register("online-ivectors", &online_ivector_rspecifier,
        "Rspecifier for iVectors estimated online, as matrices.  If you supply this, you must set the --online-ivector-period option.");
    po.Register("online-ivector-period", &online_ivector_period, "Number of frames between iVectors in matrices supplied to the --online-ivectors option");
    po.Register("utt2spk", &utt2spk, "Rspecifier for utt2spk option used to get ivectors per speaker");
    po.Register("frame-tokens-cap", &frame_tokens_cap, "Decoder capacity: tokens alive on one frame of one utterance");
    po.Register("lane-tokens-cap", &lane_tokens_cap,
        "Decoder capacity: tokens of all frames of one utterance (an utterance that exceeds it is reported as failed)");
    po.Register("lane-links-cap", &lane_links_cap, "Decoder capacity: forward links of all frames of one utterance");
    po.Register("use-gpu", &use_gpu, "(this build always uses the GPU)"); po.Register("max-batch-size", &max_batch, "Utterances per GPU batch");
    po.Read(argc, argv);
    if (po.NumArgs() < 4 || po.NumArgs() > 6) { po.PrintUsage(); return 1; }
    DeterminizeLatticePhonePrunedOptions det_opts;
    det_opts.delta = delta;
    det_opts.max_mem = max_mem;
    det_opts.phone_determinize = phone_det;
    det_opts.word_determinize = word_det;
    det_opts.minimize = minimize;
    if (elc || erc) K3H_ERR << "extra context is not supported by this program (feed-forward TDNN / TDNN-F models do not use it)";
    IvectorInputs iv; iv.Open(ivector_rspecifier, online_ivector_rspecifier, utt2spk, online_ivector_period);
    const std::string model_rx = po.GetArg(1), fst_rx = po.GetArg(2);
    if (fst_rx.find(':') != std::string::npos && fst_rx.compare(0, 3, "ark") == 0) K3H_ERR << "a table of per-utterance FSTs is not supported; give one HCLG";
    TransitionInfo ti = ReadTransitionModel(model_rx);
    k3_nnet *nnet = nullptr; K3H_CHECK_K3(k3_nnet_load(model_rx.c_str(), &nnet));
    k3_nnet_info ni; K3H_CHECK_K3(k3_nnet_get_info(nnet, &ni));
    if (ni.output_dim != ti.num_pdfs) K3H_ERR << "Model output dimension " << ni.output_dim << " != number of pdfs in the transition model " << ti.num_pdfs;
    std::vector<float> log_priors;
    if (ni.has_priors) { log_priors.resize(ni.output_dim); K3H_CHECK_K3(k3_nnet_get_priors(nnet, log_priors.data())); for (float &p : log_priors) p = logf(p); }
    HostFst hfst = ReadFstKaldiGeneric(fst_rx);
    k3_fst *fst = nullptr;
    K3H_CHECK_K3(k3_fst_create(hfst.NumStates(), hfst.start, hfst.arc_offsets.data(), hfst.ilabel.data(), hfst.olabel.data(), hfst.weight.data(),
        hfst.nextstate.data(), hfst.final_cost.data(),
                               ti.id2pdf.data(), (int32_t)ti.id2pdf.size(), &fst));
    k3_decoder_config dc; k3_decoder_config_default(&dc);
    dc.beam = beam; dc.lattice_beam = lattice_beam; dc.max_active = max_active; dc.min_active = std::min(min_active, max_active - 1); dc.beam_delta = beam_delta;
    dc.frame_tokens_cap = frame_tokens_cap; dc.frame_cands_cap = 4 * frame_tokens_cap; dc.lane_tokens_cap = lane_tokens_cap; dc.lane_links_cap = lane_links_cap;
    dc.literal_order = literal_order ? 1 : 0; dc.hash_ratio = hash_ratio; if (literal_order) dc.frame_tokens_cap = std::min(dc.frame_tokens_cap, 65536);
    k3_decoder *dec = nullptr; K3H_CHECK_K3(k3_decoder_create(fst, &dc, max_batch, ni.output_dim, &dec));
    auto feats = ReadMatrixTable(po.GetArg(3)); TableWriter lat_writer(po.GetArg(4));
    // determinization on a pool of host threads, records written in submission order (as batched-wav-nnet3-cuda2 does): inline it took longer per batch than
    // the GPU by two orders of magnitude
    std::unique_ptr<DeterminizeSequencer> det_pool;
    if (determinize) {
      DeterminizeSequencer::Config pc; pc.num_threads = det_threads > 0 ? det_threads : std::max(1, (int)std::thread::hardware_concurrency());
      pc.beam = lattice_beam; pc.trans = &ti; pc.phone_det = det_opts; pc.post_scale = acoustic_scale != 0.0f ? 1.0 / acoustic_scale : 1.0;
      det_pool.reset(new DeterminizeSequencer(pc, &lat_writer));
    }
    std::unique_ptr<TableWriter> words_writer, ali_writer;
    if (po.NumArgs() >= 5 && !po.GetArg(5).empty()) words_writer.reset(new TableWriter(po.GetArg(5)));
    if (po.NumArgs() >= 6 && !po.GetArg(6).empty()) ali_writer.reset(new TableWriter(po.GetArg(6)));
    int num_success = 0, num_fail = 0; int64_t frame_count = 0; double tot_like = 0.0;
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
      const int U = (int)idx.size();
      k3_nnet_batch *nb = nullptr; std::vector<int64_t> ro; float *d_o = nullptr;
      RunNnetBatch(nnet, ni, nf, all, iv, utt_iv, subsampling, frames_per_chunk, log_priors, acoustic_scale, &nb, &ro, &d_o);
      K3H_CHECK_K3(k3_decoder_decode_batch(dec, U, d_o, ni.output_dim, ro.data(), nullptr));
      std::vector<int64_t> info(10 * (size_t)U); K3H_LATTICE_INFO(dec, info.data());
      int64_t NS = 0, NA = 0; for (int u = 0; u < U; u++) { NS += info[10 * u]; NA += info[10 * u + 1]; }
      std::vector<int32_t> sf(NS + 1), ss(NS + 1), as(NA + 1), ad(NA + 1), ai(NA + 1), ao(NA + 1); std::vector<float> sc(NS + 1), sfin(NS + 1), ag(NA + 1), aa(NA + 1);
      K3H_CHECK_K3(k3_decoder_get_raw_lattices(dec, sf.data(), ss.data(), sc.data(), sfin.data(), as.data(), ad.data(), ai.data(), ao.data(), ag.data(), aa.data()));
      int64_t s0 = 0, a0 = 0;
      for (int u = 0; u < U; u++) {
        const std::string &utt = feats[idx[u]].first; const int64_t ns = info[10 * u], na = info[10 * u + 1];
        const bool ok = info[10 * u + 2] == 0 && ns > 0, reached_final = info[10 * u + 3] != 0;
        if (!ok) { K3H_WARN << "Failed to decode utterance with id " << utt; num_fail++; s0 += ns; a0 += na; continue; }
        if (!reached_final) {
          if (allow_partial) K3H_WARN << "Outputting partial output for utterance " << utt << " since no final-state reached";
          else { K3H_WARN << "Not producing output for utterance " << utt << " since no final-state reached and --allow-partial=false."; num_fail++; s0 += ns; a0 += na; continue; }
        }
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
        s0 += ns; a0 += na;
        std::vector<int32_t> ali, words; double gc = 0, ac = 0;
        if (!BestPath(lat, &ali, &words, &gc, &ac)) { K3H_WARN << "Failed to get traceback for utterance " << utt; num_fail++; continue; }
        if (words_writer) words_writer->WriteInt32Vector(utt, words);
        if (ali_writer) ali_writer->WriteInt32Vector(utt, ali);
        Connect(&lat);
        if (determinize) {
          det_pool->Run(utt, std::move(lat));
        } else {
          if (acoustic_scale != 0.0f) ScaleAcoustic(&lat, 1.0 / acoustic_scale);
          lat_writer.WriteLattice(utt, lat);
        }
        const double like = -(gc + ac); const size_t nfr = ali.size();
        K3H_LOG << "Log-like per frame for utterance " << utt << " is " << (like / std::max<size_t>(nfr, 1)) << " over " << nfr << " frames.";
        tot_like += like; frame_count += (int64_t)nfr; num_success++;
      }
      k3_nnet_batch_destroy(nb); HIPCHK(hipFree(d_o));
    }
    if (det_pool) { det_pool->Wait(); det_pool.reset(); }
    lat_writer.Flush(); if (words_writer) words_writer->Flush(); if (ali_writer) ali_writer->Flush();
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    K3H_LOG << "Time taken " << elapsed << "s: real-time factor assuming 100 frames/sec is " << (elapsed * 100.0 / std::max<int64_t>(frame_count, 1));
    K3H_LOG << "Done " << num_success << " utterances, failed for " << num_fail;
    K3H_LOG << "Overall log-likelihood per frame is " << (tot_like / std::max<int64_t>(frame_count, 1)) << " over " << frame_count << " frames.";
    k3_decoder_destroy(dec); k3_fst_destroy(fst); k3_nnet_destroy(nnet);
    return num_success != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
