// lattice-mbr-decode -- same command line as the reference's latbin/lattice-mbr-decode.cc:28-120: word-level Minimum Bayes Risk decoding of every lattice of a table
// (Lattice or CompactLattice records, text or binary): the MBR word sequence, the Bayes risk, the sausage statistics and the times (k3_mbr.cc = lat/sausages.cc).
// The extra outputs are written in Kaldi's TEXT layouts (BaseFloatWriter / PosteriorWriter / BaseFloatPairVectorWriter with ",t"); host-only.
#include <cmath>
#include <fstream>
#include <iostream>
#include "k3_host.h"
using namespace k3host;
int main(int argc, char **argv) {
  try {
    g_program = "lattice-mbr-decode";
    const char *usage =
        "Do Minimum Bayes Risk decoding (decoding that aims to minimize the expected word error rate).\n"
        "Usage: lattice-mbr-decode [options]  lattice-rspecifier transcriptions-wspecifier [ bayes-risk-wspecifier [ sausage-stats-wspecifier [ times-wspecifier] ] ]\n"
        " e.g.: lattice-mbr-decode --acoustic-scale=0.1 ark:1.lats 'ark,t:text.int' ark:/dev/null ark,t:1.sau\n";
    ParseOptions po(usage);
    float acoustic_scale = 1.0f, lm_scale = 1.0f; bool one_best_times = false; std::string word_syms; MinimumBayesRiskOptions mbr_opts;
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods");
    po.Register("lm-scale", &lm_scale, "Scaling factor for language model probabilities");
    po.Register("word-symbol-table", &word_syms, "Symbol table for words [for debug output]");
    po.Register("one-best-times", &one_best_times, "If true, output times corresponding to one-best, not whole sausage.");
    po.Register("decode-mbr", &mbr_opts.decode_mbr, "(lattice-to-ctm-conf's option) If true, do Minimum Bayes Risk decoding (else, Maximum a Posteriori)");
    po.Register("print-silence", &mbr_opts.print_silence, "(lattice-to-ctm-conf's option) Keep the inter-word '<eps>' bins in the 1-best output");
    po.Read(argc, argv);
    if (po.NumArgs() < 2 || po.NumArgs() > 5) { po.PrintUsage(); return 1; }
    auto text_out = [&](int k) -> std::unique_ptr<std::ofstream> {      // "ark,t:file" / "ark:file" -> the file (text either way); "" or /dev/null -> none
      if (po.NumArgs() < k || po.GetArg(k).empty()) return nullptr;
      const std::string w = po.GetArg(k); const size_t c = w.find(':'); const std::string path = c == std::string::npos ? w : w.substr(c + 1);
      if (path.empty() || path == "/dev/null") return nullptr;
      std::unique_ptr<std::ofstream> f(new std::ofstream(path)); if (!*f) K3H_ERR << "cannot open " << path; f->precision(7); return f;
    };
    std::unique_ptr<TableWriter> trans; if (!po.GetArg(2).empty()) trans.reset(new TableWriter(po.GetArg(2)));
    auto risk_out = text_out(3), saus_out = text_out(4), times_out = text_out(5);
    int32_t n_done = 0; int64_t n_words = 0; double tot_risk = 0.0;
    for (auto &kv : ReadLatticeTable(po.GetArg(1))) {
      Lattice &lat = kv.second;
      for (float &g : lat.arc_graph) g = (float)((double)lm_scale * g);            // fst::ScaleLattice(fst::LatticeScale(lm_scale, acoustic_scale), &clat)
      for (float &f : lat.st_final) if (std::isfinite(f)) f = (float)((double)lm_scale * f);
      ScaleAcoustic(&lat, acoustic_scale);
      Connect(&lat);      // (a table of raw decoder lattices may hold states off every complete path)
      CompactLattice clat; if (lat.NumStates() > 0) ConvertLattice(lat, &clat);
      if (clat.NumStates() == 0) { K3H_WARN << "Empty lattice for key " << kv.first; continue; }
      MinimumBayesRisk mbr(clat, mbr_opts);
      if (trans) trans->WriteInt32Vector(kv.first, mbr.GetOneBest());
      if (risk_out) *risk_out << kv.first << " " << mbr.GetBayesRisk() << "\n";
      if (saus_out) {
        *saus_out << kv.first;
        for (const auto &bin : mbr.GetSausageStats()) {
          *saus_out << " [";
          for (const auto &e : bin) *saus_out << " " << e.first << " " << e.second;
          *saus_out << " ]";
        }
        *saus_out << "\n";
      }
      if (times_out) {
        *times_out << kv.first;
        const auto &t = one_best_times ? mbr.GetOneBestTimes() : mbr.GetSausageTimes();
        for (size_t i = 0; i < t.size(); i++) *times_out << " " << t[i].first << " " << t[i].second << (i + 1 < t.size() ? " ;" : "");
        *times_out << "\n";
      }
      n_done++; n_words += (int64_t)mbr.GetOneBest().size(); tot_risk += mbr.GetBayesRisk();
    }
    if (trans) trans->Flush();
    K3H_LOG << "Done " << n_done << " lattices.";
    K3H_LOG << "Average Bayes Risk per sentence is " << (tot_risk / std::max(1, n_done)) << " and per word, " << (tot_risk / std::max<int64_t>(1, n_words));
    return n_done != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
