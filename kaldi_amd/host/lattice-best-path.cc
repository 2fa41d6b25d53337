// lattice-best-path -- same command line as the reference's latbin/lattice-best-path.cc:30-140: the 1-best word sequence and alignment of
// every lattice of a table (Lattice or CompactLattice records, text or binary), after scaling graph costs by --lm-scale and acoustic
// costs by --acoustic-scale.  Host-only; with it the output of the decoding programs here can be scored without a Kaldi build.
#include <cmath>
#include <iostream>
#include "k3_host.h"
using namespace k3host;
int main(int argc, char **argv) {
  try {
    g_program = "lattice-best-path";
    const char *usage =
        "Generate 1-best path through lattices; output as transcriptions and alignments\n"
        "Usage: lattice-best-path [options]  <lattice-rspecifier> [ <transcriptions-wspecifier> [ <alignments-wspecifier>] ]\n"
        " e.g.: lattice-best-path --acoustic-scale=0.1 ark:1.lats 'ark,t:|int2sym.pl -f 2- words.txt > text' ark:1.ali\n";
    ParseOptions po(usage);
    float acoustic_scale = 1.0f, lm_scale = 1.0f; std::string word_syms;
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods");
    po.Register("lm-scale", &lm_scale, "Scaling factor for LM probabilities. Note: the ratio acoustic-scale/lm-scale is all that matters.");
    po.Register("word-symbol-table", &word_syms, "Symbol table for words [for debug output]");
    po.Read(argc, argv);
    if (po.NumArgs() < 1 || po.NumArgs() > 3) { po.PrintUsage(); return 1; }
    std::map<int32_t, std::string> syms;                  // fst::SymbolTable::ReadText: lines "symbol id"
    if (!word_syms.empty()) {
      std::istringstream in(ReadWholeInput(word_syms)); std::string sym; long id;
      while (in >> sym >> id) syms[(int32_t)id] = sym;
      if (syms.empty()) K3H_ERR << "Could not read symbol table from file " << word_syms;
    }
    std::unique_ptr<TableWriter> words_writer, ali_writer;
    if (po.NumArgs() >= 2 && !po.GetArg(2).empty()) words_writer.reset(new TableWriter(po.GetArg(2)));
    if (po.NumArgs() >= 3 && !po.GetArg(3).empty()) ali_writer.reset(new TableWriter(po.GetArg(3)));
    int32_t n_done = 0, n_fail = 0; int64_t n_frame = 0; double tot_graph = 0.0, tot_ac = 0.0;
    for (auto &kv : ReadLatticeTable(po.GetArg(1))) {
      Lattice &lat = kv.second;
      for (float &g : lat.arc_graph) g = (float)((double)lm_scale * g);            // fst::ScaleLattice(fst::LatticeScale(lm_scale, acoustic_scale), &clat)
      for (float &f : lat.st_final) if (std::isfinite(f)) f = (float)((double)lm_scale * f);
      ScaleAcoustic(&lat, acoustic_scale);
      std::vector<int32_t> ali, words; double g = 0, a = 0;
      if (lat.NumStates() == 0 || !BestPath(lat, &ali, &words, &g, &a)) { K3H_WARN << "Best-path failed for key " << kv.first; n_fail++; continue; }
      K3H_LOG << "For utterance " << kv.first << ", best cost " << g << " + " << a << " = " << (g + a) << " over " << ali.size() << " frames.";
      if (words_writer) words_writer->WriteInt32Vector(kv.first, words);
      if (ali_writer) ali_writer->WriteInt32Vector(kv.first, ali);
      if (!syms.empty()) {
        std::string line = kv.first + " ";
        for (int32_t w : words) { auto it = syms.find(w); if (it == syms.end()) K3H_ERR << "Word-id " << w << " not in symbol table."; line += it->second + " "; }
        std::cerr << line << "\n";
      }
      n_done++; n_frame += (int64_t)ali.size(); tot_graph += g; tot_ac += a;
    }
    if (words_writer) words_writer->Flush();
    if (ali_writer) ali_writer->Flush();
    K3H_LOG << "Overall cost per frame is " << ((tot_graph + tot_ac) / n_frame) << " = " << (tot_graph / n_frame) << " [graph] + " << (tot_ac / n_frame) <<
        " [acoustic] over " << n_frame << " frames.";
    K3H_LOG << "Done " << n_done << " lattices, failed for " << n_fail;
    return n_done != 0 ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
