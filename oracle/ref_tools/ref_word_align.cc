// oracle/ref_tools/ref_word_align.cc -- TEST INFRASTRUCTURE.  Runs the REFERENCE's WordAlignLattice (lat/word-align-lattice.cc, compiled unmodified from /root/reference
// against the OpenFst stand-in in third_party/minifst) and, behind it, its MinimumBayesRisk (lat/sausages.cc) -- LatticePostprocessor::GetCTM's two steps
// (cudadecoder/lattice-postprocessor.cc:55-110) -- on CompactLattices made from raw lattices by the reference's ConvertLattice.  kaldi_amd/host/k3_mbr.cc is pinned to this output
// (tests/test_word_align.py).
//   ref-word-align <final.mdl> <word_boundary.int> <lattices.txt> <out.txt> [reorder(1|0) [silence-label [partial-word-label [max-expand]]]]
// Output per lattice: "key", "ok 0|1" (WordAlignLattice's return value), "states n start s", one line per arc "a src dst label graph acoustic tid_tid_..." in stored order,
// "f state graph acoustic" per final state, then "words ..." / "times ..." / "conf ..." of MinimumBayesRisk on the aligned lattice.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include "fstext/lattice-utils.h"
#include "hmm/transition-model.h"
#include "lat/kaldi-lattice.h"
#include "lat/sausages.h"
#include "lat/word-align-lattice.h"
namespace kaldi {
int32 CompactLatticeStateTimes(const CompactLattice &lat, std::vector<int32> *times) {      // lat/lattice-functions.cc:109-147 (see ref_mbr.cc)
  KALDI_ASSERT(lat.Start() == 0);
  const int32 n = lat.NumStates(); times->clear(); times->resize(n, -1); (*times)[0] = 0; int32 utt_len = -1;
  for (int32 s = 0; s < n; s++) {
    const int32 cur = (*times)[s];
    for (fst::ArcIterator<CompactLattice> it(lat, s); !it.Done(); it.Next()) {
      const CompactLatticeArc &arc = it.Value(); const int32 len = (int32)arc.weight.String().size();
      if ((*times)[arc.nextstate] == -1) (*times)[arc.nextstate] = cur + len; else KALDI_ASSERT((*times)[arc.nextstate] == cur + len);
    }
    if (lat.Final(s) != CompactLatticeWeight::Zero()) { const int32 l = (*times)[s] + (int32)lat.Final(s).String().size(); utt_len = std::max(utt_len, l); }
  }
  return utt_len < 0 ? 0 : utt_len;
}
}
namespace {
using kaldi::Lattice; using kaldi::CompactLattice; using kaldi::LatticeArc; using kaldi::LatticeWeight;
float Num(const std::string &t) { if (t == "Infinity") return std::numeric_limits<float>::infinity(); if (t == "-Infinity") return -std::numeric_limits<float>::infinity(); return std::strtof(t.c_str(), nullptr); }
LatticeWeight ParseWeight(const std::string &t) { const size_t c = t.find(','); return LatticeWeight(Num(t.substr(0, c)), Num(t.substr(c + 1))); }
}
int main(int argc, char **argv) {
  if (argc < 5) { std::cerr << "usage: ref-word-align <final.mdl> <word_boundary.int> <lattices.txt> <out.txt> [reorder [silence-label [partial-word-label [max-expand]]]]\n"; return 1; }
  try {
    kaldi::TransitionModel trans; { bool binary; kaldi::Input ki(argv[1], &binary); trans.Read(ki.Stream(), binary); }
    kaldi::WordBoundaryInfoNewOpts wo; if (argc > 5) wo.reorder = atoi(argv[5]) != 0; if (argc > 6) wo.silence_label = atoi(argv[6]); if (argc > 7) wo.partial_word_label = atoi(argv[7]);
    const float max_expand = argc > 8 ? (float)atof(argv[8]) : 0.0f;
    kaldi::WordBoundaryInfo info(wo, argv[2]);
    std::ifstream in(argv[3]); std::ofstream out(argv[4]); std::string line; out.precision(9);
    while (std::getline(in, line)) {
      std::istringstream ks(line); std::string key; if (!(ks >> key)) continue;
      Lattice lat; bool first = true;
      auto need = [&](int s) { while (lat.NumStates() <= s) lat.AddState(); };
      while (std::getline(in, line)) {
        std::vector<std::string> col; { std::istringstream ss(line); std::string t; while (ss >> t) col.push_back(t); }
        if (col.empty()) break;
        const int s = atoi(col[0].c_str()); need(s);
        if (first) { lat.SetStart(s); first = false; }
        if (col.size() <= 2) lat.SetFinal(s, col.size() == 2 ? ParseWeight(col[1]) : LatticeWeight::One());
        else { const int d = atoi(col[1].c_str()); need(d); lat.AddArc(s, LatticeArc(atoi(col[2].c_str()), atoi(col[3].c_str()), col.size() == 5 ? ParseWeight(col[4]) : LatticeWeight::One(), d)); }
      }
      fst::Connect(&lat);
      CompactLattice c; fst::ConvertLattice(lat, &c);
      const int32 max_states = max_expand > 0 ? (int32)(1000 + max_expand * c.NumStates()) : 0;      // lattice-postprocessor.cc:70-74
      CompactLattice aligned; const bool ok = c.NumStates() == 0 ? true : kaldi::WordAlignLattice(c, trans, info, max_states, &aligned);
      out << key << "\nok " << (ok ? 1 : 0) << "\nstates " << aligned.NumStates() << " start " << aligned.Start() << "\n";
      for (int32 s = 0; s < aligned.NumStates(); s++) {
        for (fst::ArcIterator<CompactLattice> it(aligned, s); !it.Done(); it.Next()) {
          const kaldi::CompactLatticeArc &a = it.Value(); out << "a " << s << " " << a.nextstate << " " << a.ilabel << " " << a.weight.Weight().Value1() << " " << a.weight.Weight().Value2() << " ";
          for (size_t k = 0; k < a.weight.String().size(); k++) out << (k ? "_" : "") << a.weight.String()[k];
          out << "\n";
        }
        if (aligned.Final(s) != kaldi::CompactLatticeWeight::Zero()) out << "f " << s << " " << aligned.Final(s).Weight().Value1() << " " << aligned.Final(s).Weight().Value2() << "\n";
      }
      if (aligned.NumStates() > 0) {
        kaldi::MinimumBayesRisk mbr(aligned, kaldi::MinimumBayesRiskOptions());
        out << "words"; for (int32 w : mbr.GetOneBest()) out << " " << w;
        out << "\ntimes"; for (const auto &t : mbr.GetOneBestTimes()) out << " " << t.first << " " << t.second;
        out << "\nconf"; for (float x : mbr.GetOneBestConfidences()) out << " " << x;
        out << "\n";
      }
      out << "end\n";
    }
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return 1; }
}
