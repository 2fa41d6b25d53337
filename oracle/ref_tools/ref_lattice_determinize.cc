// oracle/ref_tools/ref_lattice_determinize.cc -- TEST INFRASTRUCTURE.  Runs the REFERENCE's lattice determinization --
// lat/determinize-lattice-pruned.cc compiled unmodified from /root/reference against the OpenFst stand-in in third_party/minifst --
// the way the reference's programs call it, and prints the CompactLattices in Kaldi's text layout.  kaldi_amd/host/k3_lattice.cc (the
// restated determinizer behind the drop-in programs) is pinned to this program's output in tests/test_lattice_det.py.
//   ref-lattice-determinize word  <beam> <acoustic-scale> <lattices.txt> <out.txt>            = latbin/lattice-determinize-pruned.cc:96-140
//   ref-lattice-determinize phone <beam> <acoustic-scale> <lattices.txt> <out.txt> <model>    = latbin/lattice-determinize-phone-pruned.cc:100-140
// lattices.txt: Kaldi text archive of state-level lattices ("key", then "src dst ilabel olabel [graph,acoustic]" / "state [graph,acoustic]"
// lines, blank line after each lattice).  Extra arguments after these: --max-mem=N --delta=X --minimize=true (defaults of the programs).
// --minimize runs the reference's lat/push-lattice.cc and lat/minimize-lattice.cc (compiled unmodified as well).
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include "fstext/lattice-utils.h"
#include "hmm/transition-model.h"
#include "lat/determinize-lattice-pruned.h"
#include "lat/minimize-lattice.h"
#include "lat/push-lattice.h"
#include "util/common-utils.h"

namespace {
using kaldi::Lattice; using kaldi::CompactLattice; using kaldi::LatticeArc; using kaldi::LatticeWeight;
float Num(const std::string &t) { if (t == "Infinity") return std::numeric_limits<float>::infinity(); if (t == "-Infinity") return -std::numeric_limits<float>::infinity(); return std::strtof(t.c_str(), nullptr); }
LatticeWeight ParseWeight(const std::string &t) { const size_t c = t.find(','); return LatticeWeight(Num(t.substr(0, c)), Num(t.substr(c + 1))); }
void ScaleAcoustic(Lattice *lat, double scale) {             // fst::ScaleLattice(fst::AcousticLatticeScale(scale), lat) (fstext/lattice-utils-inl.h:169-200)
  for (int s = 0; s < lat->NumStates(); s++) {
    for (fst::MutableArcIterator<Lattice> it(lat, s); !it.Done(); it.Next()) { LatticeArc a = it.Value(); if (a.weight != LatticeWeight::Zero()) a.weight = LatticeWeight(a.weight.Value1(), (float)(scale * a.weight.Value2())); it.SetValue(a); }
    const LatticeWeight f = lat->Final(s); if (f != LatticeWeight::Zero()) lat->SetFinal(s, LatticeWeight(f.Value1(), (float)(scale * f.Value2())));
  }
}
void ScaleAcoustic(CompactLattice *lat, double scale) {
  typedef kaldi::CompactLatticeWeight W;
  for (int s = 0; s < lat->NumStates(); s++) {
    for (fst::MutableArcIterator<CompactLattice> it(lat, s); !it.Done(); it.Next()) { kaldi::CompactLatticeArc a = it.Value(); a.weight = W(LatticeWeight(a.weight.Weight().Value1(), (float)(scale * a.weight.Weight().Value2())), a.weight.String()); it.SetValue(a); }
    const W f = lat->Final(s); if (f != W::Zero()) lat->SetFinal(s, W(LatticeWeight(f.Weight().Value1(), (float)(scale * f.Weight().Value2())), f.String()));
  }
}
void Print(std::ostream &os, const std::string &key, const CompactLattice &c) {      // WriteCompactLattice text mode = fst::FstPrinter, acceptor (lat/kaldi-lattice.cc:73-92)
  os << key << " \n";
  auto state = [&](int s) {
    for (fst::ArcIterator<CompactLattice> it(c, s); !it.Done(); it.Next()) {
      const auto &a = it.Value(); os << s << "\t" << a.nextstate << "\t" << a.ilabel;
      if (a.weight != kaldi::CompactLatticeWeight::One()) os << "\t" << a.weight;
      os << "\n";
    }
    if (c.Final(s) != kaldi::CompactLatticeWeight::Zero()) { os << s; if (c.Final(s) != kaldi::CompactLatticeWeight::One()) os << "\t" << c.Final(s); os << "\n"; }
  };
  if (c.Start() != fst::kNoStateId) state(c.Start());
  for (int s = 0; s < c.NumStates(); s++) if (s != c.Start()) state(s);
  os << "\n";
}
}  // namespace

int main(int argc, char **argv) {
  try {
    if (argc < 6) { std::cerr << "usage: ref-lattice-determinize word|phone <beam> <acoustic-scale> <lattices.txt> <out.txt> [<model>] [--max-mem=N] [--delta=X]\n"; return 1; }
    const std::string mode = argv[1]; const double beam = (float)atof(argv[2]), acoustic_scale = (float)atof(argv[3]);     // BaseFloat options in the reference programs
    int max_mem = 50000000; float delta = fst::kDelta; std::string model; bool minimize = false, word_det = true, phone_det = true;
    for (int i = 6; i < argc; i++) {
      const std::string a = argv[i];
      if (a.compare(0, 10, "--max-mem=") == 0) max_mem = atoi(a.c_str() + 10); else if (a.compare(0, 8, "--delta=") == 0) delta = (float)atof(a.c_str() + 8);
      else if (a == "--minimize=true") minimize = true; else if (a == "--word-determinize=false") word_det = false; else if (a == "--phone-determinize=false") phone_det = false; else model = a;
    }
    kaldi::TransitionModel trans;
    if (mode == "phone") { bool binary; kaldi::Input ki(model, &binary); trans.Read(ki.Stream(), binary); }
    std::ifstream in(argv[4]); std::ofstream out(argv[5]);
    std::string line; int n_done = 0, n_fail = 0;
    while (std::getline(in, line)) {
      std::istringstream ks(line); std::string key; if (!(ks >> key)) continue;
      Lattice lat; bool first = true, compact = false; CompactLattice cin;
      auto need = [&](int s) { while (lat.NumStates() <= s) lat.AddState(); };
      auto cneed = [&](int s) { while (cin.NumStates() <= s) cin.AddState(); };
      auto cweight = [&](const std::string &t) {       // "graph,acoustic,t1_t2_..." -> CompactLatticeWeight
        const size_t c1 = t.find(','), c2 = t.find(',', c1 + 1); std::vector<kaldi::int32> str;
        for (size_t q = c2 + 1; q < t.size();) { size_t e = t.find('_', q); if (e == std::string::npos) e = t.size(); str.push_back(atoi(t.substr(q, e - q).c_str())); q = e + 1; }
        return kaldi::CompactLatticeWeight(LatticeWeight(Num(t.substr(0, c1)), Num(t.substr(c1 + 1, c2 - c1 - 1))), str);
      };
      while (std::getline(in, line)) {
        std::vector<std::string> col; { std::istringstream ss(line); std::string t; while (ss >> t) col.push_back(t); }
        if (col.empty()) break;
        auto commas = [](const std::string &t) { return std::count(t.begin(), t.end(), ','); };
        if (first && (col.size() == 3 || (col.size() == 4 && commas(col[3]) == 2) || (col.size() == 2 && commas(col[1]) == 2))) compact = true;
        if (compact) {       // a CompactLattice record: read it as one and let the REFERENCE's ConvertLattice (fstext/lattice-utils-inl.h:88-152) expand it below
          const int s = atoi(col[0].c_str()); cneed(s);
          if (first) { cin.SetStart(s); first = false; }
          if (col.size() <= 2) cin.SetFinal(s, col.size() == 2 ? cweight(col[1]) : kaldi::CompactLatticeWeight::One());
          else { const int d = atoi(col[1].c_str()); cneed(d); const int l = atoi(col[2].c_str()); cin.AddArc(s, kaldi::CompactLatticeArc(l, l, col.size() == 4 ? cweight(col[3]) : kaldi::CompactLatticeWeight::One(), d)); }
          continue;
        }
        const int s = atoi(col[0].c_str()); need(s);
        if (first) { lat.SetStart(s); first = false; }
        if (col.size() <= 2) lat.SetFinal(s, col.size() == 2 ? ParseWeight(col[1]) : LatticeWeight::One());
        else { const int d = atoi(col[1].c_str()); need(d); lat.AddArc(s, LatticeArc(atoi(col[2].c_str()), atoi(col[3].c_str()), col.size() == 5 ? ParseWeight(col[4]) : LatticeWeight::One(), d)); }
      }
      if (compact) fst::ConvertLattice(cin, &lat);
      CompactLattice clat; bool ok;
      if (mode == "word") {                                   // lattice-determinize-pruned.cc:104-131
        fst::DeterminizeLatticePrunedOptions opts; opts.max_mem = max_mem; opts.max_loop = 0; opts.delta = delta;
        fst::Invert(&lat);
        ScaleAcoustic(&lat, acoustic_scale);
        if (!fst::TopSort(&lat)) std::cerr << "WARNING could not topologically sort lattice " << key << "\n";
        fst::ArcSort(&lat, fst::ILabelCompare<LatticeArc>());
        ok = fst::DeterminizeLatticePruned(lat, beam, &clat, opts);
        fst::Connect(&clat);
        if (minimize) { fst::PushCompactLatticeStrings(&clat); fst::PushCompactLatticeWeights(&clat); fst::MinimizeCompactLattice(&clat); }      // lattice-determinize-pruned.cc:122-126
      } else {                                                // lattice-determinize-phone-pruned.cc:117-123
        fst::DeterminizeLatticePhonePrunedOptions opts; opts.max_mem = max_mem; opts.delta = delta; opts.minimize = minimize; opts.word_determinize = word_det; opts.phone_determinize = phone_det;
        ScaleAcoustic(&lat, acoustic_scale);
        ok = fst::DeterminizeLatticePhonePrunedWrapper(trans, &lat, beam, &clat, opts);
      }
      if (!ok) n_fail++;
      if (clat.Properties(fst::kTopSorted, true) == 0) fst::TopSort(&clat);          // TopSortCompactLatticeIfNeeded
      ScaleAcoustic(&clat, 1.0 / acoustic_scale);
      Print(out, key, clat);
      n_done++;
    }
    std::cerr << "ref-lattice-determinize: done " << n_done << ", stopped early on " << n_fail << "\n";
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
