// oracle/ref_tools/ref_convert_lattice.cc -- TEST INFRASTRUCTURE.  Runs the REFERENCE's ConvertLattice(Lattice -> CompactLattice)
// (fstext/lattice-utils-inl.h:33-86 with Factor, fstext/factor-inl.h, both compiled unmodified from /root/reference against the
// OpenFst stand-in in third_party/minifst) and prints the CompactLattices in Kaldi's text layout; kaldi_amd/host/k3_lattice.cc's
// ConvertLattice is pinned to this output in tests/test_lattice_det.py.
//   ref-convert-lattice <lattices.txt> <out.txt>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include "fstext/lattice-utils.h"
#include "lat/kaldi-lattice.h"
namespace {
using kaldi::Lattice; using kaldi::CompactLattice; using kaldi::LatticeArc; using kaldi::LatticeWeight;
float Num(const std::string &t) { if (t == "Infinity") return std::numeric_limits<float>::infinity(); if (t == "-Infinity") return -std::numeric_limits<float>::infinity(); return std::strtof(t.c_str(), nullptr); }
LatticeWeight ParseWeight(const std::string &t) { const size_t c = t.find(','); return LatticeWeight(Num(t.substr(0, c)), Num(t.substr(c + 1))); }
}
int main(int argc, char **argv) {
  if (argc != 3) { std::cerr << "usage: ref-convert-lattice <lattices.txt> <out.txt>\n"; return 1; }
  try {
    std::ifstream in(argv[1]); std::ofstream out(argv[2]); std::string line;
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
      fst::Connect(&lat);                       // the decoding programs trim before they convert (decoder-wrappers.cc:353)
      CompactLattice c;
      fst::ConvertLattice(lat, &c);
      out << key << " \n";
      auto state = [&](int s) {
        for (fst::ArcIterator<CompactLattice> it(c, s); !it.Done(); it.Next()) {
          const auto &a = it.Value(); out << s << "\t" << a.nextstate << "\t" << a.ilabel;
          if (a.weight != kaldi::CompactLatticeWeight::One()) out << "\t" << a.weight;
          out << "\n";
        }
        if (c.Final(s) != kaldi::CompactLatticeWeight::Zero()) { out << s; if (c.Final(s) != kaldi::CompactLatticeWeight::One()) out << "\t" << c.Final(s); out << "\n"; }
      };
      if (c.Start() != fst::kNoStateId) state(c.Start());
      for (int s = 0; s < c.NumStates(); s++) if (s != c.Start()) state(s);
      out << "\n";
    }
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return -1; }
}
