// oracle/ref_tools/ref_mbr.cc -- TEST INFRASTRUCTURE.  Runs the REFERENCE's MinimumBayesRisk (lat/sausages.cc, compiled unmodified from /root/reference against
// the OpenFst stand-in in third_party/minifst) on CompactLattices made from raw lattices by the reference's ConvertLattice; kaldi_amd/host/k3_mbr.cc is pinned to this
// output in tests/test_lattice_det.py.      ref-mbr <lattices.txt> <out.txt> [decode-mbr(1|0) [print-silence(0|1)]]
// Output per lattice: "key", "words w...", "times b e ...", "conf c ...", "risk r", "bins n" and per bin "bin b e word:post ...".
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include "fstext/lattice-utils.h"
#include "lat/kaldi-lattice.h"
#include "lat/sausages.h"
namespace kaldi {
// lat/lattice-functions.cc:109-147 (that file as a whole needs much more of OpenFst than the stand-in has): times of the states of a topologically sorted CompactLattice
int32 CompactLatticeStateTimes(const CompactLattice &lat, std::vector<int32> *times) {
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
  if (argc < 3) { std::cerr << "usage: ref-mbr <lattices.txt> <out.txt> [decode-mbr [print-silence]]\n"; return 1; }
  try {
    kaldi::MinimumBayesRiskOptions opts; if (argc > 3) opts.decode_mbr = atoi(argv[3]) != 0; if (argc > 4) opts.print_silence = atoi(argv[4]) != 0;
    std::ifstream in(argv[1]); std::ofstream out(argv[2]); std::string line; out.precision(9);
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
      kaldi::MinimumBayesRisk mbr(c, opts);
      out << key << "\nwords"; for (int32 w : mbr.GetOneBest()) out << " " << w;
      out << "\ntimes"; for (const auto &t : mbr.GetOneBestTimes()) out << " " << t.first << " " << t.second;
      out << "\nconf"; for (float x : mbr.GetOneBestConfidences()) out << " " << x;
      out << "\nrisk " << mbr.GetBayesRisk() << "\nbins " << mbr.GetSausageStats().size() << "\n";
      for (size_t q = 0; q < mbr.GetSausageStats().size(); q++) {
        out << "bin " << mbr.GetSausageTimes()[q].first << " " << mbr.GetSausageTimes()[q].second;
        for (const auto &e : mbr.GetSausageStats()[q]) out << " " << e.first << ":" << e.second;
        out << "\n";
      }
    }
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return 1; }
}
