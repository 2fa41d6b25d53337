// tests/minifst_check.cc -- TEST INFRASTRUCTURE: runs the algorithms of the OpenFst stand-in (third_party/minifst/fst/fstlib.h) on an automaton given
// on stdin and prints the result, so that tests/test_minifst_independent.py can hold them to an INDEPENDENT implementation (scipy.sparse.csgraph for
// reachability / acyclicity, numpy.lexsort for arc order): the decoder / determinizer oracles are the reference's sources compiled over these containers,
// and nothing else checks the containers' algorithms.
//   minifst_check <connect|topsort|arcsort|invert|shortestpath|rmepsilon|rmepsilon_noconnect|project_input|project_output|map_ilabel_plus1>  < fst.txt  > result.txt
// text form, both ways:  "n <states> <start>" / "a <src> <dst> <ilabel> <olabel> <weight>" (arcs of a state in stored order) / "f <state> <weight>";
// topsort prints "cyclic" instead when TopSort returns false.
#include <fst/fstlib.h>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
int main(int argc, char **argv) {
  using namespace fst;
  if (argc != 2) { std::cerr << "usage: minifst_check <op>\n"; return 2; }
  const std::string op = argv[1];
  StdVectorFst f; std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream is(line); char k; is >> k;
    if (k == 'n') { int n, s; is >> n >> s; for (int i = 0; i < n; i++) f.AddState(); if (s >= 0) f.SetStart(s); }
    else if (k == 'a') { int s, d, il, ol; float w; is >> s >> d >> il >> ol >> w; f.AddArc(s, StdArc(il, ol, TropicalWeight(w), d)); }
    else if (k == 'f') { int s; float w; is >> s >> w; f.SetFinal(s, TropicalWeight(w)); }
  }
  if (op == "connect") Connect(&f);
  else if (op == "topsort") { if (!TopSort(&f)) { std::cout << "cyclic\n"; return 0; } }
  else if (op == "arcsort") ArcSort(&f, ILabelCompare<StdArc>());
  else if (op == "invert") Invert(&f);
  else if (op == "shortestpath") { StdVectorFst o; ShortestPath(f, &o); f = o; }
  else if (op == "rmepsilon") RmEpsilon(&f);
  else if (op == "rmepsilon_noconnect") RmEpsilon(&f, false);
  else if (op == "project_input") Project(&f, PROJECT_INPUT);
  else if (op == "project_output") Project(&f, PROJECT_OUTPUT);
  else if (op == "map_ilabel_plus1") Map(&f, [](const StdArc &a) { return StdArc(a.ilabel + 1, a.olabel, a.weight, a.nextstate); });
  else { std::cerr << "unknown op " << op << "\n"; return 2; }
  std::printf("n %d %d\n", (int)f.NumStates(), (int)f.Start());
  for (StateIterator<StdVectorFst> si(f); !si.Done(); si.Next()) {
    const int s = si.Value();
    for (ArcIterator<StdVectorFst> ai(f, s); !ai.Done(); ai.Next()) { const StdArc &a = ai.Value(); std::printf("a %d %d %d %d %.9g\n", s, (int)a.nextstate, (int)a.ilabel, (int)a.olabel, a.weight.Value()); }
    if (f.Final(s) != TropicalWeight::Zero()) std::printf("f %d %.9g\n", s, f.Final(s).Value());
  }
  return 0;
}
