// oracle/ref_tools/ref_chain_den.cc -- TEST INFRASTRUCTURE.  Runs the REFERENCE's LF-MMI denominator -- chain/chain-den-graph.cc (the
// DenominatorGraph constructor) and chain/chain-denominator.cc (DenominatorComputation::Forward / Backward, CPU path), both compiled
// unmodified from /root/reference against the OpenFst stand-in in third_party/minifst -- on a denominator FST and a network
// output read from one binary file, and writes the objective, Backward()'s verdict, the graph's initial probabilities and the
// derivative to another.  oracle/chain_oracle.py (the numpy restatement) and the HIP kernel (kaldi_amd/csrc/k3_chain.hip) are pinned to it.
//   ref-chain-den <in.bin> <out.bin>
// in.bin : int32 {magic 0x4b34, num_states, start, num_arcs, num_pdfs, num_sequences, frames_per_sequence}; float {leaky_hmm_coefficient, deriv_weight}
//          int64 arc_offsets[S+1]; int32 ilabel[A] (pdf-id + 1), nextstate[A]; float weight[A], final_cost[S]; float nnet_output[T*B][P]
// out.bin: float objf; int32 ok; float initial_probs[S]; float deriv[T*B][P]
#include <cstdio>
#include <iostream>
#include <limits>
#include <vector>
#include "chain/chain-denominator.h"

namespace {
struct Reader {
  FILE *f;
  template <class T> void get(T *p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { std::cerr << "ref-chain-den: short read\n"; exit(2); } }
};
}

int main(int argc, char **argv) {
  using namespace kaldi;
  if (argc != 3) { std::cerr << "usage: ref-chain-den in.bin out.bin\n"; return 1; }
  FILE *fi = fopen(argv[1], "rb"); if (!fi) { std::cerr << "cannot open " << argv[1] << "\n"; return 1; }
  Reader r{fi};
  int32_t hdr[7]; r.get(hdr, 7); float fo[2]; r.get(fo, 2);
  if (hdr[0] != 0x4b34) { std::cerr << "bad magic\n"; return 1; }
  const int32_t S = hdr[1], start = hdr[2], A = hdr[3], P = hdr[4], B = hdr[5], T = hdr[6];
  std::vector<int64_t> off(S + 1); r.get(off.data(), S + 1);
  std::vector<int32_t> il(A), nx(A); r.get(il.data(), A); r.get(nx.data(), A);
  std::vector<float> w(A), fin(S); r.get(w.data(), A); r.get(fin.data(), S);
  Matrix<BaseFloat> out(T * B, P);
  for (int32_t i = 0; i < T * B; i++) r.get(out.RowData(i), P);
  fclose(fi);
  fst::StdVectorFst den;
  for (int32_t s = 0; s < S; s++) den.AddState();
  den.SetStart(start);
  for (int32_t s = 0; s < S; s++) {
    if (fin[s] != std::numeric_limits<float>::infinity()) den.SetFinal(s, fst::TropicalWeight(fin[s]));
    for (int64_t a = off[s]; a < off[s + 1]; a++) den.AddArc(s, fst::StdArc(il[a], il[a], fst::TropicalWeight(w[a]), nx[a]));
  }
  chain::DenominatorGraph graph(den, P);
  chain::ChainTrainingOptions opts; opts.leaky_hmm_coefficient = fo[0];
  CuMatrix<BaseFloat> cu_out(out), cu_deriv(T * B, P);
  chain::DenominatorComputation comp(opts, graph, B, cu_out);
  const BaseFloat objf = comp.Forward();
  const bool ok = comp.Backward(fo[1], &cu_deriv);
  FILE *fw = fopen(argv[2], "wb"); if (!fw) { std::cerr << "cannot open " << argv[2] << "\n"; return 1; }
  const int32_t oki = ok ? 1 : 0;
  fwrite(&objf, sizeof(float), 1, fw); fwrite(&oki, sizeof(int32_t), 1, fw);
  Vector<BaseFloat> init(graph.InitialProbs()); fwrite(init.Data(), sizeof(float), S, fw);
  Matrix<BaseFloat> deriv(cu_deriv);
  for (int32_t i = 0; i < T * B; i++) fwrite(deriv.RowData(i), sizeof(float), P, fw);
  fclose(fw);
  return 0;
}
