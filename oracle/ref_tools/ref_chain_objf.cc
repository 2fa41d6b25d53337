// oracle/ref_tools/ref_chain_objf.cc -- TEST INFRASTRUCTURE.  Runs the REFERENCE's LF-MMI objective: chain::ComputeChainObjfAndDeriv (chain/chain-training.cc)
// with chain::NumeratorComputation (chain/chain-numerator.cc) and chain::DenominatorComputation (chain/chain-denominator.cc, chain-den-graph.cc), all four
// files compiled unmodified from /root/reference against the OpenFst stand-in, on a denominator FST, a MERGED supervision FST and a network output read from
// one binary file.  Writes objf, l2 term, weight, the derivative and the cross-entropy derivative.
// One function of chain/chain-supervision.cc (which needs real OpenFst as a whole) is restated below because NumeratorComputation's constructor calls it:
// ComputeFstStateTimes (:663-700).  --out-of-range-regularize is forced to 0: the reference applies that penalty on a coin flip (RandInt(0, 1), :273).
//   ref-chain-objf <in.bin> <out.bin>
// End-to-end supervisions (magic 0x4b37: the `sup` section is int32 state_offsets[B+1] followed by the B per-sequence FSTs concatenated in one CSR numbering, nextstate local to the
// sequence, as in k3_chain_supervision_create): Supervision::e2e_fsts is filled instead of fst and ComputeChainObjfAndDeriv takes its end-to-end branch
// (chain-training.cc:86-215) with chain::GenericNumeratorComputation (chain/chain-generic-numerator.cc, compiled unmodified).
// in.bin : int32 {magic 0x4b35, den_states, den_start, den_arcs, num_pdfs, num_sequences, frames_per_sequence, sup_states, sup_arcs}; float {leaky, l2_regularize, supervision weight}
//          den: int64 arc_offsets[S+1]; int32 ilabel[A], nextstate[A]; float weight[A], final_cost[S];   sup (merged): the same five arrays;   float nnet_output[T*B][P]
// out.bin: float {objf, l2_term, weight}; float deriv[T*B][P]; float xent_deriv[T*B][P]
#include <cstdio>
#include <iostream>
#include <limits>
#include <vector>
#include "chain/chain-training.h"
#include "chain/chain-denominator.h"

namespace kaldi { namespace chain {
int32 ComputeFstStateTimes(const fst::StdVectorFst &fst, std::vector<int32> *state_times) {      // restated: chain-supervision.cc:663-700
  const int32 n = fst.NumStates(); int32 total = -1;
  if (fst.Start() != 0) KALDI_ERR << "Expecting input FST start state to be zero";
  state_times->assign(n, -1); (*state_times)[0] = 0;
  for (int32 s = 0; s < n; s++) {
    const int32 nt = (*state_times)[s] + 1;
    if (nt <= 0) KALDI_ERR << "Input FST does not have required properties.";
    for (fst::ArcIterator<fst::StdVectorFst> it(fst, s); !it.Done(); it.Next()) {
      int32 &r = (*state_times)[it.Value().nextstate];
      if (r == -1) r = nt; else if (r != nt) KALDI_ERR << "Input FST does not have required properties.";
    }
    if (fst.Final(s) != fst::TropicalWeight::Zero()) { if (total == -1) total = nt - 1; else if (total != nt - 1) KALDI_ERR << "Input FST does not have required properties."; }
  }
  if (total < 0) KALDI_ERR << "Input FST does not have required properties.";
  return total;
}
} }

namespace {
struct Reader { FILE *f; template <class T> void get(T *p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { std::cerr << "ref-chain-objf: short read\n"; exit(2); } } };
void ReadFst(Reader &r, int32_t S, int32_t A, int32_t start, fst::StdVectorFst *out) {
  std::vector<int64_t> off(S + 1); r.get(off.data(), S + 1); std::vector<int32_t> il(A), nx(A); r.get(il.data(), A); r.get(nx.data(), A);
  std::vector<float> w(A), fin(S); r.get(w.data(), A); r.get(fin.data(), S);
  for (int32_t s = 0; s < S; s++) out->AddState();
  out->SetStart(start);
  for (int32_t s = 0; s < S; s++) {
    if (fin[s] != std::numeric_limits<float>::infinity()) out->SetFinal(s, fst::TropicalWeight(fin[s]));
    for (int64_t a = off[s]; a < off[s + 1]; a++) out->AddArc(s, fst::StdArc(il[a], il[a], fst::TropicalWeight(w[a]), nx[a]));
  }
}
}

int main(int argc, char **argv) {
  using namespace kaldi;
  if (argc != 3) { std::cerr << "usage: ref-chain-objf in.bin out.bin\n"; return 1; }
  FILE *fi = fopen(argv[1], "rb"); if (!fi) { std::cerr << "cannot open " << argv[1] << "\n"; return 1; }
  Reader r{fi}; int32_t h[9]; r.get(h, 9); float fo[3]; r.get(fo, 3);
  if (h[0] != 0x4b35 && h[0] != 0x4b37) { std::cerr << "bad magic\n"; return 1; }
  const int32_t P = h[4], B = h[5], T = h[6];
  fst::StdVectorFst den; ReadFst(r, h[1], h[3], h[2], &den);
  chain::Supervision sup;
  if (h[0] == 0x4b35) ReadFst(r, h[7], h[8], 0, &sup.fst);
  else {
    const int32_t S = h[7], A = h[8]; std::vector<int32_t> so(B + 1); r.get(so.data(), B + 1);
    std::vector<int64_t> off(S + 1); r.get(off.data(), S + 1); std::vector<int32_t> il(A), nx(A); r.get(il.data(), A); r.get(nx.data(), A); std::vector<float> w(A), fin(S); r.get(w.data(), A); r.get(fin.data(), S);
    sup.e2e_fsts.resize(B);
    for (int32_t b = 0; b < B; b++) {
      fst::StdVectorFst &f = sup.e2e_fsts[b];
      for (int32_t s = so[b]; s < so[b + 1]; s++) f.AddState();
      f.SetStart(0);
      for (int32_t s = so[b]; s < so[b + 1]; s++) {
        if (fin[s] != std::numeric_limits<float>::infinity()) f.SetFinal(s - so[b], fst::TropicalWeight(fin[s]));
        for (int64_t a = off[s]; a < off[s + 1]; a++) f.AddArc(s - so[b], fst::StdArc(il[a], il[a], fst::TropicalWeight(w[a]), nx[a]));
      }
    }
  }
  sup.weight = fo[2]; sup.num_sequences = B; sup.frames_per_sequence = T; sup.label_dim = P;
  Matrix<BaseFloat> out(T * B, P); for (int32_t i = 0; i < T * B; i++) r.get(out.RowData(i), P);
  fclose(fi);
  chain::DenominatorGraph graph(den, P);
  chain::ChainTrainingOptions opts; opts.leaky_hmm_coefficient = fo[0]; opts.l2_regularize = fo[1]; opts.out_of_range_regularize = 0.0;
  CuMatrix<BaseFloat> cu_out(out), deriv(T * B, P), xent;
  BaseFloat objf, l2, weight;
  chain::ComputeChainObjfAndDeriv(opts, graph, sup, cu_out, &objf, &l2, &weight, &deriv, &xent);
  FILE *fw = fopen(argv[2], "wb"); if (!fw) return 1;
  float res[3] = {objf, l2, weight}; fwrite(res, sizeof(float), 3, fw);
  Matrix<BaseFloat> d(deriv), x(xent);
  for (int32_t i = 0; i < T * B; i++) fwrite(d.RowData(i), sizeof(float), P, fw);
  for (int32_t i = 0; i < T * B; i++) fwrite(x.RowData(i), sizeof(float), P, fw);
  fclose(fw);
  return 0;
}
