// tests/adapter/nnet3_chain_grad.cc -- the gradient of the LF-MMI objective w.r.t. every parameter for one minibatch, the sequence nnet3/nnet-chain-training.cc:136-300 runs:
//   NnetComputer forward (training mode) -> ComputeChainObjfAndDeriv on the network output -> NnetComputer backward into a gradient nnet.
// Linked twice from this one source.  Oracle (oracle/_ref/bin/ref-nnet3-chain-grad): the reference's nnet3 + chain code on its CPU matrices (chain::ComputeChainObjfAndDeriv with a
// merged supervision FST).  MI355X (-DK3_ADAPTER, kaldi_amd/adapter/_build/nnet3-chain-grad): the reference's unmodified NnetComputer over the CuMatrix adapter, and the objective
// by k3_chain_objf_and_deriv on the adapter's device pointers -- what a Kaldi build that binds libk3hip.so would run.
//   nnet3-chain-grad <raw-nnet3-in> <frame-subsampling-factor> <input-matrix-in> <chain-spec-in> <objf-and-gradient-vector-out>
// chain-spec: int32 {magic 0x4b36, den_states, den_start, den_arcs, num_pdfs, num_sequences, frames_per_sequence, merged_states, merged_arcs, sup_states, sup_arcs}; float {leaky, l2_regularize, weight};
//   den FST, merged supervision FST (each: int64 arc_offsets[S+1]; int32 ilabel[A], nextstate[A]; float weight[A], final[S]); unmerged supervisions: int32 state_offsets[B+1], then the five arrays.
// output: a vector [objf, l2_term, weight, gradient...]
#include "base/kaldi-common.h"
#include "util/common-utils.h"
#include "nnet3/nnet-nnet.h"
#include "nnet3/nnet-utils.h"
#include "nnet3/nnet-optimize.h"
#include "nnet3/nnet-compute.h"
#ifdef K3_ADAPTER
#include "k3hip.h"
#else
#include "chain/chain-training.h"
#include "chain/chain-denominator.h"
namespace kaldi { namespace chain {
int32 ComputeFstStateTimes(const fst::StdVectorFst &fst, std::vector<int32> *state_times) {      // restated: chain-supervision.cc:663-700 (that file needs real OpenFst as a whole)
  const int32 n = fst.NumStates(); int32 total = -1; state_times->assign(n, -1); (*state_times)[0] = 0;
  for (int32 s = 0; s < n; s++) {
    const int32 nt = (*state_times)[s] + 1; if (nt <= 0) KALDI_ERR << "Input FST does not have required properties.";
    for (fst::ArcIterator<fst::StdVectorFst> it(fst, s); !it.Done(); it.Next()) { int32 &r = (*state_times)[it.Value().nextstate]; if (r == -1) r = nt; else if (r != nt) KALDI_ERR << "Input FST does not have required properties."; }
    if (fst.Final(s) != fst::TropicalWeight::Zero()) { if (total == -1) total = nt - 1; else if (total != nt - 1) KALDI_ERR << "Input FST does not have required properties."; }
  }
  return total;
} } }
#endif
namespace {
struct Reader { FILE *f; template <class T> void get(T *p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { std::cerr << "nnet3-chain-grad: short read\n"; exit(2); } } };
struct Csr { std::vector<int64_t> off; std::vector<int32_t> il, nx; std::vector<float> w, fin; void read(Reader &r, int32_t S, int32_t A) { off.resize(S + 1); il.resize(A); nx.resize(A); w.resize(A); fin.resize(S); r.get(off.data(), S + 1); r.get(il.data(), A); r.get(nx.data(), A); r.get(w.data(), A); r.get(fin.data(), S); } };
#ifndef K3_ADAPTER
void ToFst(const Csr &c, int32_t start, fst::StdVectorFst *out) {
  const int32_t S = (int32_t)c.fin.size(); for (int32_t s = 0; s < S; s++) out->AddState(); out->SetStart(start);
  for (int32_t s = 0; s < S; s++) { if (c.fin[s] != std::numeric_limits<float>::infinity()) out->SetFinal(s, fst::TropicalWeight(c.fin[s])); for (int64_t a = c.off[s]; a < c.off[s + 1]; a++) out->AddArc(s, fst::StdArc(c.il[a], c.il[a], fst::TropicalWeight(c.w[a]), c.nx[a])); }
}
#endif
}
int main(int argc, char *argv[]) {
  try {
    using namespace kaldi; using namespace kaldi::nnet3;
    ParseOptions po("nnet3-chain-grad <raw-nnet3-in> <frame-subsampling-factor> <input-matrix-in> <chain-spec-in> <objf-and-gradient-out>");
    po.Read(argc, argv);
    if (po.NumArgs() != 5) { po.PrintUsage(); return 1; }
    Nnet nnet; ReadKaldiObject(po.GetArg(1), &nnet); int32 s; if (!ConvertStringToInteger(po.GetArg(2), &s)) KALDI_ERR << "bad subsampling factor";
    Reader r{fopen(po.GetArg(4).c_str(), "rb")}; if (!r.f) KALDI_ERR << "cannot open " << po.GetArg(4);
    int32_t h[11]; float fo[3]; r.get(h, 11); r.get(fo, 3); if (h[0] != 0x4b36) KALDI_ERR << "bad chain spec";
    const int32 P = h[4], B = h[5], T = h[6];
    Csr den, merged, sup; den.read(r, h[1], h[3]); merged.read(r, h[7], h[8]); std::vector<int32_t> state_off(B + 1); r.get(state_off.data(), B + 1); sup.read(r, h[9], h[10]); fclose(r.f);
    SetBatchnormTestMode(false, &nnet); SetDropoutTestMode(false, &nnet);
    Nnet deriv_nnet(nnet); ScaleNnet(0.0, &deriv_nnet); SetNnetAsGradient(&deriv_nnet);
    int32 left, right; ComputeSimpleNnetContext(nnet, &left, &right);
    ComputationRequest request; request.need_model_derivative = true; request.store_component_stats = false;
    IoSpecification in; in.name = "input"; in.has_deriv = false; for (int32 t = -left; t <= (T - 1) * s + right; t++) for (int32 n = 0; n < B; n++) in.indexes.push_back(Index(n, t));
    IoSpecification out; out.name = "output"; out.has_deriv = true; for (int32 f = 0; f < T; f++) for (int32 n = 0; n < B; n++) out.indexes.push_back(Index(n, f * s));
    request.inputs.push_back(in); request.outputs.push_back(out);
    Matrix<BaseFloat> input; ReadKaldiObject(po.GetArg(3), &input);
    if (input.NumRows() != (int32)in.indexes.size() || nnet.OutputDim("output") != P) KALDI_ERR << "input / model do not fit the chain spec";
    NnetOptimizeOptions optimize_opts; CachingOptimizingCompilerOptions compiler_opts; CachingOptimizingCompiler compiler(nnet, optimize_opts, compiler_opts);
    std::shared_ptr<const NnetComputation> computation = compiler.Compile(request);
    NnetComputeOptions compute_opts; NnetComputer computer(compute_opts, *computation, nnet, &deriv_nnet);
    CuMatrix<BaseFloat> cu_in(input); computer.AcceptInput("input", &cu_in); computer.Run();
    const CuMatrixBase<BaseFloat> &nnet_output = computer.GetOutput("output");
    CuMatrix<BaseFloat> nnet_output_deriv(nnet_output.NumRows(), nnet_output.NumCols(), kUndefined);
    BaseFloat objf = 0, l2_term = 0, weight = 0;
#ifdef K3_ADAPTER
    k3_chain_den *kden = NULL; k3_chain_supervision *ksup = NULL;
    if (k3_chain_den_create((int32_t)den.fin.size(), h[2], P, den.off.data(), den.il.data(), den.nx.data(), den.w.data(), den.fin.data(), &kden) != K3_OK) KALDI_ERR << k3_last_error();
    if (k3_chain_supervision_create(B, T, P, fo[2], state_off.data(), sup.off.data(), sup.il.data(), sup.nx.data(), sup.w.data(), sup.fin.data(), &ksup) != K3_OK) KALDI_ERR << k3_last_error();
    k3_chain_training_opts o = {fo[1], 0.0f, fo[0], 0};
    if (k3_chain_objf_and_deriv(kden, ksup, &o, nnet_output.Data(), nnet_output.Stride(), nnet_output_deriv.Data(), nnet_output_deriv.Stride(), NULL, 0, &objf, &l2_term, &weight, NULL) != K3_OK) KALDI_ERR << k3_last_error();
    k3_chain_supervision_destroy(ksup); k3_chain_den_destroy(kden);
#else
    fst::StdVectorFst den_fst; ToFst(den, h[2], &den_fst); chain::DenominatorGraph den_graph(den_fst, P);
    chain::Supervision supervision; ToFst(merged, 0, &supervision.fst); supervision.weight = fo[2]; supervision.num_sequences = B; supervision.frames_per_sequence = T; supervision.label_dim = P;
    chain::ChainTrainingOptions opts; opts.leaky_hmm_coefficient = fo[0]; opts.l2_regularize = fo[1]; opts.out_of_range_regularize = 0.0;
    chain::ComputeChainObjfAndDeriv(opts, den_graph, supervision, nnet_output, &objf, &l2_term, &weight, &nnet_output_deriv, NULL);
#endif
    computer.AcceptInput("output", &nnet_output_deriv); computer.Run();
    Vector<BaseFloat> res(3 + NumParameters(deriv_nnet)); res(0) = objf; res(1) = l2_term; res(2) = weight;
    { SubVector<BaseFloat> g(res, 3, res.Dim() - 3); VectorizeNnet(deriv_nnet, &g); }
    WriteKaldiObject(res, po.GetArg(5), true);
    KALDI_LOG << "LF-MMI objf per frame " << objf / weight << " (+ l2 " << l2_term / weight << ") over " << weight << " frames; gradient of " << res.Dim() - 3 << " parameters";
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what(); return -1; }
}
