// kaldi_amd/adapter/nnet3-chain-train.cc -- the drop-in for chainbin/nnet3-chain-train.cc:33-104 on the MI355X: N iterations of chain (LF-MMI) TRAINING over a
// list of minibatches, the sequence NnetChainTrainer::Train / TrainInternal runs
// (nnet3/nnet-chain-training.cc:60-144): NnetComputer over (nnet, delta_nnet) with component statistics stored, forward in training mode, ComputeChainObjfAndDeriv,
// backward (every updatable component's Update(): natural-gradient preconditioning (nnet3/natural-gradient-online.cc) times its learning rate into delta_nnet),
// ApplyL2Regularization, UpdateNnetWithMaxChange (per-component and global max-change), ScaleBatchnormStats, ConstrainOrthonormal (the semi-orthogonal
// constraint of the TDNN-F bottlenecks), momentum.  Everything but the objective is the reference's own unmodified code.
// Linked twice from this one source.  Oracle (oracle/_ref/bin/ref-nnet3-chain-train): on the reference's CPU matrices, objective by chain::ComputeChainObjfAndDeriv.
// MI355X (-DK3_ADAPTER, kaldi_amd/adapter/_build/nnet3-chain-train): the same objects over the CuMatrix adapter (every matrix operation of the forward pass, the
// backprop, the preconditioner and the constraint is a k3_mat_* / k3_vec_* kernel), objective by k3_chain_objf_and_deriv on the adapter's device pointers.
// nnet3-chain-train <raw-nnet3-in> <frame-subsampling-factor> <input-matrix-in>[,...] <chain-spec-in>[,...] <num-iters> <learning-rate> <momentum>
// <raw-nnet3-out> <objf-vector-out>
// The reference program reads NnetChainExample archives (nnet3/nnet-chain-example.cc), whose Supervision objects are OpenFst objects; /root/reference does not
// vendor OpenFst, so the
// minibatches come as pairs of files instead (iteration i trains on pair i mod n; all pairs the same shape, like the minibatches merged from one egs archive):
//   input-matrix  Kaldi binary Matrix<float> [(T-1)*s + 1 + left + right frames x num_sequences, sequence-minor rows (n, t) like a merged NnetChainExample's input, feature dim columns]
//   chain-spec    int32[11] {0x4b36, den states, den start, den arcs, num pdfs P, num sequences B, frames per sequence T, merged-supervision states, arcs, per-sequence states, arcs}
// (0x4b38 instead of 0x4b36: END-TO-END supervisions -- the per-sequence FSTs are Supervision::e2e_fsts, chain/chain-generic-numerator.cc; the merged FST is
// ignored)
// float[3] {leaky-hmm-coefficient, l2-regularize, supervision weight}; the denominator FST, the merged supervision FST (chain::Supervision::fst after
// MergeSupervision),
//                 int32[B+1] first state of every sequence's own FST, the B per-sequence FSTs concatenated; every FST as CSR: int64[S+1] arc offsets, int32[A] pdf-id + 1 labels,
//                 int32[A] next states, float[A] weights, float[S] final costs.  (tools/debug_chain_train.py and bench.py write it.)
// objf-vector: per iteration [objf, l2_term, weight], then the parameters of the trained model.
// Data-parallel: K3_TRAIN_ID_FILE / K3_TRAIN_RANK / K3_TRAIN_WORLD make the process one rank of a job whose parameter changes are all-reduced over RCCL every
// iteration (k3_comm_allreduce_f32).
#include "base/kaldi-common.h"
#include "base/timer.h"
#include "util/common-utils.h"
#include "nnet3/nnet-nnet.h"
#include "nnet3/nnet-utils.h"
#include "nnet3/nnet-optimize.h"
#include "nnet3/nnet-compute.h"
#ifdef K3_ADAPTER
#include "k3hip.h"
extern "C" void *k3_adapter_stream();      // kaldi_amd/adapter/cu-k3.cc: the calling thread's stream (every CuMatrix operation of this thread is queued on it)
#else
#include "chain/chain-training.h"
#include "chain/chain-denominator.h"
namespace kaldi { namespace chain {
int32 ComputeFstStateTimes(const fst::StdVectorFst &fst, std::vector<int32> *state_times) {      // restated: chain-supervision.cc:663-700 (that file needs real OpenFst as a whole)
  const int32 n = fst.NumStates(); int32 total = -1; state_times->assign(n, -1); (*state_times)[0] = 0;
  for (int32 s = 0; s < n; s++) {
    const int32 nt = (*state_times)[s] + 1; if (nt <= 0) KALDI_ERR << "Input FST does not have required properties.";
    for (fst::ArcIterator<fst::StdVectorFst> it(fst, s); !it.Done(); it.Next()) {
      int32 &r = (*state_times)[it.Value().nextstate];
      if (r == -1) r = nt;
      else if (r != nt) KALDI_ERR << "Input FST does not have required properties.";
    }
    if (fst.Final(s) != fst::TropicalWeight::Zero()) { if (total == -1) total = nt - 1; else if (total != nt - 1) KALDI_ERR << "Input FST does not have required properties."; }
  }
  return total;
} } }
#endif
namespace {
struct Reader { FILE *f; template <class T> void get(T *p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { std::cerr << "nnet3-chain-train: short read\n"; exit(2); } } };
struct Csr {
  std::vector<int64_t> off;
  std::vector<int32_t> il, nx;
  std::vector<float> w, fin;
  void read(Reader &r, int32_t S, int32_t A) {
    off.resize(S + 1);
    il.resize(A);
    nx.resize(A);
    w.resize(A);
    fin.resize(S);
    r.get(off.data(), S + 1);
    r.get(il.data(), A);
    r.get(nx.data(), A);
    r.get(w.data(), A);
    r.get(fin.data(), S);
  }
};
#ifndef K3_ADAPTER
void ToFst(const Csr &c, int32_t start, fst::StdVectorFst *out) {
  const int32_t S = (int32_t)c.fin.size(); for (int32_t s = 0; s < S; s++) out->AddState(); out->SetStart(start);
  for (int32_t s = 0; s < S; s++) {
    if (c.fin[s] != std::numeric_limits<float>::infinity()) out->SetFinal(s, fst::TropicalWeight(c.fin[s]));
    for (int64_t a = c.off[s]; a < c.off[s + 1]; a++) out->AddArc(s, fst::StdArc(c.il[a], c.il[a], fst::TropicalWeight(c.w[a]), c.nx[a]));
  }
}
#endif
}
int main(int argc, char *argv[]) {
  try {
    using namespace kaldi; using namespace kaldi::nnet3;
    ParseOptions po("nnet3-chain-train <raw-nnet3-in> <frame-subsampling-factor> <input-matrix-in> <chain-spec-in> <num-iters> <learning-rate> <momentum> <raw-nnet3-out> <objf-vector-out>");
    po.Read(argc, argv);
    if (po.NumArgs() != 9) { po.PrintUsage(); return 1; }
    Nnet nnet; ReadKaldiObject(po.GetArg(1), &nnet); int32 s; if (!ConvertStringToInteger(po.GetArg(2), &s)) KALDI_ERR << "bad subsampling factor";
    std::vector<std::string> in_files, spec_files; SplitStringToVector(po.GetArg(3), ",", true, &in_files); SplitStringToVector(po.GetArg(4), ",", true, &spec_files);
    if (in_files.empty() || in_files.size() != spec_files.size()) KALDI_ERR << "need as many input matrices as chain specs";
    const size_t num_mb = spec_files.size();
    struct Minibatch { int32_t h[11]; float fo[3]; Csr den, merged, sup; std::vector<int32_t> state_off; };
    std::vector<Minibatch> mbs(num_mb);
    for (size_t m = 0; m < num_mb; m++) {
      Reader r{fopen(spec_files[m].c_str(), "rb")}; if (!r.f) KALDI_ERR << "cannot open " << spec_files[m];
      Minibatch &mb = mbs[m]; r.get(mb.h, 11); r.get(mb.fo, 3); if (mb.h[0] != 0x4b36 && mb.h[0] != 0x4b38) KALDI_ERR << "bad chain spec " << spec_files[m];
      if (mb.h[0] != mbs[0].h[0]) KALDI_ERR << "ordinary and end-to-end minibatches cannot be mixed";
      mb.den.read(r, mb.h[1], mb.h[3]);
      mb.merged.read(r, mb.h[7], mb.h[8]);
      mb.state_off.resize(mb.h[5] + 1);
      r.get(mb.state_off.data(), mb.h[5] + 1);
      mb.sup.read(r, mb.h[9], mb.h[10]);
      fclose(r.f);
      if (mb.h[4] != mbs[0].h[4] || mb.h[5] != mbs[0].h[5] || mb.h[6] != mbs[0].h[6] || mb.h[1] != mbs[0].h[1] || mb.h[3] != mbs[0].h[3]) KALDI_ERR <<
          "the minibatches differ in shape or denominator graph";
    }
    const int32_t *h = mbs[0].h; const float *fo = mbs[0].fo; const Csr &den = mbs[0].den;
    const int32 P = h[4], B = h[5], T = h[6];
    int32 num_iters; double lrate, momentum;
    if (!ConvertStringToInteger(po.GetArg(5), &num_iters) || !ConvertStringToReal(po.GetArg(6), &lrate) ||
        !ConvertStringToReal(po.GetArg(7), &momentum)) KALDI_ERR << "bad iteration count / learning rate / momentum";
    const BaseFloat max_param_change = 2.0, l2_regularize_factor = 1.0, batchnorm_stats_scale = 0.8;      // NnetTrainerOptions' defaults (nnet3/nnet-training.h:36-80)
    SetBatchnormTestMode(false, &nnet); SetDropoutTestMode(false, &nnet); SetLearningRate(lrate, &nnet);
    ZeroComponentStats(&nnet);                                                                            // NnetChainTrainer::NnetChainTrainer, nnet-chain-training.cc:36-41
    Nnet *delta_nnet = nnet.Copy(); ScaleNnet(0.0, delta_nnet);
    int32 left, right; ComputeSimpleNnetContext(nnet, &left, &right);
    ComputationRequest request; request.need_model_derivative = true; request.store_component_stats = true;
    IoSpecification in;
    in.name = "input";
    in.has_deriv = false;
    for (int32 t = -left; t <= (T - 1) * s + right; t++) for (int32 n = 0; n < B; n++) in.indexes.push_back(Index(n, t));
    IoSpecification out; out.name = "output"; out.has_deriv = true; for (int32 f = 0; f < T; f++) for (int32 n = 0; n < B; n++) out.indexes.push_back(Index(n, f * s));
    request.inputs.push_back(in); request.outputs.push_back(out);
    std::vector<Matrix<BaseFloat> > inputs(num_mb);
    for (size_t m = 0; m < num_mb; m++) {
      ReadKaldiObject(in_files[m], &inputs[m]);
      if (inputs[m].NumRows() != (int32)in.indexes.size() || nnet.OutputDim("output") != P) KALDI_ERR << "input / model do not fit the chain spec";
    }
    NnetOptimizeOptions optimize_opts; CachingOptimizingCompilerOptions compiler_opts; CachingOptimizingCompiler compiler(nnet, optimize_opts, compiler_opts);
    std::shared_ptr<const NnetComputation> computation = compiler.Compile(request);
#ifdef K3_ADAPTER
    k3_chain_den *kden = NULL; std::vector<k3_chain_supervision *> ksups(num_mb, NULL);
    if (k3_chain_den_create((int32_t)den.fin.size(), h[2], P, den.off.data(), den.il.data(), den.nx.data(), den.w.data(), den.fin.data(),
        &kden) != K3_OK) KALDI_ERR << k3_last_error();
    // end-to-end (flat-start) supervisions: the per-sequence FSTs are Supervision::e2e_fsts (self-loops, several final states), the merged FST is unused
    const bool e2e = h[0] == 0x4b38;
    for (size_t m = 0; m < num_mb; m++) {
      const Minibatch &mb = mbs[m];
      if ((e2e ? k3_chain_supervision_create_e2e : k3_chain_supervision_create)(B, T, P, mb.fo[2], mb.state_off.data(), mb.sup.off.data(), mb.sup.il.data(),
          mb.sup.nx.data(), mb.sup.w.data(), mb.sup.fin.data(), &ksups[m]) != K3_OK) KALDI_ERR << k3_last_error();
    }
#else
    fst::StdVectorFst den_fst; ToFst(den, h[2], &den_fst); chain::DenominatorGraph den_graph(den_fst, P);
    std::vector<chain::Supervision> supervisions(num_mb);
    const bool e2e = h[0] == 0x4b38;
    for (size_t m = 0; m < num_mb; m++) {
      chain::Supervision &sv = supervisions[m]; sv.weight = mbs[m].fo[2]; sv.num_sequences = B; sv.frames_per_sequence = T; sv.label_dim = P;
      if (!e2e) { ToFst(mbs[m].merged, 0, &sv.fst); continue; }
      const Minibatch &mb = mbs[m]; sv.e2e_fsts.resize(B);      // the per-sequence FSTs of the spec as Supervision::e2e_fsts
      for (int32 b = 0; b < B; b++) {
        fst::StdVectorFst &f = sv.e2e_fsts[b]; const int32 s0 = mb.state_off[b], S = mb.state_off[b + 1] - s0;
        for (int32 st = 0; st < S; st++) f.AddState();
        f.SetStart(0);
        for (int32 st = 0; st < S; st++) {
          if (mb.sup.fin[s0 + st] != std::numeric_limits<float>::infinity()) f.SetFinal(st, fst::TropicalWeight(mb.sup.fin[s0 + st]));
          for (int64_t a = mb.sup.off[s0 + st]; a < mb.sup.off[s0 + st + 1]; a++) f.AddArc(st,
              fst::StdArc(mb.sup.il[a], mb.sup.il[a], fst::TropicalWeight(mb.sup.w[a]), mb.sup.nx[a]));
        }
      }
    }
#endif
#ifdef K3_ADAPTER
    // K3_TRAIN_ID_FILE [+ K3_TRAIN_RANK / K3_TRAIN_WORLD]: this process is one rank of a data-parallel job (k3_comm_create: RCCL communicator through a rendezvous file)
    void *comm = NULL; int32 world = 1;
    if (const char *idf = getenv("K3_TRAIN_ID_FILE")) {
      const int32 rank = getenv("K3_TRAIN_RANK") ? atoi(getenv("K3_TRAIN_RANK")) : 0; world = getenv("K3_TRAIN_WORLD") ? atoi(getenv("K3_TRAIN_WORLD")) : 1;
      if (k3_comm_create(idf, rank, world, 120, &comm) != K3_OK) KALDI_ERR << k3_last_error();
      KALDI_LOG << "data-parallel rank " << rank << " of " << world << ": parameter changes all-reduced over RCCL every iteration";
    }
#endif
    MaxChangeStats max_change_stats(nnet);
    Vector<BaseFloat> objfs(3 * num_iters + NumParameters(nnet));      // per iteration [objf, l2_term, weight], then every parameter of the trained model (VectorizeNnet)
    std::vector<CuMatrix<BaseFloat> > cu_in_orig(num_mb); for (size_t m = 0; m < num_mb; m++) cu_in_orig[m] = inputs[m];
    for (int32 iter = 0; iter < num_iters; iter++) {      // TrainInternal
      Timer iter_timer;
      // The reference draws from the host's rand() for its sampled decisions (which minibatches store statistics / repair gradients / get the orthonormal
      // constraint) AND inside its CPU
      // chain code's self-checks (chain-denominator.cc: RandInt(0, 10)), which k3_chain_objf_and_deriv does not have: reseeded at the same two points in both
      // builds, every such decision is the same.
      srand(2 * iter + 1);
      NnetComputeOptions compute_opts; NnetComputer computer(compute_opts, *computation, &nnet, delta_nnet);
      const size_t mi = (size_t)iter % num_mb;      // the next minibatch of the list
      CuMatrix<BaseFloat> cu_in(cu_in_orig[mi]); computer.AcceptInput("input", &cu_in); computer.Run();
      const CuMatrixBase<BaseFloat> &nnet_output = computer.GetOutput("output");
      CuMatrix<BaseFloat> nnet_output_deriv(nnet_output.NumRows(), nnet_output.NumCols(), kUndefined);
      BaseFloat objf = 0, l2_term = 0, weight = 0;
#ifdef K3_ADAPTER
      k3_chain_training_opts o = {fo[1], 0.0f, fo[0], 0};
      if (k3_chain_objf_and_deriv(kden, ksups[mi], &o, nnet_output.Data(), nnet_output.Stride(), nnet_output_deriv.Data(), nnet_output_deriv.Stride(), NULL, 0,
          &objf, &l2_term, &weight, k3_adapter_stream()) != K3_OK) KALDI_ERR << k3_last_error();
#else
      chain::ChainTrainingOptions opts; opts.leaky_hmm_coefficient = fo[0]; opts.l2_regularize = fo[1]; opts.out_of_range_regularize = 0.0;
      chain::ComputeChainObjfAndDeriv(opts, den_graph, supervisions[mi], nnet_output, &objf, &l2_term, &weight, &nnet_output_deriv, NULL);
#endif
      srand(2 * iter + 2);
      computer.AcceptInput("output", &nnet_output_deriv); computer.Run();
#ifdef K3_ADAPTER
      if (comm) {      // data-parallel ranks (one process per GPU, each with its share of the minibatch): the ranks' parameter changes are summed over RCCL / xGMI and averaged --
        // what the reference's recipes do with the jobs' models after every iteration (egs/wsj/s5/steps/libs/nnet3/train/chain_objf/acoustic_model.py:121,238),
        // here inside the iteration
        // (nnet-utils.h:148 works on host vectors: one bucket for the whole model)
        Vector<BaseFloat> flat_host(NumParameters(*delta_nnet), kUndefined);
        VectorizeNnet(*delta_nnet, &flat_host);
        CuVector<BaseFloat> flat(flat_host);
        if (k3_comm_allreduce_f32(comm, flat.Data(), flat.Dim(), k3_adapter_stream()) != K3_OK) KALDI_ERR << k3_last_error();
        flat.Scale(1.0 / world); flat.CopyToVec(&flat_host); UnVectorizeNnet(flat_host, delta_nnet);
      }
#endif
      ApplyL2Regularization(nnet, B * l2_regularize_factor, delta_nnet);      // GetNumNvalues(eg.inputs, false) = the number of sequences
      const bool success = UpdateNnetWithMaxChange(*delta_nnet, max_param_change, 1.0, 1.0 - momentum, &nnet, &max_change_stats);
      ScaleBatchnormStats(batchnorm_stats_scale, &nnet);
      ConstrainOrthonormal(&nnet);
      ScaleNnet(success ? momentum : 0.0, delta_nnet);
      objfs(3 * iter) = objf; objfs(3 * iter + 1) = l2_term; objfs(3 * iter + 2) = weight;
#ifdef K3_ADAPTER
      KALDI_LOG << "iteration " << iter << ": " << (double)k3_mat_gemm_flops(1) * 1.0e-9 <<
          " GFLOP in matrix products (2MNK over the AddMatMat calls of the forward, backward and update)";
#endif
      KALDI_LOG << "iteration " << iter << ": LF-MMI objf per frame " << objf / weight << " (+ l2 " << l2_term / weight << ") over " << weight << " frames; "
          << iter_timer.Elapsed() * 1000.0 << " ms";
    }
    max_change_stats.Print(nnet);
#ifdef K3_ADAPTER
    for (k3_chain_supervision *k : ksups) k3_chain_supervision_destroy(k);
    k3_chain_den_destroy(kden);
    if (comm) k3_comm_destroy(comm);
#endif
    delete delta_nnet;
    { SubVector<BaseFloat> pv(objfs, 3 * num_iters, objfs.Dim() - 3 * num_iters); VectorizeNnet(nnet, &pv); }
    WriteKaldiObject(nnet, po.GetArg(8), true); WriteKaldiObject(objfs, po.GetArg(9), true);
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what(); return -1; }
}
