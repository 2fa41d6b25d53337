// tests/adapter/nnet3_two_threads.cc -- the adapter under a multi-threaded caller (SURVEY 8b "threading"; cudamatrix/cu-device.cc:112-124: the reference gives every host thread
// cudaStreamPerThread so that the threads of batched-threaded-* programs do not share one queue).
//   nnet3-two-threads <raw-nnet3-in> <threads> <iterations> <num-sequences> <frames-per-sequence> <frame-subsampling-factor>
// Every thread owns an NnetComputer per iteration over the SAME compiled computation and the same Nnet (read-only), its own random input and output derivative, and runs the
// reference's training-mode forward + backward (the calls of tests/adapter/nnet3_train_grad.cc) `iterations` times while the other threads do the same -- CuMatrix allocation and
// release, AddMatMat, CopyRows / AddRows through the index cache, reductions, host read-backs all interleave.  Before that, the same work is done by ONE thread; the program
// checks that every concurrent run reproduces its thread's single-threaded output and gradient BIT FOR BIT (same kernels, same order within a thread: any difference is a race --
// a buffer recycled across threads too early, a shared workspace, an index array overwritten under a queued kernel).  Prints "two-threads ok ..." and exits 0, or the first mismatch and 1.
#include <atomic>
#include <thread>
#include "base/kaldi-common.h"
#include "util/common-utils.h"
#include "nnet3/nnet-nnet.h"
#include "nnet3/nnet-utils.h"
#include "nnet3/nnet-optimize.h"
#include "nnet3/nnet-compute.h"
int main(int argc, char *argv[]) {
  try {
    using namespace kaldi; using namespace kaldi::nnet3;
    ParseOptions po("nnet3-two-threads <raw-nnet3-in> <threads> <iterations> <num-sequences> <frames-per-sequence> <frame-subsampling-factor>");
    po.Read(argc, argv);
    if (po.NumArgs() != 6) { po.PrintUsage(); return 1; }
    Nnet nnet; ReadKaldiObject(po.GetArg(1), &nnet);
    int32 NT, IT, B, T, s;
    if (!ConvertStringToInteger(po.GetArg(2), &NT) || !ConvertStringToInteger(po.GetArg(3), &IT) || !ConvertStringToInteger(po.GetArg(4), &B) || !ConvertStringToInteger(po.GetArg(5), &T) ||
        !ConvertStringToInteger(po.GetArg(6), &s)) KALDI_ERR << "bad integer argument";
    SetBatchnormTestMode(false, &nnet); SetDropoutTestMode(false, &nnet);
    int32 left, right; ComputeSimpleNnetContext(nnet, &left, &right);
    ComputationRequest request; request.need_model_derivative = true; request.store_component_stats = false;
    IoSpecification in; in.name = "input"; in.has_deriv = false;
    for (int32 t = -left; t <= (T - 1) * s + right; t++) for (int32 n = 0; n < B; n++) in.indexes.push_back(Index(n, t));
    IoSpecification out; out.name = "output"; out.has_deriv = true;
    for (int32 f = 0; f < T; f++) for (int32 n = 0; n < B; n++) out.indexes.push_back(Index(n, f * s));
    request.inputs.push_back(in); request.outputs.push_back(out);
    NnetOptimizeOptions optimize_opts; CachingOptimizingCompilerOptions compiler_opts;
    CachingOptimizingCompiler compiler(nnet, optimize_opts, compiler_opts);
    std::shared_ptr<const NnetComputation> computation = compiler.Compile(request);
    const int32 in_rows = (int32)in.indexes.size(), in_dim = nnet.InputDim("input"), out_rows = B * T, out_dim = nnet.OutputDim("output");
    // per-thread data (host side; a small deterministic generator: rand() is not thread safe)
    std::vector<Matrix<BaseFloat>> input(NT), deriv(NT), want_out(NT); std::vector<Vector<BaseFloat>> want_grad(NT);
    for (int32 k = 0; k < NT; k++) {
      uint64_t x = 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);
      auto next = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (BaseFloat)((double)(x >> 11) / 9007199254740992.0 - 0.5); };
      input[k].Resize(in_rows, in_dim); for (int32 r = 0; r < in_rows; r++) for (int32 c = 0; c < in_dim; c++) input[k](r, c) = 2.0f * next() + 16.0f;
      deriv[k].Resize(out_rows, out_dim); for (int32 r = 0; r < out_rows; r++) for (int32 c = 0; c < out_dim; c++) deriv[k](r, c) = 0.01f * next();
    }
    auto one_pass = [&](int32 k, Matrix<BaseFloat> *output, Vector<BaseFloat> *grad) {
      Nnet deriv_nnet(nnet); ScaleNnet(0.0, &deriv_nnet); SetNnetAsGradient(&deriv_nnet);
      NnetComputeOptions compute_opts; NnetComputer computer(compute_opts, *computation, nnet, &deriv_nnet);
      CuMatrix<BaseFloat> cu_in(input[k]); computer.AcceptInput("input", &cu_in);
      computer.Run();
      output->Resize(out_rows, out_dim); computer.GetOutput("output").CopyToMat(output);
      CuMatrix<BaseFloat> cu_deriv(deriv[k]); computer.AcceptInput("output", &cu_deriv);
      computer.Run();
      grad->Resize(NumParameters(deriv_nnet)); VectorizeNnet(deriv_nnet, grad);
    };
    for (int32 k = 0; k < NT; k++) one_pass(k, &want_out[k], &want_grad[k]);      // single-threaded reference results
    std::atomic<int> bad(0); std::vector<std::string> msg(NT);
    auto worker = [&](int32 k) {
      try {
        for (int32 it = 0; it < IT && bad.load() == 0; it++) {
          Matrix<BaseFloat> o; Vector<BaseFloat> g; one_pass(k, &o, &g);
          if (!(o.NumRows() == want_out[k].NumRows() && memcmp(o.Data(), want_out[k].Data(), sizeof(BaseFloat) * (size_t)o.NumRows() * o.Stride()) == 0) ||
              memcmp(g.Data(), want_grad[k].Data(), sizeof(BaseFloat) * (size_t)g.Dim()) != 0) {
            Matrix<BaseFloat> d(o); d.AddMat(-1.0, want_out[k]); Vector<BaseFloat> dg(g); dg.AddVec(-1.0, want_grad[k]);
            std::ostringstream os; os << "thread " << k << " iteration " << it << ": output differs by " << d.LargestAbsElem() << ", gradient by " << dg.Norm(2.0) << " of " << want_grad[k].Norm(2.0);
            msg[k] = os.str(); bad.fetch_add(1); return;
          }
        }
      } catch (const std::exception &e) { msg[k] = std::string("thread ") + std::to_string(k) + ": " + e.what(); bad.fetch_add(1); }
    };
    std::vector<std::thread> th; for (int32 k = 0; k < NT; k++) th.emplace_back(worker, k);
    for (auto &t : th) t.join();
    if (bad.load()) { for (auto &m : msg) if (!m.empty()) std::cerr << "two-threads MISMATCH: " << m << "\n"; return 1; }
    // ---- the index cache under churn: every thread keeps making NEW index arrays -- freed and reallocated, so host addresses come back with other content (the
    // cache must replace the copy without pulling it from under another thread's queued kernel), 8 MB each and more of them than the cache's cap (least-recently-used copies are
    // retired while other threads hold theirs) -- and checks every CopyRows against the host.
    {
      const int32 R = 1 << 21, SRC = 4099, ROUNDS = 24;      // 2 M row indexes per array (8 MB), 24 arrays per thread: 4 threads x 192 MB > the 256 MB cap
      Matrix<BaseFloat> src_h(SRC, 2); for (int32 r = 0; r < SRC; r++) { src_h(r, 0) = (BaseFloat)r; src_h(r, 1) = (BaseFloat)(-r); }
      CuMatrix<BaseFloat> src(src_h);
      std::atomic<int> bad2(0); std::vector<std::string> msg2(NT);
      auto churn = [&](int32 k) {
        try {
          uint64_t x = 0xD1B54A32D192ED03ull * (uint64_t)(k + 7);
          for (int32 it = 0; it < ROUNDS && bad2.load() == 0; it++) {
            std::vector<int32> idx(R);
            for (int32 i = 0; i < R; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; idx[i] = (x >> 40) % 17 == 0 ? -1 : (int32)((x >> 20) % SRC); }
            CuArray<int32> cu_idx(idx);
            CuMatrix<BaseFloat> dst(R, 2, kUndefined); dst.CopyRows(src, cu_idx);
            Matrix<BaseFloat> got(dst);
            for (int32 i = 0; i < R; i += 997) {
              const BaseFloat w0 = idx[i] < 0 ? 0.0f : (BaseFloat)idx[i], w1 = idx[i] < 0 ? 0.0f : (BaseFloat)(-idx[i]);
              if (got(i, 0) != w0 || got(i, 1) != w1) {
                std::ostringstream os; os << "thread " << k << " array " << it << " row " << i << ": CopyRows gave (" << got(i, 0) << ", " << got(i, 1) << ") for index " << idx[i];
                msg2[k] = os.str(); bad2.fetch_add(1); return;
              }
            }
          }
        } catch (const std::exception &e) { msg2[k] = std::string("thread ") + std::to_string(k) + ": " + e.what(); bad2.fetch_add(1); }
      };
      std::vector<std::thread> th2; for (int32 k = 0; k < NT; k++) th2.emplace_back(churn, k);
      for (auto &t : th2) t.join();
      if (bad2.load()) { for (auto &m : msg2) if (!m.empty()) std::cerr << "two-threads INDEX CACHE MISMATCH: " << m << "\n"; return 1; }
    }
    std::cout << "two-threads ok: " << NT << " threads x " << IT << " iterations of forward + backward (" << B << " sequences x " << T << " output frames, " << want_grad[0].Dim()
              << " parameters) reproduce the single-threaded outputs and gradients bit for bit; index cache churn ok\n";
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what(); return -1; }
}
