// tests/adapter/nnet3_train_grad.cc -- the reference's OWN training computation for one minibatch, written against the reference's nnet3 API (the calls
// nnet3/nnet-chain-training.cc:136-206 makes: request with need_model_derivative, CachingOptimizingCompiler, NnetComputer with a gradient nnet, AcceptInput /
// Run / GetOutput / AcceptInput(output derivative) / Run): forward in TRAINING mode (BatchNorm on batch statistics), then Backprop of every component, giving the
// gradient of <output, output_deriv> w.r.t. every parameter.
//   nnet3-train-grad <raw-nnet3-in> <num-sequences> <frames-per-sequence(output)> <frame-subsampling-factor> <input-matrix-in> <output-deriv-matrix-in> <output-matrix-out> <gradient-vector-out>
// input rows: t-major, sequence-minor, t = -left_context .. (frames - 1) * subsampling + right_context; output / derivative rows: row = frame * num_sequences + sequence (the layout
// of a chain minibatch).  Linked twice from this one source: against the reference's CPU cudamatrix (oracle/_ref/bin/ref-nnet3-train-grad: the oracle) and against
// kaldi_amd/adapter/cu-k3.cc (kaldi_amd/adapter/_build/nnet3-train-grad: every matrix operation of the reference's NnetComputer runs on the MI355X).
#include "base/kaldi-common.h"
#include "util/common-utils.h"
#include "nnet3/nnet-nnet.h"
#include "nnet3/nnet-utils.h"
#include "nnet3/nnet-optimize.h"
#include "nnet3/nnet-compute.h"
int main(int argc, char *argv[]) {
  try {
    using namespace kaldi; using namespace kaldi::nnet3;
    ParseOptions po("nnet3-train-grad <raw-nnet3-in> <num-sequences> <frames-per-sequence> <frame-subsampling-factor> <input-matrix-in> <output-deriv-in> <output-out> <gradient-out>");
    po.Read(argc, argv);
    if (po.NumArgs() != 8) { po.PrintUsage(); return 1; }
    Nnet nnet; ReadKaldiObject(po.GetArg(1), &nnet);
    int32 B, T, s; if (!ConvertStringToInteger(po.GetArg(2), &B) || !ConvertStringToInteger(po.GetArg(3), &T) || !ConvertStringToInteger(po.GetArg(4), &s)) KALDI_ERR << "bad integer argument";
    SetBatchnormTestMode(false, &nnet); SetDropoutTestMode(false, &nnet);      // training mode (dropout proportions of the test models are 0)
    Nnet deriv_nnet(nnet); ScaleNnet(0.0, &deriv_nnet); SetNnetAsGradient(&deriv_nnet);      // nnet-chain-training.cc:69-75 / nnet-training.cc:62-66
    int32 left, right; ComputeSimpleNnetContext(nnet, &left, &right);
    ComputationRequest request; request.need_model_derivative = true; request.store_component_stats = false;
    IoSpecification in; in.name = "input"; in.has_deriv = false;
    for (int32 t = -left; t <= (T - 1) * s + right; t++) for (int32 n = 0; n < B; n++) in.indexes.push_back(Index(n, t));
    IoSpecification out; out.name = "output"; out.has_deriv = true;
    for (int32 f = 0; f < T; f++) for (int32 n = 0; n < B; n++) out.indexes.push_back(Index(n, f * s));
    request.inputs.push_back(in); request.outputs.push_back(out);
    Matrix<BaseFloat> input, output_deriv; ReadKaldiObject(po.GetArg(5), &input); ReadKaldiObject(po.GetArg(6), &output_deriv);
    if (input.NumRows() != (int32)in.indexes.size() || input.NumCols() != nnet.InputDim("input")) KALDI_ERR << "input matrix is " << input.NumRows() << " x " << input.NumCols() << ", expected " << in.indexes.size() << " x " << nnet.InputDim("input");
    if (output_deriv.NumRows() != B * T || output_deriv.NumCols() != nnet.OutputDim("output")) KALDI_ERR << "output derivative has the wrong size";
    NnetOptimizeOptions optimize_opts; CachingOptimizingCompilerOptions compiler_opts;
    CachingOptimizingCompiler compiler(nnet, optimize_opts, compiler_opts);
    std::shared_ptr<const NnetComputation> computation = compiler.Compile(request);
    NnetComputeOptions compute_opts; NnetComputer computer(compute_opts, *computation, nnet, &deriv_nnet);
    CuMatrix<BaseFloat> cu_in(input); computer.AcceptInput("input", &cu_in);
    computer.Run();
    Matrix<BaseFloat> output(computer.GetOutput("output")); WriteKaldiObject(output, po.GetArg(7), true);
    CuMatrix<BaseFloat> cu_deriv(output_deriv); computer.AcceptInput("output", &cu_deriv);
    computer.Run();
    Vector<BaseFloat> grad(NumParameters(deriv_nnet)); VectorizeNnet(deriv_nnet, &grad); WriteKaldiObject(grad, po.GetArg(8), true);
    KALDI_LOG << "forward (training mode) and backward over " << B << " sequences x " << T << " output frames; " << grad.Dim() << " parameters, |gradient| = " << grad.Norm(2.0);
    return 0;
  } catch (const std::exception &e) { std::cerr << e.what(); return -1; }
}
