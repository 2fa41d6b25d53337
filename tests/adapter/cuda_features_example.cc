// tests/adapter/cuda_features_example.cc -- a caller of kaldi::CudaSpectralFeatures written against the REFERENCE's signatures (cudafeat/feature-spectral-cuda.h:70-107,
// the way cudafeatbin/compute-fbank-feats-cuda.cc uses the class) and compiled against include/k3_cuda_features.h + the reference's own feat / cudamatrix / util headers:
//   cuda-features-example [--config / FbankOptions or MfccOptions flags] <fbank|mfcc> <wav-rspecifier> <feats-wspecifier> [vtln-warp]
#include <cstring>
#include "base/kaldi-common.h"
#include "util/common-utils.h"
#include "feat/wave-reader.h"
#include "k3_cuda_features.h"
int main(int argc, char *argv[]) {
  try {
    using namespace kaldi;
    ParseOptions po("cuda-features-example [options] <fbank|mfcc> <wav-rspecifier> <feats-wspecifier> [vtln-warp]");
    bool mfcc = false; for (int i = 1; i < argc; i++) if (strncmp(argv[i], "--", 2) != 0) { mfcc = std::string(argv[i]) == "mfcc"; break; }      // the first positional argument picks the option class
    FbankOptions fbank_opts; MfccOptions mfcc_opts;
    if (mfcc) mfcc_opts.Register(&po); else fbank_opts.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() < 3 || po.NumArgs() > 4) { po.PrintUsage(); return 1; }
    BaseFloat vtln_warp = 1.0; if (po.NumArgs() == 4 && !ConvertStringToReal(po.GetArg(4), &vtln_warp)) KALDI_ERR << "bad vtln warp";
    CudaSpectralFeatureOptions opts = mfcc ? CudaSpectralFeatureOptions(mfcc_opts) : CudaSpectralFeatureOptions(fbank_opts);
    CudaSpectralFeatures feats(opts);
    SequentialTableReader<WaveHolder> reader(po.GetArg(2)); BaseFloatMatrixWriter writer(po.GetArg(3));
    int32 n = 0;
    for (; !reader.Done(); reader.Next(), n++) {
      const WaveData &wave = reader.Value(); SubVector<BaseFloat> waveform(wave.Data(), 0);
      CuVector<BaseFloat> cu_wave(waveform); CuMatrix<BaseFloat> cu_features;
      feats.ComputeFeatures(cu_wave, wave.SampFreq(), vtln_warp, &cu_features);
      Matrix<BaseFloat> features(cu_features.NumRows(), cu_features.NumCols()); features.CopyFromMat(cu_features);
      writer.Write(reader.Key(), features);
    }
    KALDI_LOG << "Done " << n << " utterances, dim " << feats.Dim();
    return n ? 0 : 1;
  } catch (const std::exception &e) { std::cerr << e.what(); return -1; }
}
