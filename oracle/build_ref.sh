#!/usr/bin/env bash
# Build the REFERENCE's own CPU programs (compute-fbank-feats, compute-mfcc-feats, apply-cmvn,
# compute-cmvn-stats, nnet3-init, nnet3-info, nnet3-compute, nnet3-copy) straight from the sources
# where they lie under /root/reference/src, into oracle/_ref/ (git-ignored, travels to the GPU box).
#
# TEST INFRASTRUCTURE ONLY: these binaries pin the restated oracle (oracle/*.c, oracle/*.py) and
# may serve as the "reference" CPU baseline in bench.py.  Nothing in kaldi_amd/ may call them.
#
# No reference source is copied: we compile in place with g++ (the reference's own build system --
# configure + per-dir Makefiles + OpenFst -- is not run).  Two generated headers stand in for
# pieces the reference's build would generate/download:
#   base/version.h   (normally written by base/get_version.sh)
#   fst/fst-decl.h   (OpenFst forward declarations only; hmm/transition-model.h:26 includes it)
# OpenFst 1.8.4 is not vendored, so the reference's own build of src/decoder, src/lat, src/fstext is impossible here.  But
# decoder/lattice-faster-decoder.{h,cc} only touch a small part of OpenFst's public interface (a graph's read interface, a vector
# FST to fill, arc iterators, a memory pool): third_party/minifst/ is a stand-in for exactly that part, and with it the
# reference's decoder source compiles UNMODIFIED into oracle/_ref/bin/ref-lattice-decoder (driver: ref_tools/ref_lattice_decoder.cc).
# That binary pins the restated decoder oracle (oracle/lattice_faster_oracle.cc, tests/test_oracle_decoder.py).  With TopSort / ArcSort /
# Invert / Connect added to the stand-in, lat/determinize-lattice-pruned.cc compiles unmodified too: oracle/_ref/bin/
# ref-lattice-determinize pins the host-side determinizer of the drop-in programs (kaldi_amd/host/k3_lattice.cc, tests/test_lattice_det.py).
set -euo pipefail
REF=${KALDI_REFERENCE:-/root/reference}
R=$REF/src
HERE=$(cd "$(dirname "$0")" && pwd)
W=$HERE/_ref
if [ ! -d "$R" ]; then echo "build_ref: $R absent (GPU box?) - using prebuilt files in $W"; exit 0; fi
mkdir -p $W/inc/base $W/stub/fst $W/obj $W/mkl $W/bin
printf '#define KALDI_VERSION "5.5-oracle"\n#define KALDI_GIT_HEAD "oracle"\n' > $W/inc/base/version.h
cat > $W/stub/fst/fst-decl.h <<'EOS'
#ifndef ORACLE_FST_DECL_STUB_H_
#define ORACLE_FST_DECL_STUB_H_
namespace fst {
template <class A> class Fst; template <class A> class VectorFst;
template <class W> class ArcTpl; template <class T> class TropicalWeightTpl;
using StdArc = ArcTpl<TropicalWeightTpl<float>>;
using StdFst = Fst<StdArc>; using StdVectorFst = VectorFst<StdArc>;
}
#endif
EOS
MKL=/opt/conda/lib/libmkl_rt.so
FLAGS="-std=c++17 -O2 -DNDEBUG -w -I $W/inc -I $W/stub -I $R -I $REF/tools/CLAPACK -DHAVE_CLAPACK -DOPENFST_VER=10804 -DHAVE_EXECINFO_H=1 -DHAVE_CXXABI_H -DHAVE_CUDA=0 -pthread"

list_sources() {
  for d in base matrix util feat cudamatrix tree itf; do
    ls $R/$d/*.cc 2>/dev/null | grep -v -e '-test\.cc$' -e 'tree/tree-renderer\.cc' || true
  done
  echo $R/transform/cmvn.cc
  echo $R/hmm/transition-model.cc; echo $R/hmm/hmm-topology.cc
  ls $R/nnet3/*.cc | grep -v -e '-test\.cc$' -e 'nnet-example' -e 'nnet-chain-' -e 'nnet-discriminative-' \
      -e 'discriminative-' -e 'nnet-batch-compute\.cc'
}
compile_one() {
  src=$1; o=$W/obj/$(basename $(dirname $src))_$(basename ${src%.cc}).o
  if [ ! -f $o ] || [ $src -nt $o ]; then g++ $FLAGS -c $src -o $o || { echo "FAILED $src"; exit 1; }; fi
}
export -f compile_one; export W FLAGS
list_sources | xargs -P ${JOBS:-8} -I{} bash -c 'compile_one {}'
rm -f $W/libref.a; ar rcs $W/libref.a $W/obj/*.o
link() { # name src
  g++ $FLAGS $2 $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/$1
}
link compute-fbank-feats $R/featbin/compute-fbank-feats.cc
link compute-mfcc-feats  $R/featbin/compute-mfcc-feats.cc
link apply-cmvn          $R/featbin/apply-cmvn.cc
link compute-cmvn-stats  $R/featbin/compute-cmvn-stats.cc
link copy-feats          $R/featbin/copy-feats.cc
link apply-cmvn-online  $R/online2bin/apply-cmvn-online.cc
link nnet3-init          $R/nnet3bin/nnet3-init.cc
link nnet3-info          $R/nnet3bin/nnet3-info.cc
link nnet3-compute       $R/nnet3bin/nnet3-compute.cc
link nnet3-copy          $R/nnet3bin/nnet3-copy.cc
link nnet3-am-info       $R/nnet3bin/nnet3-am-info.cc
link nnet3-am-copy       $R/nnet3bin/nnet3-am-copy.cc
# the reference's training computation of one minibatch (forward in training mode + Backprop of every component, gradient nnet) on its CPU matrices: the oracle of
# the same program linked against the MI355X CuMatrix adapter (kaldi_amd/adapter/_build/nnet3-train-grad); source shared with the adapter build
link ref-nnet3-train-grad $HERE/../tests/adapter/nnet3_train_grad.cc
link dump-tid2pdf        $HERE/ref_tools/dump_tid2pdf.cc
link dump-tidinfo        $HERE/ref_tools/dump_tidinfo.cc
# the reference's LatticeFasterDecoder over the OpenFst stand-in (include path: minifst first, so that fst/*.h, fstext/fstext-lib.h,
# lat/*.h and decoder/grammar-fst.h resolve to the stand-in; every other header, and the .cc itself, is the reference's)
MF="-std=c++17 -O2 -DNDEBUG -w -I $HERE/../third_party/minifst -I $W/inc -I $R -I $REF/tools/CLAPACK -DHAVE_CLAPACK -DOPENFST_VER=10804 -DHAVE_EXECINFO_H=1 -DHAVE_CXXABI_H -DHAVE_CUDA=0 -pthread"
mkdir -p $W/obj_minifst
g++ $MF -c $R/decoder/lattice-faster-decoder.cc -o $W/obj_minifst/lattice-faster-decoder.o
g++ $MF $HERE/ref_tools/ref_lattice_decoder.cc $W/obj_minifst/lattice-faster-decoder.o $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/ref-lattice-decoder
# the reference's lattice determinization (lat/determinize-lattice-pruned.cc, unmodified) over the same stand-in: pins kaldi_amd/host/k3_lattice.cc
for f in determinize-lattice-pruned push-lattice minimize-lattice; do g++ $MF -c $R/lat/$f.cc -o $W/obj_minifst/$f.o; done
g++ $MF $HERE/ref_tools/ref_lattice_determinize.cc $W/obj_minifst/determinize-lattice-pruned.o $W/obj_minifst/push-lattice.o $W/obj_minifst/minimize-lattice.o $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/ref-lattice-determinize
# ConvertLattice + Factor (fstext/lattice-utils-inl.h, fstext/factor-inl.h: header-only templates of the reference) over the stand-in
g++ $MF $HERE/ref_tools/ref_convert_lattice.cc $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/ref-convert-lattice
# the reference's word-level MBR decoder (lat/sausages.cc, unmodified) over the same stand-in: pins kaldi_amd/host/k3_mbr.cc (the CTM output of the lattice post-processor)
g++ $MF -c $R/lat/sausages.cc -o $W/obj_minifst/sausages.o
g++ $MF $HERE/ref_tools/ref_mbr.cc $W/obj_minifst/sausages.o $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/ref-mbr
# the reference's word aligner (lat/word-align-lattice.cc, unmodified) over the same stand-in, followed by its MBR decoder -- the two steps of LatticePostprocessor::GetCTM
# with --word-boundary-rxfilename: pins the restatement in kaldi_amd/host/k3_mbr.cc (tests/test_word_align.py)
g++ $MF -c $R/lat/word-align-lattice.cc -o $W/obj_minifst/word-align-lattice.o
g++ $MF $HERE/ref_tools/ref_word_align.cc $W/obj_minifst/word-align-lattice.o $W/obj_minifst/sausages.o $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/ref-word-align
# i-vector extraction (the next row to build): the reference's gmm/, ivector/ and online2/online-ivector-feature.cc with
# the programs that create a small extractor from features (gmm-global-init-from-feats -> gmm-global-to-fgmm -> ivector-extractor-init)
# and the one that is the oracle for the GPU path to come (ivector-extract-online2 = OnlineIvectorFeature, what
# cudafeat/online-ivector-feature-cuda.cc re-implements); *-copy dump the models as text for the host readers' tests.
mkdir -p $W/obj_iv
compile_iv() { src=$1; o=$W/obj_iv/$(basename $(dirname $src))_$(basename ${src%.cc}).o; if [ ! -f $o ] || [ $src -nt $o ]; then g++ $MF -c $src -o $o || { echo "FAILED $src"; exit 1; }; fi; }
export -f compile_iv; export MF
{ ls $R/gmm/*.cc | grep -v -e '-test\.cc$'; echo $R/ivector/ivector-extractor.cc; echo $R/online2/online-ivector-feature.cc; echo $R/hmm/posterior.cc; } | xargs -P ${JOBS:-8} -I{} bash -c 'compile_iv {}'
rm -f $W/libref_iv.a; ar rcs $W/libref_iv.a $W/obj_iv/*.o
link_iv() { g++ $MF $2 $W/libref_iv.a $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/$1; }
link_iv ivector-extract-online2     $R/online2bin/ivector-extract-online2.cc
link_iv gmm-global-init-from-feats  $R/gmmbin/gmm-global-init-from-feats.cc
link_iv gmm-global-to-fgmm          $R/gmmbin/gmm-global-to-fgmm.cc
link_iv gmm-global-copy             $R/gmmbin/gmm-global-copy.cc
link_iv ivector-extractor-init      $R/ivectorbin/ivector-extractor-init.cc
link_iv ivector-extractor-copy      $R/ivectorbin/ivector-extractor-copy.cc
# the reference's LF-MMI denominator (chain/chain-den-graph.cc + chain/chain-denominator.cc, CPU path) over the OpenFst stand-in.  Both files compile
# UNMODIFIED; the graph-compilation half of chain-den-graph.cc (CreateDenominatorFst ...: real OpenFst algorithms) is never called and its callees are
# declarations only (minifst/hmm/hmm-utils.h, minifst/fstext/{deterministic-fst,push-special}.h) -- the link line ignores unresolved symbols like the others.
# chain/chain-numerator.cc and chain/chain-training.cc (ComputeChainObjfAndDeriv) compile unmodified too: ref-chain-objf is the oracle of the whole LF-MMI objective.
mkdir -p $W/obj_chain
g++ $MF -c $R/chain/chain-den-graph.cc -o $W/obj_chain/chain-den-graph.o
g++ $MF -c $R/chain/chain-denominator.cc -o $W/obj_chain/chain-denominator.o
g++ $MF -c $R/chain/chain-numerator.cc -o $W/obj_chain/chain-numerator.o
g++ $MF -c $R/chain/chain-training.cc -o $W/obj_chain/chain-training.o
g++ $MF -c $R/chain/chain-generic-numerator.cc -o $W/obj_chain/chain-generic-numerator.o      # the end-to-end (flat-start) numerator: compiles unmodified too
g++ $MF $HERE/ref_tools/ref_chain_objf.cc $W/obj_chain/chain-den-graph.o $W/obj_chain/chain-denominator.o $W/obj_chain/chain-numerator.o $W/obj_chain/chain-generic-numerator.o $W/obj_chain/chain-training.o $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/ref-chain-objf
# the whole chain gradient on the reference's CPU code: NnetComputer forward (training mode) -> ComputeChainObjfAndDeriv -> backward (source shared with the adapter build)
g++ $MF $HERE/../tests/adapter/nnet3_chain_grad.cc $W/obj_chain/chain-den-graph.o $W/obj_chain/chain-denominator.o $W/obj_chain/chain-numerator.o $W/obj_chain/chain-generic-numerator.o $W/obj_chain/chain-training.o $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/ref-nnet3-chain-grad
# N iterations of chain TRAINING (NnetChainTrainer::TrainInternal's sequence: natural-gradient update, max-change, batch-norm stats, orthonormal constraint) on the CPU (source shared with the adapter build)
g++ $MF $HERE/../kaldi_amd/adapter/nnet3-chain-train.cc $W/obj_chain/chain-den-graph.o $W/obj_chain/chain-denominator.o $W/obj_chain/chain-numerator.o $W/obj_chain/chain-generic-numerator.o $W/obj_chain/chain-training.o $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/ref-nnet3-chain-train
g++ $MF $HERE/ref_tools/ref_chain_den.cc $W/obj_chain/chain-den-graph.o $W/obj_chain/chain-denominator.o $W/libref.a $MKL -ldl -lm -Wl,--unresolved-symbols=ignore-all -Wl,-rpath,$W/mkl -o $W/bin/ref-chain-den
for f in /opt/conda/lib/libmkl_{rt,core,intel_lp64,sequential,gnu_thread,intel_thread,avx2,avx512,def,mc3,vml_avx2,vml_avx512,vml_def}.so.1; do
  [ -e $f ] && ln -sf $f $W/mkl/ || true; done
cat > $W/env.sh <<EOS
export LD_LIBRARY_PATH=$W/mkl\${LD_LIBRARY_PATH:+:\$LD_LIBRARY_PATH}
export MKL_THREADING_LAYER=SEQUENTIAL
export PATH=$W/bin:\$PATH
EOS
strip $W/bin/* 2>/dev/null || true      # the binaries travel to the GPU box with every gpurun call: keep them small
echo "build_ref: ok -> $W/bin"
