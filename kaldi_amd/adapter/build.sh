#!/usr/bin/env bash
# Builds the reference's OWN nnet3 programs on top of the MI355X CuMatrix implementation of this directory:
#   kaldi_amd/adapter/_build/nnet3-compute      = /root/reference/src/nnet3bin/nnet3-compute.cc, unmodified
# linked from the reference's nnet3 / hmm / tree / matrix / util / base objects (compiled from the sources where they lie, HAVE_CUDA=0: not one
# CUDA or hipify header is involved) + kaldi_amd/adapter/cu-k3.cc in place of the reference's src/cudamatrix/*.cc + kaldi_amd/lib/libk3hip.so.
# Every CuMatrix / CuVector member the reference's objects reference and cu-k3.cc does not define gets a trampoline (generated below from the
# linker's list of undefined symbols) that raises "not implemented by the k3 adapter" with the member's name: nothing falls back silently.
# Needs /root/reference (sources to compile against) -- on the GPU box the prebuilt files in _build/ are used (they travel with the snapshot).
set -euo pipefail
REF=${KALDI_REFERENCE:-/root/reference}; R=$REF/src
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd); B=$HERE/_build
if [ ! -d "$R" ]; then echo "adapter/build.sh: $R absent (GPU box?) - using prebuilt files in $B"; exit 0; fi
# nothing to do when every program is newer than everything it is made from (the sources of this adapter and of its callers, the headers of include/ and of the host layer, the
# host layer's objects; the reference's sources are read-only): a no-op build is what the test suite calls a dozen times
STAMP=$B/.built
if [ -f $STAMP ] && [ -z "$(find $HERE/build.sh $HERE/*.cc $ROOT/tests/adapter/*.cc $ROOT/include/*.h $ROOT/kaldi_amd/host/*.h $ROOT/kaldi_amd/bin/k3_host.o $ROOT/kaldi_amd/bin/k3_lattice.o $ROOT/kaldi_amd/bin/k3_mbr.o \
      $ROOT/third_party/minifst -newer $STAMP 2>/dev/null | head -1)" ]; then exit 0; fi
rm -f $STAMP
mkdir -p $B/obj $B/inc/base $B/stub/fst $B/mkl
printf '#define KALDI_VERSION "5.5-k3"\n#define KALDI_GIT_HEAD "k3"\n' > $B/inc/base/version.h
cat > $B/stub/fst/fst-decl.h <<'EOS'
#ifndef K3_FST_DECL_STUB_H_
#define K3_FST_DECL_STUB_H_
namespace fst {   // forward declarations only (hmm/transition-model.h:26); OpenFst itself is not needed by the nnet3 forward path
template <class A> class Fst; template <class A> class VectorFst; template <class W> class ArcTpl; template <class T> class TropicalWeightTpl;
using StdArc = ArcTpl<TropicalWeightTpl<float>>; using StdFst = Fst<StdArc>; using StdVectorFst = VectorFst<StdArc>;
}
#endif
EOS
MKL=/opt/conda/lib/libmkl_rt.so
for f in /opt/conda/lib/libmkl_{rt,core,intel_lp64,sequential,gnu_thread,intel_thread,avx2,avx512,def,mc3,vml_avx2,vml_avx512,vml_def}.so.1 /opt/conda/lib/libmkl_rt.so; do [ -e $f ] && ln -sf $f $B/mkl/ || true; done
FLAGS="-std=c++17 -O2 -DNDEBUG -w -I $B/inc -I $B/stub -I $R -I $REF/tools/CLAPACK -DHAVE_CLAPACK -DOPENFST_VER=10804 -DHAVE_EXECINFO_H=1 -DHAVE_CXXABI_H -DHAVE_CUDA=0 -pthread"
list_sources() {
  for d in base matrix util tree itf feat; do ls $R/$d/*.cc 2>/dev/null | grep -v -e '-test\.cc$' -e 'tree/tree-renderer\.cc' || true; done
  echo $R/cudamatrix/cu-array.cc          # Int32Pair stream operators; the classes themselves come from cu-k3.cc
  echo $R/hmm/transition-model.cc; echo $R/hmm/hmm-topology.cc
  ls $R/nnet3/*.cc | grep -v -e '-test\.cc$' -e 'nnet-example' -e 'nnet-chain-' -e 'nnet-discriminative-' -e 'discriminative-' -e 'nnet-batch-compute\.cc'
}
compile_one() { src=$1; o=$B/obj/$(basename $(dirname $src))_$(basename ${src%.cc}).o; if [ ! -f $o ] || [ $src -nt $o ]; then g++ $FLAGS -c $src -o $o || { echo "FAILED $src"; exit 1; }; fi; }
export -f compile_one; export B FLAGS
list_sources | xargs -P ${JOBS:-8} -I{} bash -c 'compile_one {}'
rm -f $B/libkaldi-nocudamatrix.a; ar rcs $B/libkaldi-nocudamatrix.a $B/obj/*.o
g++ $FLAGS -I $ROOT/include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -c $HERE/cu-k3.cc -o $B/cu-k3.o
LIBS="$B/cu-k3.o $B/libkaldi-nocudamatrix.a $ROOT/kaldi_amd/lib/libk3hip.so -L/opt/rocm/lib -lamdhip64 $MKL -ldl -lm"
RPATH="-Wl,-rpath,\$ORIGIN/mkl -Wl,-rpath,\$ORIGIN/../../lib -Wl,-rpath,/opt/rocm/lib"
link_with_trampolines() { # name main.cc
  # pass 1: collect what is still undefined, write one trampoline per symbol, pass 2: link for real
  g++ $FLAGS $2 $LIBS -o /dev/null 2> $B/$1.undef.log || true
  grep -o "undefined reference to \`[^']*'" $B/$1.undef.log | sed "s/undefined reference to \`//; s/'\$//" | sort -u > $B/$1.undef.txt || true
  python3 - "$B/$1.undef.txt" "$B/$1.trampolines.s" <<'PY'
import subprocess, sys
names = [l.rstrip("\n") for l in open(sys.argv[1]) if l.strip()]
# the linker prints demangled names; get the mangled ones back from the objects' symbol tables
import glob, os
B = os.path.dirname(sys.argv[1])
und = subprocess.run("nm -u " + " ".join(glob.glob(B + "/obj/*.o") + glob.glob(B + "/obj_chain/*.o") + [B + "/cu-k3.o"]) + " | awk '{print $2}' | sort -u", shell=True, capture_output=True, text=True).stdout.split()
dem = subprocess.run(["c++filt"], input="\n".join(und), capture_output=True, text=True).stdout.split("\n")
want = set(names); out = []
for m, d in zip(und, dem):
    if d in want: out.append((m, d))
with open(sys.argv[2], "w") as f:
    f.write("# generated by kaldi_amd/adapter/build.sh: members of the reference's CuMatrix family that cu-k3.cc does not define\n\t.text\n")
    for i, (m, d) in enumerate(out):
        f.write(f"\t.globl {m}\n\t.type {m}, @function\n{m}:\n\tleaq .Lname{i}(%rip), %rdi\n\tjmp k3_adapter_unimplemented@PLT\n")
    f.write("\t.section .rodata\n")
    for i, (m, d) in enumerate(out): f.write(f'.Lname{i}:\t.asciz "{d}"\n')
    f.write('\t.section .note.GNU-stack,"",@progbits\n')
print(f"  {len(out)} trampolines for {len(names)} undefined members")
PY
  g++ $FLAGS $2 $B/$1.trampolines.s $LIBS $RPATH -o $B/$1
}
link_with_trampolines nnet3-compute $R/nnet3bin/nnet3-compute.cc
echo "built $B/nnet3-compute"
# the reference's training computation of one minibatch (NnetComputer forward in training mode + Backprop of every component into a gradient nnet) over the adapter
link_with_trampolines nnet3-train-grad $ROOT/tests/adapter/nnet3_train_grad.cc
echo "built $B/nnet3-train-grad"
# the same training computation on several host threads at once (per-thread streams, thread-aware memory pool, shared index cache): tests/adapter/nnet3_two_threads.cc
link_with_trampolines nnet3-two-threads $ROOT/tests/adapter/nnet3_two_threads.cc
echo "built $B/nnet3-two-threads"
# the gradient of the LF-MMI objective: the reference's NnetComputer over the adapter + k3_chain_objf_and_deriv on the adapter's device pointers
FLAGS_SAVE="$FLAGS"; FLAGS="$FLAGS -DK3_ADAPTER -I $ROOT/include"
link_with_trampolines nnet3-chain-grad $ROOT/tests/adapter/nnet3_chain_grad.cc
link_with_trampolines nnet3-chain-train $HERE/nnet3-chain-train.cc      # N training iterations: + natural gradient, max-change, batch-norm stats, orthonormal constraint
FLAGS="$FLAGS_SAVE"
echo "built $B/nnet3-chain-grad $B/nnet3-chain-train"
# The CudaFst / CudaDecoder adapter header (include/k3_cuda_decoder.h) compiled into a caller written against the reference's signatures.  The lattice
# types are the reference's own (fstext/lattice-weight.h, lat/kaldi-lattice.h names); OpenFst, which /root/reference does not vendor, is stood in for
# by the small OpenFst stand-in headers of third_party/minifst (include path only; no object code).
MF="-std=c++17 -O2 -DNDEBUG -w -I $ROOT/third_party/minifst -I $B/inc -I $R -I $REF/tools/CLAPACK -DHAVE_CLAPACK -DOPENFST_VER=10804 -DHAVE_EXECINFO_H=1 -DHAVE_CXXABI_H -DHAVE_CUDA=0 -pthread"
g++ $MF -I $ROOT/include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ $ROOT/tests/adapter/cuda_decoder_example.cc $B/obj/base_*.o $ROOT/kaldi_amd/lib/libk3hip.so -L/opt/rocm/lib -lamdhip64 -ldl -lm \
    -Wl,-rpath,\$ORIGIN/../../lib -Wl,-rpath,/opt/rocm/lib -o $B/cuda-decoder-example
echo "built $B/cuda-decoder-example"
# kaldi::cuda_decoder::BatchedThreadedNnet3CudaPipeline2 with the reference's constructor (fst::Fst<StdArc>, nnet3::AmNnetSimple, TransitionModel) and callback type (include/k3_batched_pipeline.h)
# in a caller that reads final.mdl with the reference's own readers; host layer objects from kaldi_amd/host (make -C kaldi_amd/host first)
FLAGS_SAVE2="$FLAGS"; FLAGS="$MF -I $ROOT/include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__"
link_with_trampolines cuda-pipeline-example "$ROOT/tests/adapter/cuda_pipeline_example.cc $ROOT/kaldi_amd/bin/k3_host.o $ROOT/kaldi_amd/bin/k3_lattice.o $ROOT/kaldi_amd/bin/k3_mbr.o"
link_with_trampolines cuda-online-pipeline-example "$ROOT/tests/adapter/cuda_online_pipeline_example.cc $ROOT/kaldi_amd/bin/k3_host.o $ROOT/kaldi_amd/bin/k3_lattice.o $ROOT/kaldi_amd/bin/k3_mbr.o"
FLAGS="$FLAGS_SAVE2"
echo "built $B/cuda-pipeline-example $B/cuda-online-pipeline-example"
# nnet3-chain-train with the REFERENCE's command line and example archives: chainbin/nnet3-chain-train.cc, nnet3/nnet-chain-training.cc (NnetChainTrainer), nnet3/nnet-chain-example.cc,
# nnet3/nnet-example.cc and nnet3/nnet-example-utils.cc compiled UNMODIFIED over the adapter; chain-k3.cc stands behind chain::ComputeChainObjfAndDeriv, chain::DenominatorGraph,
# chain::Supervision::Read and fst::ReadFstKaldi (k3_chain_* kernels, the host layer's FST reader); the chain library's other members resolve to loud trampolines
mkdir -p $B/obj_chain
MFA="$MF -DK3_ADAPTER -I $ROOT/include -I $ROOT/kaldi_amd/host -I /opt/rocm/include -D__HIP_PLATFORM_AMD__"
for f in nnet3/nnet-chain-training nnet3/nnet-chain-example nnet3/nnet-example nnet3/nnet-example-utils; do
  o=$B/obj_chain/$(basename $f).o; if [ ! -f $o ] || [ $R/$f.cc -nt $o ]; then g++ $MFA -c $R/$f.cc -o $o || { echo "FAILED $f"; exit 1; }; fi
done
g++ $MFA -c $HERE/chain-k3.cc -o $B/obj_chain/chain-k3.o
FLAGS_SAVE3="$FLAGS"; FLAGS="$MFA"
link_with_trampolines nnet3-chain-train-egs "$R/chainbin/nnet3-chain-train.cc $B/obj_chain/chain-k3.o $B/obj_chain/nnet-chain-training.o $B/obj_chain/nnet-chain-example.o $B/obj_chain/nnet-example.o $B/obj_chain/nnet-example-utils.o $ROOT/kaldi_amd/bin/k3_host.o $ROOT/kaldi_amd/bin/k3_lattice.o $ROOT/kaldi_amd/bin/k3_mbr.o"
# ... and the reference's chainbin/nnet3-chain-copy-egs.cc over the same objects: text <-> binary example archives (Supervision::Write of chain-k3.cc on the way out)
link_with_trampolines nnet3-chain-copy-egs "$R/chainbin/nnet3-chain-copy-egs.cc $B/obj_chain/chain-k3.o $B/obj_chain/nnet-chain-training.o $B/obj_chain/nnet-chain-example.o $B/obj_chain/nnet-example.o $B/obj_chain/nnet-example-utils.o $ROOT/kaldi_amd/bin/k3_host.o $ROOT/kaldi_amd/bin/k3_lattice.o $ROOT/kaldi_amd/bin/k3_mbr.o"
FLAGS="$FLAGS_SAVE3"
echo "built $B/nnet3-chain-train-egs $B/nnet3-chain-copy-egs"
# kaldi::CudaSpectralFeatures with the reference's signatures (include/k3_cuda_features.h) in a caller that uses the reference's own WaveData / table IO / option classes
FLAGS="$FLAGS -I $ROOT/include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__"
link_with_trampolines cuda-features-example $ROOT/tests/adapter/cuda_features_example.cc
echo "built $B/cuda-features-example"
touch $STAMP
