#!/bin/bash
# tools/build_variant.sh <name> [-D...]: a profiling / experiment build of libk3hip.so -> build/libk3hip_<name>.so (the decoder's translation units compiled with the given defines,
# the other objects taken from kaldi_amd/csrc as built by make).  Select it with K3HIP_LIB=build/libk3hip_<name>.so (kaldi_amd/lib.py announces it; bench.py refuses it).
set -e
name=$1; shift; R=$(cd $(dirname $0)/.. && pwd); mkdir -p $R/build/$name; make -s -C $R/kaldi_amd/csrc
for f in k3_decoder k3_decoder_lit; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -I$R/kaldi_amd/csrc -c $R/kaldi_amd/csrc/$f.hip -o $R/build/$name/$f.o & done; wait
others=$(ls $R/kaldi_amd/csrc/*.o | grep -v -E "/k3_decoder(_lit)?\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/libk3hip_$name.so $R/build/$name/*.o $others -ldl
echo built $R/build/libk3hip_$name.so
