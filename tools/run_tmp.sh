K3HIP_LIB=build/libk3hip_framecyc.so python tools/prof_frames.py 512 2>&1 | grep -v amdgpu.ids
for v in qbig qmid; do echo "== $v"; K3HIP_LIB=build/libk3hip_$v.so K3_LIT_PROF_Q=1 python tools/prof_literal.py 512 2>&1 | grep -v amdgpu.ids | grep -v "^default"; done
for v in fbig fmid; do echo "== $v"; K3HIP_LIB=build/libk3hip_$v.so K3_LIT_PROF_FINE=1 python tools/prof_literal.py 512 2>&1 | grep -v amdgpu.ids | grep -v "^default"; done
