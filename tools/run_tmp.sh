for rep in 1 2 3; do
for v in base oz o2; do
echo -n "$v "; K3HIP_LIB=build/libk3hip_$v.so K3_PROF_PATHS=1 python tools/prof_literal.py 512 2>&1 | grep "^literal" | sed 's/status.*//'
done; done
