"""k3_mat_reduce_scalar (TraceMatMat & co.): host-side time per call (launch -> result on the host) against torch's (a * b).sum().item()"""
import sys, os, time, ctypes, torch
sys.path.insert(0, os.getcwd())
from kaldi_amd import lib as _l
L = _l.load(); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); out = ctypes.c_double()
for (r, c) in [(1, 768), (768, 768), (4736, 96), (4736, 768), (14336, 768), (14336, 1536)]:
    a = torch.randn(r, c, device="cuda"); b = torch.randn(r, c, device="cuda")
    def f(): _l.check(L.k3_mat_reduce_scalar(0, ctypes.c_void_p(a.data_ptr()), ctypes.c_int64(c), ctypes.c_void_p(b.data_ptr()), ctypes.c_int64(c), r, c, ctypes.byref(out), st))
    def g(): return float((a * b).sum().item())
    res = []
    for fn in (f, g):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        res.append((time.perf_counter() - t0) / 200 * 1e6)
    f(); ref = float((a.double() * b.double()).sum().item())
    print("%6d x %5d: k3 %.1f us per call, torch %.1f us; value %.6g vs %.6g" % (r, c, res[0], res[1], out.value, ref))
