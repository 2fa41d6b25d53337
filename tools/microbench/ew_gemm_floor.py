import sys, os, torch
sys.path.insert(0, os.getcwd())
from kaldi_amd.cumatrix import CuMatrix
dev = torch.device("cuda:0")
def t(f, n=40):
    f(); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (M, N) in [(4736, 768), (14336, 768), (4736, 96)]:
    C = CuMatrix(torch.randn(M, N, device=dev)); D = CuMatrix(torch.randn(M, N, device=dev))
    print(M, N, "Scale %.1f us" % t(lambda: C.Scale(1.0001)), "AddMat %.1f us" % t(lambda: C.AddMat(0.001, D)), "torch mul_ %.1f us" % t(lambda: C.t.mul_(1.0001)), "torch add_ %.1f" % t(lambda: C.t.add_(D.t, alpha=0.001)))
for (M, N, K, ta, tb, beta) in [(4736, 768, 96, 0, 1, 1), (4736, 768, 96, 0, 1, 0), (4736, 768, 16, 0, 1, 1), (4736, 768, 16, 0, 1, 0), (4736, 768, 384, 0, 1, 1), (4736, 768, 768, 0, 1, 1), (4736, 768, 1536, 0, 1, 1)]:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev); C = torch.zeros((M, N), device=dev)
    c = CuMatrix(C); a = CuMatrix(A); b = CuMatrix(B)
    us = t(lambda: c.AddMatMat(1e-3, a, bool(ta), b, bool(tb), float(beta)))
    print("M %5d N %5d K %5d ta %d tb %d beta %d: %6.1f us = %5.1f TFLOP/s" % (M, N, K, ta, tb, beta, us, 2.0 * M * N * K / us / 1e6), "torch addmm %.1f us" % t(lambda: torch.addmm(C, A.T if ta else A, B.T if tb else B, beta=beta, alpha=1e-3, out=C)))
