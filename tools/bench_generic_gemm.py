import sys, time, torch
sys.path.insert(0, "/root/repo")
import os; sys.path.insert(0, os.getcwd())
from kaldi_amd.cumatrix import CuMatrix
dev = torch.device("cuda:0")
for (M, N, K, ta, tb) in [(8192, 768, 1536, False, True), (8192, 1536, 768, False, False), (768, 1536, 8192, True, False), (4096, 6024, 192, False, True)]:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev); C = torch.zeros((M, N), device=dev)
    c = CuMatrix(C); a = CuMatrix(A); b = CuMatrix(B)
    c.AddMatMat(1.0, a, ta, b, tb, 0.0); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(5): c.AddMatMat(1.0, a, ta, b, tb, 0.0)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
    ref = (A.T if ta else A).double() @ (B.T if tb else B).double()
    print("M %5d N %5d K %5d ta %d tb %d: %.3f ms = %.1f TFLOP/s, max rel err %.2e" % (M, N, K, ta, tb, ms, 2.0 * M * N * K / ms / 1e9, float(((C.double() - ref).abs().max() / ref.abs().max()))))
