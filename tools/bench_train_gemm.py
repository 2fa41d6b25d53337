"""The products of one nnet3-chain-train iteration on the benchmark model (shapes from tools/train_trace_report.py), each timed over 40 back-to-back launches through k3_mat_add_mat_mat
(beta as in the run), checked against float64.   K3_GEMM_BK=16|32|64 python tools/bench_train_gemm.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd.cumatrix import CuMatrix
dev = torch.device("cuda:0"); tot = 0.0
SH = [(4736, 768, 96, 0, 1, 1, 27), (4736, 768, 96, 0, 0, 1, 25), (4736, 96, 768, 0, 0, 1, 27), (4736, 96, 768, 0, 1, 1, 25), (4736, 1536, 20, 0, 0, 1, 12), (14336, 768, 96, 0, 0, 1, 6), (4736, 768, 80, 0, 0, 1, 15),
      (768, 192, 4736, 1, 0, 1, 14), (96, 1536, 4736, 1, 0, 1, 12), (768, 1, 4736, 1, 0, 1, 15), (14336, 1536, 20, 0, 0, 1, 3), (4736, 80, 768, 0, 1, 0, 15), (14336, 768, 96, 0, 1, 1, 4), (14336, 96, 768, 0, 1, 1, 6),
      (4736, 20, 1536, 0, 1, 0, 12), (4736, 196, 20, 0, 0, 1, 15), (4736, 20, 196, 0, 1, 0, 15), (96, 1536, 14336, 1, 0, 1, 3), (14336, 768, 80, 0, 0, 1, 3), (4736, 96, 80, 0, 0, 1, 13), (4736, 80, 96, 0, 1, 0, 13),
      (4736, 6024, 192, 0, 1, 1, 1), (160, 80, 768, 0, 1, 0, 12), (40, 20, 1536, 0, 1, 0, 9), (80, 768, 80, 0, 0, 0, 12)]
for (M, N, K, ta, tb, beta, n) in SH:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev); C = torch.zeros((M, N), device=dev)
    c = CuMatrix(C); a = CuMatrix(A); b = CuMatrix(B)
    c.AddMatMat(1.0, a, bool(ta), b, bool(tb), 0.0); torch.cuda.synchronize()
    ref = (A.T if ta else A).double() @ (B.T if tb else B).double(); err = float(((C.double() - ref).abs().max() / ref.abs().max()))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(40): c.AddMatMat(1.0 if beta == 0 else 1e-3, a, bool(ta), b, bool(tb), float(beta))
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) / 40 * 1e3; tot += us * n
    print("M %5d N %5d K %5d ta %d tb %d beta %d: %6.1f us = %5.1f TFLOP/s, x%d per iteration, max rel err %.1e" % (M, N, K, ta, tb, beta, us, 2.0 * M * N * K / us / 1e6, n, err))
print("sum over an iteration's calls of these shapes: %.2f ms" % (tot / 1e3))
