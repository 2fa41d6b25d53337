"""GPU tests of the CuMatrix-operation kernels (k3_mat_*), in the style of the reference's cudamatrix/cu-matrix-test.cc: random
sizes incl. ragged edges and strides larger than the row length, each method against the same operation in numpy."""
import numpy as np, pytest, torch
pytestmark = pytest.mark.gpu

def _mat(rng, r, c, pad=0):
    base = torch.from_numpy(rng.standard_normal((r, c + pad)).astype(np.float32)).cuda()
    return base[:, :c] if pad else base

def test_elementwise_and_row_ops_match_numpy():
    from kaldi_amd.cumatrix import CuMatrix
    rng = np.random.default_rng(0)
    for r, c, pad in [(1, 1, 0), (7, 130, 3), (257, 65, 0), (33, 768, 8)]:
        t = _mat(rng, r, c, pad); M = CuMatrix(t); ref = t.cpu().numpy().copy()
        v = torch.from_numpy(rng.standard_normal(c).astype(np.float32)).cuda(); w = torch.from_numpy(rng.standard_normal(r).astype(np.float32)).cuda()
        M.Scale(0.5); ref *= np.float32(0.5)
        M.ApplyFloor(-0.1); ref = np.maximum(ref, np.float32(-0.1))
        M.ApplyCeiling(0.9); ref = np.minimum(ref, np.float32(0.9))
        M.MulColsVec(v); ref = ref * v.cpu().numpy()
        M.AddVecToRows(0.75, v, 0.5); ref = np.float32(0.75) * v.cpu().numpy() + np.float32(0.5) * ref
        M.MulRowsVec(w); ref = ref * w.cpu().numpy()[:, None]
        M.AddVecToCols(2.0, w); ref = np.float32(2.0) * w.cpu().numpy()[:, None] + ref
        M.Add(0.25); ref = ref + np.float32(0.25)
        A = CuMatrix(_mat(rng, r, c, 5)); M.AddMat(-1.5, A); ref = ref + np.float32(-1.5) * A.t.cpu().numpy()
        At = CuMatrix(_mat(rng, c, r, 2)); M.AddMat(0.3, At, True); ref = ref + np.float32(0.3) * At.t.cpu().numpy().T
        torch.cuda.synchronize()
        assert np.allclose(t.cpu().numpy(), ref, rtol=1e-5, atol=1e-5), (r, c)
        M.CopyRowsFromVec(v); assert np.array_equal(t.cpu().numpy(), np.tile(v.cpu().numpy(), (r, 1)))
        M.CopyFromMat(At, True); assert np.array_equal(t.cpu().numpy(), At.t.cpu().numpy().T)
        M.SetZero(); assert not t.cpu().numpy().any()
        src = CuMatrix(_mat(rng, 19, c, 1)); idx = torch.from_numpy(rng.integers(-1, 19, r).astype(np.int32)).cuda(); ih = idx.cpu().numpy()
        M.CopyRows(src, idx); want = np.where(ih[:, None] >= 0, src.t.cpu().numpy()[np.maximum(ih, 0)], 0.0); assert np.array_equal(t.cpu().numpy(), want)
        M.AddRows(2.0, src, idx); want = want + np.where(ih[:, None] >= 0, np.float32(2.0) * src.t.cpu().numpy()[np.maximum(ih, 0)], 0.0)
        assert np.allclose(t.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
        if pad: assert t._base is not None                                   # the padding columns of the parent buffer were never written

@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_add_mat_mat_matches_numpy(ta, tb):
    from kaldi_amd.cumatrix import CuMatrix
    rng = np.random.default_rng(1)
    for M_, N_, K_ in [(1, 1, 1), (64, 64, 16), (65, 63, 17), (130, 96, 192), (300, 6024 // 8, 192)]:
        A = CuMatrix(_mat(rng, *((K_, M_) if ta else (M_, K_)), 3)); B = CuMatrix(_mat(rng, *((N_, K_) if tb else (K_, N_)), 1)); C = CuMatrix(_mat(rng, M_, N_, 2))
        c0 = C.t.cpu().numpy().copy(); a = A.t.cpu().numpy().T if ta else A.t.cpu().numpy(); b = B.t.cpu().numpy().T if tb else B.t.cpu().numpy()
        C.AddMatMat(0.7, A, ta, B, tb, 1.3); torch.cuda.synchronize()
        want = 0.7 * (a.astype(np.float64) @ b.astype(np.float64)) + 1.3 * c0
        assert np.allclose(C.t.cpu().numpy(), want, rtol=2e-5, atol=2e-4), (M_, N_, K_)
        C.AddMatMat(1.0, A, ta, B, tb, 0.0); torch.cuda.synchronize()             # beta = 0 must not read C (it may hold NaN, cu-matrix.cc:1340)
        assert np.allclose(C.t.cpu().numpy(), a.astype(np.float64) @ b.astype(np.float64), rtol=2e-5, atol=2e-4)
