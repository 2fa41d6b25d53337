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
    for r, c, pad in [(1, 1, 0), (7, 130, 3), (257, 65, 0), (33, 768, 8), (70, 256, 0), (19, 64, 4)]:
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
        A4 = CuMatrix(_mat(rng, r, c, 4)); M.AddMat(0.7, A4); ref = ref + np.float32(0.7) * A4.t.cpu().numpy()      # (a source whose stride keeps rows 16-byte aligned: the four-columns-at-a-time kernel)
        M.DivElements(CuMatrix(A4.t.abs() + 1.0)); ref = ref / (np.abs(A4.t.cpu().numpy()) + np.float32(1.0))
        torch.cuda.synchronize()
        assert np.allclose(t.cpu().numpy(), ref, rtol=1e-5, atol=1e-5), (r, c)
        M.CopyRowsFromVec(v); assert np.array_equal(t.cpu().numpy(), np.tile(v.cpu().numpy(), (r, 1)))
        M.CopyFromMat(At, True); assert np.array_equal(t.cpu().numpy(), At.t.cpu().numpy().T)
        M.SetZero(); assert not t.cpu().numpy().any()
        src = CuMatrix(_mat(rng, 19, c, 1 if r % 2 else 4)); idx = torch.from_numpy(rng.integers(-1, 19, r).astype(np.int32)).cuda(); ih = idx.cpu().numpy()
        M.CopyRows(src, idx); want = np.where(ih[:, None] >= 0, src.t.cpu().numpy()[np.maximum(ih, 0)], 0.0); assert np.array_equal(t.cpu().numpy(), want)
        M.AddRows(2.0, src, idx); want = want + np.where(ih[:, None] >= 0, np.float32(2.0) * src.t.cpu().numpy()[np.maximum(ih, 0)], 0.0)
        assert np.allclose(t.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
        if pad: assert t._base is not None                                   # the padding columns of the parent buffer were never written

@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_add_mat_mat_matches_numpy(ta, tb):
    from kaldi_amd.cumatrix import CuMatrix
    rng = np.random.default_rng(1)
    for M_, N_, K_ in [(1, 1, 1), (64, 64, 16), (65, 63, 17), (130, 96, 192), (300, 6024 // 8, 192), (96, 200, 5000), (33, 70, 3072), (4736, 96, 768), (160, 80, 768), (1000, 1536, 20), (20, 196, 4700), (4700, 20, 196)]:      # long K and few tiles -> the split-K paths; the shapes of a training minibatch (64-tiles, K split in steps of 64)
        tol_a = 2e-4 * max(1.0, (K_ / 192.0) ** 0.5)
        A = CuMatrix(_mat(rng, *((K_, M_) if ta else (M_, K_)), 3)); B = CuMatrix(_mat(rng, *((N_, K_) if tb else (K_, N_)), 1)); C = CuMatrix(_mat(rng, M_, N_, 2))
        c0 = C.t.cpu().numpy().copy(); a = A.t.cpu().numpy().T if ta else A.t.cpu().numpy(); b = B.t.cpu().numpy().T if tb else B.t.cpu().numpy()
        C.AddMatMat(0.7, A, ta, B, tb, 1.3); torch.cuda.synchronize()
        want = 0.7 * (a.astype(np.float64) @ b.astype(np.float64)) + 1.3 * c0
        assert np.allclose(C.t.cpu().numpy(), want, rtol=2e-5, atol=tol_a), (M_, N_, K_)
        C.AddMatMat(1.0, A, ta, B, tb, 0.0); torch.cuda.synchronize()             # beta = 0 must not read C (it may hold NaN, cu-matrix.cc:1340)
        assert np.allclose(C.t.cpu().numpy(), a.astype(np.float64) @ b.astype(np.float64), rtol=2e-5, atol=tol_a)
        C2 = CuMatrix(torch.full_like(C.t, float('nan'))); C2.AddMatMat(1.0, A, ta, B, tb, 0.0); torch.cuda.synchronize(); assert np.array_equal(C2.t.cpu().numpy(), C.t.cpu().numpy())      # deterministic, and beta = 0 ignores NaNs


def test_update_step_operations_match_numpy():
    """The operations the reference's parameter update adds to the forward / backward set (OnlineNaturalGradient::PreconditionDirections, UpdateNnetWithMaxChange,
    ConstrainOrthonormalInternal): SymAddMat2, CopyLowerToUpper, AddToDiag, Trace, TraceMatMat (both transposes), Sum / Max / Min, AddVecVec, DivElements, AddDiagVecMat, and the
    vector forms the adapter builds from matrix calls with leading dimension 1 (AddMatVec = AddMatMat with one column, CopyColFromMat)."""
    import ctypes
    from kaldi_amd import lib as _l
    from kaldi_amd.cumatrix import CuMatrix, TraceMatMat
    rng = np.random.default_rng(7); dev = torch.device("cuda:0")
    def T(a): return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    for (m, n) in [(24, 96), (5, 7), (96, 24), (130, 257)]:
        A = rng.standard_normal((m, n)).astype(np.float32); P0 = rng.standard_normal((m, m)).astype(np.float32)
        P = CuMatrix(T(P0)); P.SymAddMat2(0.5, CuMatrix(T(A)), False, 0.25); want = 0.25 * P0 + 0.5 * (A.astype(np.float64) @ A.T.astype(np.float64))
        assert np.abs(P.t.cpu().numpy() - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
        Q = CuMatrix(T(A.T.copy())); P2 = CuMatrix(T(np.zeros((m, m)))); P2.SymAddMat2(1.0, Q, True, 0.0)      # A^T given, transposed: the same product
        assert np.abs(P2.t.cpu().numpy() - A.astype(np.float64) @ A.T.astype(np.float64)).max() <= 2e-5 * np.abs(want).max()
        L = CuMatrix(T(P0)); L.CopyLowerToUpper(); w = np.tril(P0) + np.tril(P0, -1).T; assert np.array_equal(L.t.cpu().numpy(), w)
        D = CuMatrix(T(A)); D.AddToDiag(1.5); w = A.copy(); w[np.arange(min(m, n)), np.arange(min(m, n))] += 1.5; assert np.array_equal(D.t.cpu().numpy(), w)
        assert abs(CuMatrix(T(P0)).Trace() - np.trace(P0.astype(np.float64))) <= 1e-5 * m
        B = rng.standard_normal((m, n)).astype(np.float32)
        assert abs(TraceMatMat(CuMatrix(T(A)), CuMatrix(T(B)), True) - (A.astype(np.float64) * B).sum()) <= 1e-6 * m * n
        assert abs(TraceMatMat(CuMatrix(T(A)), CuMatrix(T(B.T.copy())), False) - (A.astype(np.float64) * B).sum()) <= 1e-6 * m * n
        a = CuMatrix(T(A)); assert abs(a.Sum() - A.astype(np.float64).sum()) <= 1e-6 * m * n and a.Max() == A.max() and a.Min() == A.min()
        x = rng.standard_normal(m).astype(np.float32); y = rng.standard_normal(n).astype(np.float32)
        V = CuMatrix(T(A)); V.AddVecVec(0.3, T(x), T(y)); assert np.abs(V.t.cpu().numpy() - (A + np.float32(0.3) * np.outer(x, y))).max() <= 1e-5
        E = CuMatrix(T(A)); Bp = np.abs(B) + 0.5; E.DivElements(CuMatrix(T(Bp))); assert np.abs(E.t.cpu().numpy() - A / Bp).max() <= 1e-5
        G = CuMatrix(T(A)); G.AddDiagVecMat(0.7, T(x), CuMatrix(T(B)), False, 0.2); assert np.abs(G.t.cpu().numpy() - (0.2 * A + 0.7 * x[:, None] * B)).max() <= 1e-5
        G = CuMatrix(T(A)); G.AddDiagVecMat(0.7, T(x), CuMatrix(T(B.T.copy())), True, 0.2); assert np.abs(G.t.cpu().numpy() - (0.2 * A + 0.7 * x[:, None] * B)).max() <= 1e-5
        # CuVectorBase::AddMatVec as the adapter issues it: out [m x 1] (ld 1) = beta out + alpha M v, v as a [n x 1] matrix of ld 1; and with M transposed
        Lh = _l.load(); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        out0 = rng.standard_normal(m).astype(np.float32); o = T(out0); Md = T(A); vd = T(y)
        _l.check(Lh.k3_mat_add_mat_mat(0.5, Md.data_ptr(), n, 0, vd.data_ptr(), 1, 0, 0.25, o.data_ptr(), 1, m, 1, n, st))
        assert np.abs(o.cpu().numpy() - (0.25 * out0 + 0.5 * (A.astype(np.float64) @ y))).max() <= 2e-5 * max(1.0, np.abs(A @ y).max())
        out1 = rng.standard_normal(n).astype(np.float32); o = T(out1); xd = T(x)
        _l.check(Lh.k3_mat_add_mat_mat(0.5, Md.data_ptr(), n, 1, xd.data_ptr(), 1, 0, 0.25, o.data_ptr(), 1, n, 1, m, st))
        assert np.abs(o.cpu().numpy() - (0.25 * out1 + 0.5 * (A.T.astype(np.float64) @ x))).max() <= 2e-5 * max(1.0, np.abs(A.T @ x).max())
        col = T(np.zeros(m)); _l.check(Lh.k3_mat_copy_from_mat(col.data_ptr(), 1, m, 1, Md.data_ptr() + 4 * (n // 2), n, 0, st)); assert np.array_equal(col.cpu().numpy(), A[:, n // 2])      # CopyColFromMat


def test_nonlinearity_maps_column_operations_and_row_normalisation_match_numpy():
    """Sigmoid / Tanh / Log / Pow / PowAbs / Max / DiffSigmoid / DiffTanh / DivRowsVec / CopyCols / AddCols / CopyColsFromVec / CopyColFromVec (cu-matrix.h:102-111,:281-307,:386-396,:501-511) and
    cu::NormalizePerRow / DiffNormalizePerRow (cu-math.cc:280-409, restated here in float64 from the CPU branch, incl. the log-stddev column, the floor and the in-place form)"""
    from kaldi_amd import cumatrix as cm
    from kaldi_amd.cumatrix import CuMatrix
    rng = np.random.default_rng(5)
    for r, c, pad in [(1, 1, 0), (7, 130, 3), (257, 65, 0), (33, 768, 8), (70, 256, 0), (19, 64, 4)]:
        S = CuMatrix(_mat(rng, r, c, pad)); S.t.mul_(4.0); sh = S.t.cpu().numpy().astype(np.float64)
        D = CuMatrix(_mat(rng, r, c, 4)); d0 = D.t.cpu().numpy().astype(np.float64)
        chk = lambda want, tol=2e-6: (torch.cuda.synchronize(), np.testing.assert_allclose(D.t.cpu().numpy(), want, rtol=tol, atol=tol))
        D.Sigmoid(S); chk(1.0 / (1.0 + np.exp(-sh)))
        D.Tanh(S); chk(np.tanh(sh)); y = D.t.cpu().numpy().astype(np.float64)
        G = CuMatrix(_mat(rng, r, c, 0)); gh = G.t.cpu().numpy().astype(np.float64)
        E = CuMatrix(torch.empty_like(G.t)); E.DiffTanh(D, G); torch.cuda.synchronize(); np.testing.assert_allclose(E.t.cpu().numpy(), gh * (1.0 - y * y), rtol=1e-6, atol=1e-6)
        G.DiffSigmoid(D, G); torch.cuda.synchronize(); np.testing.assert_allclose(G.t.cpu().numpy(), gh * y * (1.0 - y), rtol=1e-6, atol=1e-6)      # in place on diff
        P = CuMatrix(S.t.abs() + 0.5); ph = P.t.cpu().numpy().astype(np.float64)
        D.Log(P); chk(np.log(ph)); D.Pow(P, 1.7); chk(ph ** 1.7, 1e-5); D.PowAbs(S, 0.5, True); chk(np.sign(sh) * np.abs(sh) ** 0.5, 1e-5); D.PowAbs(S, 2.0); chk(sh * sh, 1e-5)
        D.CopyFromMat(CuMatrix(torch.from_numpy(d0.astype(np.float32)).cuda())); D.MaxMat(S); chk(np.maximum(d0, sh))
        w = torch.from_numpy((rng.random(r) + 0.5).astype(np.float32)).cuda(); D.DivRowsVec(w); chk(np.maximum(d0, sh) / w.cpu().numpy().astype(np.float64)[:, None])
        src = CuMatrix(_mat(rng, r, 23, 1)); idx = torch.from_numpy(rng.integers(-1, 23, c).astype(np.int32)).cuda(); ih = idx.cpu().numpy(); sc = src.t.cpu().numpy()
        D.CopyCols(src, idx); torch.cuda.synchronize(); want = np.where(ih[None, :] >= 0, sc[:, np.maximum(ih, 0)], 0.0).astype(np.float32); assert np.array_equal(D.t.cpu().numpy(), want)
        D.AddCols(src, idx); torch.cuda.synchronize(); assert np.array_equal(D.t.cpu().numpy(), want + want)
        msk = CuMatrix(_mat(rng, 11, c, 4 if c % 4 == 0 else 1)); ridx = torch.from_numpy(rng.integers(-1, 11, r).astype(np.int32)).cuda(); rh = ridx.cpu().numpy(); before = D.t.cpu().numpy().copy()
        D.MulRows(msk, ridx); torch.cuda.synchronize(); assert np.array_equal(D.t.cpu().numpy(), np.where(rh[:, None] >= 0, before * msk.t.cpu().numpy()[np.maximum(rh, 0)], before))
        A3 = CuMatrix(_mat(rng, r, c, 1)); B3 = CuMatrix(_mat(rng, r, c, 0)); C3 = CuMatrix(_mat(rng, r, c, 2)); C3.t[0, 0] = 0.0; ah, bh, ch = (q.t.cpu().numpy() for q in (A3, B3, C3)); before = D.t.cpu().numpy().copy()
        D.AddMatMatElements(0.5, A3, B3, 2.0); torch.cuda.synchronize(); np.testing.assert_allclose(D.t.cpu().numpy(), np.float32(2.0) * before + np.float32(0.5) * ah * bh, rtol=1e-6, atol=1e-6)
        D.SetMatMatDivMat(A3, B3, C3); torch.cuda.synchronize(); np.testing.assert_allclose(D.t.cpu().numpy(), np.where(ch != 0, ah * (bh / np.where(ch != 0, ch, 1)), ah), rtol=1e-6, atol=1e-6); assert D.t[0, 0].item() == A3.t[0, 0].item()
        D.CopyColsFromVec(w); torch.cuda.synchronize(); assert np.array_equal(D.t.cpu().numpy(), np.tile(w.cpu().numpy()[:, None], (1, c)))
        full = torch.from_numpy(rng.standard_normal(r * c).astype(np.float32)).cuda(); D.CopyColsFromVec(full); torch.cuda.synchronize(); assert np.array_equal(D.t.cpu().numpy(), full.cpu().numpy().reshape(c, r).T)
        D.CopyColFromVec(w, c - 1); torch.cuda.synchronize(); assert np.array_equal(D.t.cpu().numpy()[:, c - 1], w.cpu().numpy()) and (c == 1 or np.array_equal(D.t.cpu().numpy()[:, 0], full.cpu().numpy()[:r]))
        # row normalisation
        X = CuMatrix(_mat(rng, r, c, pad)); X.t[r // 2].zero_(); xh = X.t.cpu().numpy().astype(np.float64)
        for rms, add_log in ((1.0, False), (0.5, True)):
            ds = c * rms * rms; ss = (xh * xh).sum(1); f = np.maximum(ss / ds, 2.0 ** -66) ** -0.5
            Y = CuMatrix(_mat(rng, r, c + int(add_log), 2)); cm.NormalizePerRow(X, rms, add_log, Y); torch.cuda.synchronize()
            want = xh * f[:, None]
            if add_log: want = np.concatenate([want, (np.log(rms) - np.log(f))[:, None]], 1)
            np.testing.assert_allclose(Y.t.cpu().numpy(), want, rtol=2e-6, atol=2e-6)
            OD = CuMatrix(_mat(rng, r, c + int(add_log), 0)); od = OD.t.cpu().numpy().astype(np.float64); ID = CuMatrix(_mat(rng, r, c, 4)); id0 = ID.t.cpu().numpy().astype(np.float64)
            dot = (od[:, :c] * xh).sum(1); f3 = np.where(ss / ds <= 2.0 ** -66, 0.0, f ** 3)
            core = f[:, None] * od[:, :c] - (dot * f3 / ds)[:, None] * xh
            lsd = (od[:, c] / np.maximum(ss, c * 2.0 ** -66))[:, None] * xh if add_log else 0.0
            cm.DiffNormalizePerRow(X, OD, rms, add_log, ID); torch.cuda.synchronize()
            np.testing.assert_allclose(ID.t.cpu().numpy(), id0 + lsd + core, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(core).max()))
            if not add_log:      # in place (kBackpropInPlace): overwritten, not added to
                cm.DiffNormalizePerRow(X, OD, rms, False, OD); torch.cuda.synchronize(); np.testing.assert_allclose(OD.t.cpu().numpy(), core, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(core).max()))


def test_device_random_numbers_known_answers_reproducibility_and_distribution():
    """k3_mat_set_rand (CuRand<BaseFloat>::RandUniform / RandGaussian, cu-rand.h:50-56): the Philox-4x32-10 known-answer vector of the Random123 distribution (counter 0, key 0 ->
    6627e8d5 e169c58d bc57ac4c 9b00dbd8) through the uniform mapping, independence of stride and launch shape, stream continuity across fills, and the two laws (moments, a chi-square
    over 64 bins, tails) in the manner of the reference's cu-rand-speed-test / cu-matrix-test moments checks"""
    from kaldi_amd.cumatrix import CuMatrix, CuRand
    t = torch.empty(1, 4, device="cuda"); CuRand(0).RandUniform(CuMatrix(t)); torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy()[0], (np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], np.uint32) >> 8).astype(np.float32) / np.float32(2 ** 24))
    # the same (seed, offset) gives the same logical matrix whatever the stride; consecutive fills continue the stream
    a = torch.empty(37, 52, device="cuda"); b = torch.zeros(37, 60, device="cuda")[:, :52]; CuRand(7).RandUniform(CuMatrix(a)); CuRand(7).RandUniform(CuMatrix(b)); torch.cuda.synchronize()
    assert torch.equal(a, b) and not b._base[:, 52:].any()
    r = CuRand(7); c1 = torch.empty(10, 52, device="cuda"); c2 = torch.empty(27, 52, device="cuda"); r.RandUniform(CuMatrix(c1)); r.RandUniform(CuMatrix(c2)); torch.cuda.synchronize()
    assert torch.equal(torch.cat([c1, c2]), a) and r.offset == 37 * 52 // 4
    other = torch.empty(37, 52, device="cuda"); CuRand(8).RandUniform(CuMatrix(other)); assert not torch.equal(other, a)
    n = 1 << 22; u = torch.empty(2048, n // 2048, device="cuda"); CuRand(123).RandUniform(CuMatrix(u)); g = torch.empty_like(u); CuRand(123).RandGaussian(CuMatrix(g)); torch.cuda.synchronize()
    u = u.cpu().numpy().ravel().astype(np.float64); g = g.cpu().numpy().ravel().astype(np.float64)
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 6 * np.sqrt(1 / 12 / n) and abs(u.var() - 1 / 12) < 1e-3
    h = np.bincount((u * 64).astype(int), minlength=64); chi2 = ((h - n / 64) ** 2 / (n / 64)).sum(); assert chi2 < 63 + 6 * np.sqrt(2 * 63), chi2
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5 / np.sqrt(n)
    assert abs(g.mean()) < 6 / np.sqrt(n) and abs(g.var() - 1) < 5e-3 and abs((g ** 3).mean()) < 0.01 and abs((g ** 4).mean() - 3) < 0.03 and np.isfinite(g).all()
    assert abs((np.abs(g) > 3).mean() - 0.0026998) < 3e-4 and np.abs(g).max() < 6.5
