"""CuMatrix-style view over a torch CUDA tensor whose methods call the k3_mat_* C ABI (include/k3hip.h): the Python mirror of
the adapter a Kaldi build would put under cudamatrix/cu-matrix.cc.  Method names and argument order follow CuMatrixBase
(cudamatrix/cu-matrix.h:79-791); kTrans/kNoTrans as booleans."""
import ctypes, torch
from . import lib as _l

def _st(): return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

class CuMatrix:
    def __init__(self, t):
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1
        self.t = t; self._L = _l.load()
    def NumRows(self): return self.t.shape[0]
    def NumCols(self): return self.t.shape[1]
    def Stride(self): return self.t.stride(0)
    def _a(self): return (self.t.data_ptr(), self.Stride(), self.NumRows(), self.NumCols())
    def AddMatMat(self, alpha, A, transA, B, transB, beta):
        M, N = self.NumRows(), self.NumCols(); K = A.NumRows() if transA else A.NumCols()
        assert (A.NumCols() if transA else A.NumRows()) == M and (B.NumRows() if transB else B.NumCols()) == N and (B.NumCols() if transB else B.NumRows()) == K
        _l.check(self._L.k3_mat_add_mat_mat(alpha, A.t.data_ptr(), A.Stride(), int(transA), B.t.data_ptr(), B.Stride(), int(transB), beta, self.t.data_ptr(), self.Stride(), M, N, K, _st()))
    def SetZero(self): self.Set(0.0)
    def Set(self, v): _l.check(self._L.k3_mat_set(*self._a(), v, _st()))
    def Scale(self, v): _l.check(self._L.k3_mat_scale(*self._a(), v, _st()))
    def Add(self, v): _l.check(self._L.k3_mat_add(*self._a(), v, _st()))
    def ApplyFloor(self, v): _l.check(self._L.k3_mat_apply_floor(*self._a(), v, _st()))
    def ApplyCeiling(self, v): _l.check(self._L.k3_mat_apply_ceiling(*self._a(), v, _st()))
    def CopyRowsFromVec(self, v): _l.check(self._L.k3_mat_copy_rows_from_vec(*self._a(), v.data_ptr(), _st()))
    def MulColsVec(self, v): _l.check(self._L.k3_mat_mul_cols_vec(*self._a(), v.data_ptr(), _st()))
    def MulRowsVec(self, v): _l.check(self._L.k3_mat_mul_rows_vec(*self._a(), v.data_ptr(), _st()))
    def AddVecToRows(self, alpha, row, beta=1.0): _l.check(self._L.k3_mat_add_vec_to_rows(alpha, row.data_ptr(), beta, *self._a(), _st()))
    def AddVecToCols(self, alpha, col, beta=1.0): _l.check(self._L.k3_mat_add_vec_to_cols(alpha, col.data_ptr(), beta, *self._a(), _st()))
    def CopyFromMat(self, M, trans=False): _l.check(self._L.k3_mat_copy_from_mat(*self._a(), M.t.data_ptr(), M.Stride(), int(trans), _st()))
    def AddMat(self, alpha, A, transA=False): _l.check(self._L.k3_mat_add_mat(alpha, A.t.data_ptr(), A.Stride(), int(transA), *self._a(), _st()))
    def CopyRows(self, src, indexes): _l.check(self._L.k3_mat_copy_rows(*self._a(), src.t.data_ptr(), src.Stride(), indexes.data_ptr(), _st()))
    def AddRows(self, alpha, src, indexes): _l.check(self._L.k3_mat_add_rows(alpha, src.t.data_ptr(), src.Stride(), indexes.data_ptr(), *self._a(), _st()))
    # ---- the parameter update's operations (natural gradient, max-change, orthonormal constraint)
    def SymAddMat2(self, alpha, A, transA, beta): self.AddMatMat(alpha, A, transA, A, not transA, beta)      # both triangles
    def CopyLowerToUpper(self): assert self.NumRows() == self.NumCols(); _l.check(self._L.k3_mat_copy_lower_to_upper(self.t.data_ptr(), self.Stride(), self.NumRows(), _st()))
    def AddToDiag(self, v): _l.check(self._L.k3_mat_add_to_diag(*self._a(), v, _st()))
    def AddVecVec(self, alpha, x, y): _l.check(self._L.k3_mat_add_vec_vec(alpha, x.data_ptr(), y.data_ptr(), *self._a(), _st()))
    def DivElements(self, A): _l.check(self._L.k3_mat_div_elements(*self._a(), A.t.data_ptr(), A.Stride(), _st()))
    def AddDiagVecMat(self, alpha, v, M, transM, beta):
        _l.check(self._L.k3_mat_add_diag_vec_mat(alpha, v.data_ptr(), M.t.data_ptr(), M.Stride(), int(transM), beta, self.t.data_ptr(), self.Stride(), self.NumRows(), self.NumCols(), _st()))
    # ---- the remaining nonlinearities of nnet3's simple components and their derivatives (cu-matrix.h:288-307,:386-396,:501-511; cu-math.h:272-300)
    def _map(self, op, src, a=0.0, flag=0): _l.check(self._L.k3_mat_apply_map(op, *self._a(), src.t.data_ptr(), src.Stride(), a, flag, _st()))
    def Sigmoid(self, src): self._map(0, src)
    def Tanh(self, src): self._map(1, src)
    def Log(self, src): self._map(2, src)
    def Pow(self, src, power): self._map(3, src, power)
    def PowAbs(self, src, power, include_sign=False): self._map(4, src, power, int(include_sign))
    def MaxMat(self, A): self._map(5, A)      # CuMatrixBase::Max(const CuMatrixBase &A) (Max() without an argument is the scalar reduction)
    def DiffSigmoid(self, value, diff): _l.check(self._L.k3_mat_diff_activation(0, *self._a(), value.t.data_ptr(), value.Stride(), diff.t.data_ptr(), diff.Stride(), _st()))
    def DiffTanh(self, value, diff): _l.check(self._L.k3_mat_diff_activation(1, *self._a(), value.t.data_ptr(), value.Stride(), diff.t.data_ptr(), diff.Stride(), _st()))
    def DivRowsVec(self, div): _l.check(self._L.k3_mat_div_rows_vec(*self._a(), div.data_ptr(), _st()))
    def CopyColsFromVec(self, v):
        if v.numel() == self.NumRows() * self.NumCols(): _l.check(self._L.k3_mat_copy_from_mat(*self._a(), v.data_ptr(), self.NumRows(), 1, _st()))      # the vector is the matrix column by column
        else: assert v.numel() == self.NumRows(); _l.check(self._L.k3_mat_copy_cols_from_vec(*self._a(), v.data_ptr(), _st()))
    def CopyColFromVec(self, v, col): assert v.numel() == self.NumRows(); _l.check(self._L.k3_mat_copy_from_mat(self.t.data_ptr() + 4 * col, self.Stride(), self.NumRows(), 1, v.data_ptr(), 1, 0, _st()))
    def CopyCols(self, src, indexes): _l.check(self._L.k3_mat_copy_cols(0, *self._a(), src.t.data_ptr(), src.Stride(), indexes.data_ptr(), _st()))
    def MulRows(self, src, indexes): _l.check(self._L.k3_mat_mul_rows(*self._a(), src.t.data_ptr(), src.Stride(), indexes.data_ptr(), _st()))
    def SetMatMatDivMat(self, A, B, C): _l.check(self._L.k3_mat_elements3(0, *self._a(), 1.0, A.t.data_ptr(), A.Stride(), B.t.data_ptr(), B.Stride(), C.t.data_ptr(), C.Stride(), 0.0, _st()))
    def AddMatMatElements(self, alpha, A, B, beta): _l.check(self._L.k3_mat_elements3(1, *self._a(), alpha, A.t.data_ptr(), A.Stride(), B.t.data_ptr(), B.Stride(), None, 0, beta, _st()))
    def AddCols(self, src, indexes): _l.check(self._L.k3_mat_copy_cols(1, *self._a(), src.t.data_ptr(), src.Stride(), indexes.data_ptr(), _st()))
    def _reduce(self, op, B=None):
        r = ctypes.c_double(0.0)
        _l.check(self._L.k3_mat_reduce_scalar(op, self.t.data_ptr(), self.Stride(), B.t.data_ptr() if B is not None else None, B.Stride() if B is not None else 0, self.NumRows(), self.NumCols(), ctypes.byref(r), _st()))
        return r.value
    def Trace(self): return self._reduce(2)
    def Sum(self): return self._reduce(3)
    def Max(self): return self._reduce(4)
    def Min(self): return self._reduce(5)

def TraceMatMat(A, B, trans=False): return A._reduce(0 if trans else 1, B)

def NormalizePerRow(inp, target_rms, add_log_stddev, out):      # cu::NormalizePerRow (cudamatrix/cu-math.h:272)
    assert out.NumRows() == inp.NumRows() and out.NumCols() == inp.NumCols() + int(add_log_stddev)
    _l.check(out._L.k3_mat_normalize_rows(0, out.t.data_ptr(), out.Stride(), inp.t.data_ptr(), inp.Stride(), None, 0, inp.NumRows(), inp.NumCols(), target_rms, int(add_log_stddev), _st()))
def DiffNormalizePerRow(in_value, out_deriv, target_rms, add_log_stddev, in_deriv):      # cu::DiffNormalizePerRow (cudamatrix/cu-math.h:296): ADDS to in_deriv unless it is out_deriv
    assert out_deriv.NumCols() == in_value.NumCols() + int(add_log_stddev) and in_deriv.NumCols() == in_value.NumCols()
    _l.check(in_deriv._L.k3_mat_normalize_rows(1, in_deriv.t.data_ptr(), in_deriv.Stride(), in_value.t.data_ptr(), in_value.Stride(), out_deriv.t.data_ptr(), out_deriv.Stride(), in_value.NumRows(), in_value.NumCols(), target_rms, int(add_log_stddev), _st()))

class CuRand:
    """CuRand<BaseFloat> (cudamatrix/cu-rand.h:31-64) over k3_mat_set_rand: one seed per object, the counter position advances by what each fill consumes"""
    def __init__(self, seed=0): self.seed, self.offset, self._L = int(seed), 0, _l.load()
    def _fill(self, kind, M):
        _l.check(self._L.k3_mat_set_rand(kind, M.t.data_ptr(), M.Stride(), M.NumRows(), M.NumCols(), self.seed, self.offset, _st())); self.offset += (M.NumRows() * M.NumCols() + 3) // 4
    def RandUniform(self, M): self._fill(0, M)
    def RandGaussian(self, M): self._fill(1, M)
