"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes as C
import os

import numpy as np

import spasm_jl_amd as S
from spasm_jl_amd._abi import CsrStruct, EchelonizeOptsStruct, Field, LuStruct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
_P = C.POINTER
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle`")
        h = C.CDLL(LIB_PATH)
        h.orc_field_init.argtypes = [C.c_int64, _P(Field)]
        for name in ("orc_zp_add", "orc_zp_sub", "orc_zp_mul"):
            getattr(h, name).restype = C.c_int32
            getattr(h, name).argtypes = [_P(Field), C.c_int32, C.c_int32]
        h.orc_zp_axpy.restype = C.c_int32
        h.orc_zp_axpy.argtypes = [_P(Field), C.c_int32, C.c_int32, C.c_int32]
        h.orc_zp_inverse.restype = C.c_int32
        h.orc_zp_inverse.argtypes = [_P(Field), C.c_int32]
        h.orc_zp_init.restype = C.c_int32
        h.orc_zp_init.argtypes = [_P(Field), C.c_int64]
        h.orc_csr_alloc.restype = _P(CsrStruct)
        h.orc_csr_alloc.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int]
        h.orc_csr_free.argtypes = [_P(CsrStruct)]
        h.orc_lu_free.argtypes = [_P(LuStruct)]
        h.orc_transpose.restype = _P(CsrStruct)
        h.orc_transpose.argtypes = [_P(CsrStruct)]
        h.orc_echelonize.restype = _P(LuStruct)
        h.orc_echelonize.argtypes = [_P(CsrStruct), _P(EchelonizeOptsStruct), _P(C.c_int64)]
        h.orc_echelonize_init_opts.argtypes = [_P(EchelonizeOptsStruct)]
        h.orc_kernel.restype = _P(CsrStruct)
        h.orc_kernel.argtypes = [_P(LuStruct)]
        h.orc_schur_round.restype = _P(CsrStruct)
        h.orc_schur_round.argtypes = [_P(CsrStruct), _P(C.c_int64), _P(C.c_double), _P(_P(CsrStruct)), _P(C.c_int)]
        h.orc_schur_round_range.restype = _P(CsrStruct)
        h.orc_schur_round_range.argtypes = [_P(CsrStruct), C.c_int, C.c_int, _P(C.c_int64), _P(C.c_double), _P(_P(CsrStruct)), _P(C.c_int)]
        h.orc_sparse_triangular_solve.restype = C.c_int
        h.orc_sparse_triangular_solve.argtypes = [_P(CsrStruct), _P(CsrStruct), C.c_int, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32), _P(C.c_int32), _P(C.c_int64)]
        h.orc_num_threads.restype = C.c_int
        h.orc_set_threads.argtypes = [C.c_int]
        h.orc_set_threads.restype = None
        _lib = h
    return _lib


class OCSR:
    """Read-only view of a spasm_csr owned by the oracle's allocator."""

    def __init__(self, ptr, own=True):
        assert ptr
        self.data = ptr
        self._own = own

    def __del__(self):
        if self._own and self.data:
            lib().orc_csr_free(self.data)
            self.data = None

    n = property(lambda s: int(s.data.contents.n))
    m = property(lambda s: int(s.data.contents.m))
    prime = property(lambda s: int(s.data.contents.field.p))
    p = property(lambda s: np.ctypeslib.as_array(s.data.contents.p, (s.n + 1,)))

    @property
    def nnz(self):
        return int(self.p[self.n])

    @property
    def j(self):
        return np.ctypeslib.as_array(self.data.contents.j, (max(self.nnz, 1),))[: self.nnz]

    @property
    def x(self):
        return np.ctypeslib.as_array(self.data.contents.x, (max(self.nnz, 1),))[: self.nnz]

    def rows(self):
        p, j, x = self.p, self.j, self.x
        return [sorted(zip(j[p[i]:p[i + 1]].tolist(), x[p[i]:p[i + 1]].tolist())) for i in range(self.n)]


class OLU:
    def __init__(self, ptr):
        assert ptr
        self.data = ptr

    def __del__(self):
        if self.data:
            lib().orc_lu_free(self.data)
            self.data = None

    r = property(lambda s: int(s.data.contents.r))
    U = property(lambda s: OCSR(s.data.contents.U, own=False))

    @property
    def qinv(self):
        m = int(self.data.contents.U.contents.m)
        return np.ctypeslib.as_array(self.data.contents.qinv, (max(m, 1),))[:m]

    @property
    def p(self):  # pivotal rows first; `n` (rows of the input) is set by echelonize()
        m = int(self.data.contents.U.contents.m)
        return np.ctypeslib.as_array(self.data.contents.p, (max(getattr(self, "n", 0), m, 1),))


def echelonize(A, **kwargs):
    """Oracle echelonize of a product-side CSR (borrowed)."""
    opts = EchelonizeOptsStruct()
    lib().orc_echelonize_init_opts(C.byref(opts))
    for k, v in kwargs.items():
        setattr(opts, k, v)
    stats = (C.c_int64 * 2)()
    lu = OLU(lib().orc_echelonize(A.data, C.byref(opts), stats))
    lu.stats = (int(stats[0]), int(stats[1]))
    lu.n = int(A.n)
    return lu


def kernel(lu):
    return OCSR(lib().orc_kernel(lu.data))


def sparse_triangular_solve(U, B, k, xj, x, qinv):
    """The oracle's restatement of spasm_sparse_triangular_solve (reference src/SpaSM.jl:694-713) on row k of B: reach (DFS) + scatter.
    xj: int32[3m] zeroed, x: int32[m]; returns top."""
    m = U.m
    q = np.ascontiguousarray(qinv, dtype=np.int32)
    pstack = np.zeros(max(m, 1), dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(_P(C.c_int32))  # noqa: E731
    return int(lib().orc_sparse_triangular_solve(U.data, B.data, int(k), ip(xj), ip(x), ip(q), ip(pstack), None))


def transpose(A):
    return OCSR(lib().orc_transpose(A.data))


def set_threads(n):
    """n > 0: that many OpenMP threads for the oracle's loops from now on; 0: all of them again."""
    lib().orc_set_threads(int(n))


def schur_round(A, want_U=False, row_lo=0, row_hi=None):
    """One Schur round on the CPU (optionally only the non-pivot rows of [row_lo,row_hi)).
    Returns (S, info[, U, qinv])."""
    out = (C.c_int64 * 6)()
    sec = (C.c_double * 2)()
    Uptr = _P(CsrStruct)()
    qinv = np.empty(max(A.m, 1), dtype=np.int32)
    Sp = lib().orc_schur_round_range(A.data, int(row_lo), int(A.n if row_hi is None else row_hi), out, sec,
                                     C.byref(Uptr) if want_U else None,
                                     qinv.ctypes.data_as(_P(C.c_int)) if want_U else None)
    info = dict(npiv=int(out[0]), applications=int(out[1]), nnz_reduced=int(out[2]), nnz_out=int(out[3]),
                rows_out=int(out[4]), nnz_U=int(out[5]), sec_pivots=sec[0], sec_schur=sec[1],
                threads=int(lib().orc_num_threads()))
    if want_U:
        return OCSR(Sp), info, OCSR(Uptr), qinv[: A.m]
    return OCSR(Sp), info


# --------------------------------------------------------------------------------------------
# independent dense checker (numpy, python ints via int64): RREF mod p with leftmost pivots
# --------------------------------------------------------------------------------------------
def dense_rref(D, p):
    """Reduced row echelon form of D mod p (values in [0,p)); returns (R, pivot_columns)."""
    M = np.mod(np.asarray(D, dtype=np.int64), p)
    if p > (1 << 31):
        M = M.astype(object)  # products exceed int64: exact python integers
    n, m = M.shape
    piv = []
    r = 0
    for c in range(m):
        if r == n:
            break
        nzr = np.nonzero(M[r:, c])[0]
        if nzr.size == 0:
            continue
        k = r + int(nzr[0])
        if k != r:
            M[[r, k]] = M[[k, r]]
        inv = pow(int(M[r, c]), -1, p)
        M[r] = (M[r] * inv) % p
        for i in range(n):
            if i != r and M[i, c]:
                M[i] = (M[i] - M[i, c] * M[r]) % p
        piv.append(c)
        r += 1
    return M[:r], piv


def dense_kernel_normal_form(D, p):
    """Right-kernel basis of D in libspasm's normal form, from the dense RREF:
    for each free column j (ascending): k[j] = -1, k[pivot col of row a] = R[a][j].  Balanced values."""
    R, piv = dense_rref(D, p)
    m = np.asarray(D).shape[1]
    free = [j for j in range(m) if j not in set(piv)]
    K = np.zeros((len(free), m), dtype=object if p > (1 << 31) else np.int64)
    for f, j in enumerate(free):
        K[f, j] = -1
        for a, c in enumerate(piv):
            K[f, c] = R[a, j]
    K = np.mod(K, p)
    K = np.where(2 * K > p, K - p, K).astype(np.int64)
    return K, piv
