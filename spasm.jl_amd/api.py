"""Host-side mirror of SpaSM.jl's CSR / echelonize / kernel surface over the C ABI.

Julia is absent from this image, so the host side above the C ABI is written in Python with the
reference's names, argument meaning and error behaviour (reference src/SpaSM.jl, lines cited per
item).  Everything that computes goes through libspasm_amd.so; nothing here falls back to Python.
"""
import ctypes as C

import numpy as np

from . import _abi

prime0 = 42013  # reference src/SpaSM.jl:16


class SpasmError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------------
# Field / ZZp  (reference src/SpaSM.jl:51-121)
# ---------------------------------------------------------------------------------------------
class Field:
    """Field(p): the finite field Z/pZ, 2 < p <= 0xfffffffb (reference src/SpaSM.jl:73-76)."""

    def __init__(self, p=prime0):
        p = int(p)
        if not (2 < p <= 0xFFFFFFFB):
            raise AssertionError("2 < p <= 0xfffffffb")  # the @assert at :74
        self.p = p
        self.halfp = p // 2
        self.mhalfp = p // 2 - p + 1
        self.dinvp = 1.0 / p

    def __call__(self, x):
        """Balanced representative of x (reference src/SpaSM.jl:83-88, :96)."""
        x = int(x) % self.p
        return x - self.p if x > self.halfp else x

    def __eq__(self, other):
        return isinstance(other, Field) and other.p == self.p

    def __hash__(self):
        return hash(self.p)

    def __repr__(self):
        return f"Field({self.p})"


def ZZp(F, x=None):
    """ZZp(F, x) / ZZp(p, x) / ZZp(x): balanced representative (reference src/SpaSM.jl:96-98)."""
    if x is None:
        return Field(prime0)(F)
    if not isinstance(F, Field):
        F = Field(F)
    return F(x)


def balanced(values, p):
    """Vectorised ZZp: integers -> balanced residues (reference src/SpaSM.jl:955-958)."""
    v = np.mod(np.asarray(values, dtype=np.int64), p)
    return np.where(2 * v > p, v - p, v).astype(np.int32)


# ---------------------------------------------------------------------------------------------
# CSR  (reference src/SpaSM.jl:126-167, :941-1023)
# ---------------------------------------------------------------------------------------------
class CSR:
    """Spasm matrix in Compressed Sparse Row format, owned through the C ABI.

    CSR(m, prime) with a scipy sparse matrix or a 2-D array stores the TRANSPOSE: every column of
    `m` becomes a row (reference src/SpaSM.jl:941-968, README.md:7).  `transpose=False` stores `m`
    itself (one extra spasm_transpose, :967).
    """

    def __init__(self, src, prime=prime0, transpose=True, own=True):
        self._own = own
        if isinstance(src, C.POINTER(_abi.CsrStruct)):
            if not src:
                raise SpasmError("NULL spasm_csr: " + _abi.last_error())
            self.data = src
            return
        import scipy.sparse as sp

        A = sp.csc_matrix(src)
        if A.dtype.kind not in "iu":
            raise TypeError("integer entries expected")
        A.sum_duplicates()
        nrow_j, ncol_j = A.shape
        vals = balanced(A.data, prime)
        keep = vals != 0  # zeros are dropped (:959)
        col_of = np.repeat(np.arange(ncol_j), np.diff(A.indptr))
        counts = np.bincount(col_of[keep], minlength=ncol_j).astype(np.int64)
        nnz_ = int(keep.sum())
        ptr = _abi.lib().spasm_csr_alloc(ncol_j, nrow_j, nnz_, int(prime), True)  # csr_alloc(n,m,nzmax,prime) :944
        if not ptr:
            raise SpasmError("spasm_csr_alloc failed: " + _abi.last_error())
        self.data = ptr
        st = ptr.contents
        p = np.ctypeslib.as_array(st.p, (ncol_j + 1,))
        p[0] = 0
        np.cumsum(counts, out=p[1:])
        if nnz_:
            np.ctypeslib.as_array(st.j, (nnz_,))[:] = A.indices[keep]
            np.ctypeslib.as_array(st.x, (nnz_,))[:] = vals[keep]
        if not transpose:
            t = _abi.lib().spasm_transpose(self.data)
            if not t:
                raise SpasmError("spasm_transpose failed: " + _abi.last_error())
            _abi.lib().spasm_csr_free(self.data)
            self.data = t

    @classmethod
    def from_rows(cls, rows, m, prime=prime0):
        """Build directly from libspasm-side rows: rows[i] = list of (column, value)."""
        n = len(rows)
        nnz_ = sum(len(r) for r in rows)
        ptr = _abi.lib().spasm_csr_alloc(n, m, nnz_, int(prime), True)
        if not ptr:
            raise SpasmError("spasm_csr_alloc failed: " + _abi.last_error())
        self = cls(ptr)
        p, j, x = self.p, self.j, self.x
        k = 0
        F = Field(prime)
        for i, r in enumerate(rows):
            p[i] = k
            for (c, v) in r:
                j[k] = c
                x[k] = F(v)
                k += 1
        p[n] = k
        return self

    @classmethod
    def from_arrays(cls, n, m, p, j, x, prime=prime0):
        """Copy host CSR arrays (0-based, values already balanced) into an owned spasm_csr."""
        nnz_ = int(p[n])
        ptr = _abi.lib().spasm_csr_alloc(int(n), int(m), nnz_, int(prime), True)
        if not ptr:
            raise SpasmError("spasm_csr_alloc failed: " + _abi.last_error())
        self = cls(ptr)
        self.p[:] = np.asarray(p, dtype=np.int64)[: n + 1]
        if nnz_:
            self.j[:nnz_] = np.asarray(j, dtype=np.int32)[:nnz_]
            self.x[:nnz_] = np.asarray(x, dtype=np.int32)[:nnz_]
        return self

    def __del__(self):  # finalizer(csr_free, x), reference src/SpaSM.jl:146-150
        if getattr(self, "_own", False) and getattr(self, "data", None):
            try:
                _abi.lib().spasm_csr_free(self.data)
            except Exception:
                pass
            self.data = None

    # getproperty, reference src/SpaSM.jl:154-167: views, no copies
    @property
    def _st(self):
        return self.data.contents

    @property
    def n(self):
        return int(self._st.n)

    @property
    def m(self):
        return int(self._st.m)

    @property
    def nzmax(self):
        return int(self._st.nzmax)

    @property
    def prime(self):
        return int(self._st.field.p)

    @property
    def field(self):
        return Field(self.prime)

    @property
    def p(self):
        return np.ctypeslib.as_array(self._st.p, (self.n + 1,))

    @property
    def j(self):
        return np.ctypeslib.as_array(self._st.j, (max(self.nzmax, 1),))

    @property
    def x(self):
        return np.ctypeslib.as_array(self._st.x, (max(self.nzmax, 1),))

    @property
    def shape(self):  # Base.size, :229
        return (self.n, self.m)

    def __repr__(self):  # Base.show, :195
        return f"{self.n}×{self.m} CSR matrix % {self.prime} with {nnz(self)} (maximum {self.nzmax}) non-zeros"

    def rows(self):
        """libspasm-side rows as sorted lists of (column, balanced value)."""
        p, j, x = self.p, self.j, self.x
        out = []
        for i in range(self.n):
            lo, hi = int(p[i]), int(p[i + 1])
            out.append(sorted(zip(j[lo:hi].tolist(), x[lo:hi].tolist())))
        return out

    def todense(self):
        """Dense libspasm-side matrix (n x m) of balanced residues."""
        D = np.zeros((self.n, self.m), dtype=np.int64)
        p, j, x = self.p, self.j, self.x
        for i in range(self.n):
            lo, hi = int(p[i]), int(p[i + 1])
            D[i, j[lo:hi]] = x[lo:hi]
        return D


def nnz(A):
    """SparseArrays.nnz (reference src/SpaSM.jl:432)."""
    return int(_abi.lib().spasm_nnz(A.data))


def sparse(A, transpose=True):
    """SparseMatrixCSC view of a CSR: column i = row i of the CSR, sorted (reference src/SpaSM.jl:1011-1023)."""
    import scipy.sparse as sp

    n, m = A.shape
    k = nnz(A)
    mat = sp.csc_matrix((A.x[:k].astype(np.int64), A.j[:k].astype(np.int64), A.p.astype(np.int64)), shape=(m, n))
    mat.sort_indices()
    return mat if transpose else mat.T.tocsc()


def transpose(A):
    """Base.transpose(::CSR) (reference src/SpaSM.jl:589)."""
    t = _abi.lib().spasm_transpose(A.data)
    if not t:
        raise SpasmError("spasm_transpose failed: " + _abi.last_error())
    return CSR(t)


# ---------------------------------------------------------------------------------------------
# LU / echelonize / kernel / rank  (reference src/SpaSM.jl:262-305, :814-884, :1147-1149)
# ---------------------------------------------------------------------------------------------
class EchelonizeOpts:
    """EchelonizeOpts() filled by spasm_echelonize_init_opts (reference src/SpaSM.jl:817)."""

    def __init__(self, **kwargs):
        self.struct = _abi.EchelonizeOptsStruct()
        _abi.lib().spasm_echelonize_init_opts(C.byref(self.struct))
        for k, v in kwargs.items():  # parse_echelonize_opts, :819-824
            if not hasattr(self.struct, k):
                raise AttributeError(f"type EchelonizeOpts has no field {k}")
            setattr(self.struct, k, v)

    def __getattr__(self, k):
        return getattr(self.__dict__["struct"], k)


class LU:
    def __init__(self, ptr):
        if not ptr:
            raise SpasmError("spasm_echelonize failed: " + _abi.last_error())
        self.data = ptr

    @classmethod
    def from_parts(cls, U, qinv, p, L=None):
        """An LU from its parts (layout reference src/SpaSM.jl:262-270), e.g. assembled from the rounds of the row-sharded
        echelonize.  U (and L, when given): CSRs whose ownership passes to the LU; qinv: m entries (row of U or -1); p: max(n, m)
        entries.  Everything is malloc'ed so that spasm_lu_free releases it like an LU of spasm_echelonize."""
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        libc.malloc.argtypes = [C.c_size_t]
        qinv = np.ascontiguousarray(qinv, dtype=np.int32)
        p = np.ascontiguousarray(p, dtype=np.int32)

        def dup(a):
            mem = libc.malloc(max(a.nbytes, 4))
            if not mem:
                raise MemoryError("malloc")
            C.memmove(mem, a.ctypes.data, a.nbytes)
            return C.cast(mem, C.POINTER(C.c_int32))

        mem = libc.malloc(C.sizeof(_abi.LuStruct))
        if not mem:
            raise MemoryError("malloc")
        st = C.cast(mem, C.POINTER(_abi.LuStruct))
        st.contents.r = int(U.n)
        st.contents.complete = False
        st.contents.L = L.data if L is not None else None
        if L is not None:
            L._own = False
        st.contents.U = U.data
        st.contents.qinv = dup(qinv)
        st.contents.p = dup(p)
        st.contents.Ltmp = None
        U._own = False  # now owned by the LU
        return cls(st)

    def __del__(self):  # finalizer(lu_free, x), reference src/SpaSM.jl:273-277
        if getattr(self, "data", None):
            try:
                _abi.lib().spasm_lu_free(self.data)
            except Exception:
                pass
            self.data = None

    @property
    def r(self):
        return int(self.data.contents.r)

    @property
    def complete(self):
        return bool(self.data.contents.complete)

    @property
    def U(self):
        st = self.data.contents
        if not st.U:
            raise SpasmError("M.U is null")  # :291
        return CSR(st.U, own=False)

    @property
    def L(self):
        st = self.data.contents
        if not st.L:
            raise SpasmError("M.L is null")  # :288
        return CSR(st.L, own=False)

    @property
    def qinv(self):
        st = self.data.contents
        if not st.qinv:
            raise SpasmError("M.qinv is null")
        return np.ctypeslib.as_array(st.qinv, (max(int(st.U.contents.m), 1),))[: int(st.U.contents.m)]

    @property
    def p(self):
        st = self.data.contents
        if not st.p:
            raise SpasmError("M.p is null")
        return np.ctypeslib.as_array(st.p, (max(int(st.U.contents.m), 1),))[: int(st.U.contents.m)]


def echelonize(A, opts=None, verbose=False, **kwargs):
    """echelonize(A; kwargs...) -> LU (reference src/SpaSM.jl:860-866)."""
    if opts is None:
        opts = EchelonizeOpts()
    for k, v in kwargs.items():
        if not hasattr(opts.struct, k):
            raise AttributeError(f"type EchelonizeOpts has no field {k}")
        setattr(opts.struct, k, v)
    with _quiet(not verbose):
        ptr = _abi.lib().spasm_echelonize(A.data, C.byref(opts.struct))
    return LU(ptr)


def echelonize_multi(A, nshards, opts=None, verbose=False, **kwargs):
    """echelonize over `nshards` row shards on the devices of this process (spasm_amd_echelonize_multi; engine extension): the
    LU of echelonize(A; enable_greedy_pivot_search=false) whatever nshards is -- same rank, pivot columns and kernel."""
    if opts is None:
        opts = EchelonizeOpts()
    for k, v in kwargs.items():
        if not hasattr(opts.struct, k):
            raise AttributeError(f"type EchelonizeOpts has no field {k}")
        setattr(opts.struct, k, v)
    with _quiet(not verbose):
        ptr = _abi.lib().spasm_amd_echelonize_multi(A.data, C.byref(opts.struct), int(nshards))
    return LU(ptr)


def kernel(A, verbose=False, **kwargs):
    """kernel(fact::LU) / kernel(A::CSR) (reference src/SpaSM.jl:876-882, :1147)."""
    fact = A if isinstance(A, LU) else echelonize(A, verbose=verbose, **kwargs)
    with _quiet(not verbose):
        ptr = _abi.lib().spasm_kernel(fact.data)
    if not ptr:
        raise SpasmError("spasm_kernel failed: " + _abi.last_error())
    return CSR(ptr)


def scatter(A, i, beta, x):
    """scatter(A, i, beta, x): x += beta * A[i] (reference src/SpaSM.jl:619-620; i 0-based here, as on the C side).  x: int32
    array of m balanced residues, updated in place."""
    assert x.dtype == np.int32 and x.flags["C_CONTIGUOUS"] and len(x) >= A.m and 0 <= i < A.n
    _abi.lib().spasm_scatter(A.data, int(i), int(beta), x.ctypes.data_as(C.POINTER(C.c_int32)))
    return x


def sparse_triangular_solve_row(U, B, k, xj, x, qinv, verbose=False):
    """sparse_triangular_solve(U, B, k, xj, x, qinv) (reference src/SpaSM.jl:694-722; k 0-based): solve x * U = B[k].  xj: int32,
    3 m entries, zero on entry; x: int32, m entries.  Returns top: the pattern of the solution is xj[top:m], its values x[xj[...]];
    with x_b on the pivot columns and x_a on the others, x_b * U + x_a == B[k]."""
    m = U.m
    assert m == B.m == len(qinv) and 0 <= k < B.n                     # the reference's assertions (:715-720)
    assert xj.dtype == np.int32 and len(xj) >= 3 * m and not xj.any()
    assert x.dtype == np.int32 and len(x) >= m
    q = np.ascontiguousarray(qinv, dtype=np.int32)
    with _quiet(not verbose):
        top = _abi.lib().spasm_sparse_triangular_solve(U.data, B.data, int(k), xj.ctypes.data_as(C.POINTER(C.c_int32)),
                                                        x.ctypes.data_as(C.POINTER(C.c_int32)), q.ctypes.data_as(C.POINTER(C.c_int32)))
    if top < 0:
        raise SpasmError("spasm_sparse_triangular_solve failed: " + _abi.last_error())
    return int(top)


def sparse_triangular_solve(U, B, qinv=None, verbose=False):
    """sparse_triangular_solve(LU, B) / sparse_triangular_solve(U, B, qinv) (reference src/SpaSM.jl:725-755): solve X * U == B in
    sparse matrices; returns X (rows of B x rows of U), or None if some row of B has no solution.  One device pass over all
    rows of B (spasm_amd_triangular_solve)."""
    if isinstance(U, LU):
        U, qinv = U.U, U.qinv
    X, ok = _triangular_solve(U, B, qinv, verbose)
    return X if bool(ok.all()) else None


def _triangular_solve(U, B, qinv, verbose=False):
    """(X, ok): x_b of every row of B and whether its x_a is empty (reference src/SpaSM.jl:694-713)"""
    q = np.ascontiguousarray(qinv, dtype=np.int32)
    ok = np.zeros(max(B.n, 1), dtype=np.uint8)
    with _quiet(not verbose):
        ptr = _abi.lib().spasm_amd_triangular_solve(U.data, q.ctypes.data_as(C.POINTER(C.c_int32)), B.data, ok.ctypes.data_as(C.POINTER(C.c_ubyte)))
    if not ptr:
        raise SpasmError("spasm_amd_triangular_solve failed: " + _abi.last_error())
    return CSR(ptr), ok[: B.n].astype(bool)


def rref(fact, verbose=False):
    """rref(fact) -> (R, Rqinv) (reference src/SpaSM.jl:871): reduced row echelon form of fact.U; Rqinv[j] = row of R whose
    pivot is column j, or -1."""
    m = fact.U.m
    rq = np.full(max(m, 1), -1, dtype=np.int32)
    with _quiet(not verbose):
        ptr = _abi.lib().spasm_rref(fact.data, rq.ctypes.data_as(C.POINTER(C.c_int32)))
    if not ptr:
        raise SpasmError("spasm_rref failed: " + _abi.last_error())
    return CSR(ptr), rq[:m]


def gesv(fact, B, verbose=False):
    """gesv(fact::LU, B::CSR) (reference src/SpaSM.jl:907-923): solve X * A == B where A has been echelonized as `fact` WITH its
    L factor (echelonize(A, L=True)).  Returns (X, ok): X is rows(B) x rows(A); ok[k] says whether row k of B has a solution."""
    ok = np.zeros(max(B.n, 1), dtype=np.uint8)
    with _quiet(not verbose):
        ptr = _abi.lib().spasm_gesv(fact.data, B.data, ok.ctypes.data)
    if not ptr:
        raise SpasmError("spasm_gesv failed: " + _abi.last_error())
    return CSR(ptr), ok[: B.n].astype(bool)


def solve(fact, b, x=None):
    """solve(fact::LU, b::Vector) (reference src/SpaSM.jl:889-905): x with x * A == b, or None when there is none.  b has one entry
    per column of A, x one per ROW OF A (the prototype sizes x by fact.U.n; see include/spasm_amd.h)."""
    st = fact.data.contents
    if not st.L:
        raise SpasmError("M.L is null")  # fact.L, :896
    m, n = int(st.U.contents.m), int(st.L.contents.n)
    b = np.ascontiguousarray(b, dtype=np.int32)
    if b.shape != (m,):
        raise ValueError(f"b must have {m} entries")
    if x is None:
        x = np.zeros(n, dtype=np.int32)
    elif x.shape != (n,) or x.dtype != np.int32:
        raise ValueError(f"x must be an int32 vector of {n} entries")
    okv = _abi.lib().spasm_solve(fact.data, b.ctypes.data, x.ctypes.data)
    if not okv and _abi.last_error():
        raise SpasmError("spasm_solve failed: " + _abi.last_error())
    return x if okv else None


def factorization_verify(A, fact, seed=0):
    """factorization_verify(A, fact, seed) (reference src/SpaSM.jl:934): probabilistic self-check, on the host, that the row
    space of A lies in the span of fact.U and that U has echelon shape (so rank(A) <= fact.r).  With L = NULL the converse
    inclusion is not checked (include/spasm_amd.h)."""
    return bool(_abi.lib().spasm_factorization_verify(A.data, fact.data, int(seed) & 0xFFFFFFFFFFFFFFFF))


def rank(A, rank_only=False, verbose=False, **kwargs):
    """rank(N::LU) = N.r; rank(A::CSR) = rank(echelonize(A)) (reference src/SpaSM.jl:305, :1149).  rank_only=True (engine extension,
    spasm_amd_rank): the rows of U are counted on the device and never assembled on the host -- for matrices whose U outgrows it."""
    if isinstance(A, LU):
        return A.r
    if not rank_only:
        return echelonize(A, verbose=verbose, **kwargs).r
    opts = EchelonizeOpts()
    for k, v in kwargs.items():
        if not hasattr(opts.struct, k):
            raise AttributeError(f"type EchelonizeOpts has no field {k}")
        setattr(opts.struct, k, v)
    with _quiet(not verbose):
        r = _abi.lib().spasm_amd_rank(A.data, C.byref(opts.struct))
    if r < 0:
        raise SpasmError("spasm_amd_rank failed: " + _abi.last_error())
    return int(r)


def last_rounds(max_rounds=4096):
    """Per-round records of the most recent echelonize call on this thread (engine extension)."""
    buf = (_abi.RoundStats * max_rounds)()
    n = _abi.lib().spasm_amd_last_rounds(buf, max_rounds)
    return [buf[i].as_dict() for i in range(min(n, max_rounds))]


# The engine's progress text goes to stderr / logcallback like libspasm's; the reference hides it
# by redirecting the process's stderr around the ccall unless verbose (src/SpaSM.jl:838-858).
_LOGFUNC = C.CFUNCTYPE(C.c_int, C.c_char_p)
_swallow = _LOGFUNC(lambda s: 0)


class _quiet:
    def __init__(self, active):
        self.active = active

    def __enter__(self):
        if self.active:
            self.slot = C.c_void_p.in_dll(_abi.lib(), "logcallback")
            self.prev = self.slot.value
            self.slot.value = C.cast(_swallow, C.c_void_p).value

    def __exit__(self, *exc):
        if self.active:
            self.slot.value = self.prev
        return False


def synth_csr(kind, n, m, density=0.0, row_nnz=0, prime=prime0, seed=0):
    """Deterministic synthetic CSR: kind 0 = Bernoulli(density), kind 1 = row_nnz per row."""
    ptr = _abi.lib().spasm_amd_synth_csr(int(kind), int(n), int(m), float(density), int(row_nnz), int(prime), int(seed))
    if not ptr:
        raise SpasmError("spasm_amd_synth_csr failed: " + _abi.last_error())
    return CSR(ptr)


# ---------------------------------------------------------------------------------------------
# Triplets and the SMS wire format  (reference src/SpaSM.jl:234-260, :482-529, :1025-1086)
# ---------------------------------------------------------------------------------------------
_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]


class Triplet:
    """Spasm matrix in coordinate format; `push` then `compress` (reference src/SpaSM.jl:244-251, :482-493)."""

    def __init__(self, n=0, m=0, nzmax=16, prime=prime0, ptr=None):
        self.data = ptr if ptr is not None else _abi.lib().spasm_triplet_alloc(int(n), int(m), int(nzmax), int(prime), True)
        if not self.data:
            raise SpasmError("spasm_triplet_alloc failed: " + _abi.last_error())

    def __del__(self):
        if getattr(self, "data", None):
            try:
                _abi.lib().spasm_triplet_free(self.data)
            except Exception:
                pass
            self.data = None

    def push(self, i, j, x):
        """push!(A, (i, j, x)) with 1-based indices (reference src/SpaSM.jl:482-488)."""
        assert 1 <= i and 1 <= j
        _abi.lib().spasm_add_entry(self.data, int(i) - 1, int(j) - 1, int(x))
        return self

    def transpose_(self):
        _abi.lib().spasm_triplet_transpose(self.data)
        return self

    @property
    def nz(self):
        return int(self.data.contents.nz)

    @property
    def shape(self):
        return (int(self.data.contents.n), int(self.data.contents.m))

    def compress(self):
        ptr = _abi.lib().spasm_compress(self.data)
        if not ptr:
            raise SpasmError("spasm_compress failed: " + _abi.last_error())
        return CSR(ptr)


def load(path, prime=prime0, csr=True, get_hash=False):
    """load(File{format"SMS"}(path); prime, csr) (reference src/SpaSM.jl:498-512): Triplet, or CSR when csr=True."""
    f = _libc.fopen(str(path).encode(), b"r")
    if not f:
        raise OSError(f"cannot open {path}")
    try:
        digest = (C.c_uint8 * 32)() if get_hash else None
        ptr = _abi.lib().spasm_triplet_load(f, int(prime), digest)
    finally:
        _libc.fclose(f)
    if not ptr:
        raise SpasmError("spasm_triplet_load failed: " + _abi.last_error())
    T = Triplet(ptr=ptr)
    out = T.compress() if csr else T
    return (out, bytes(digest)) if get_hash else out


def save(path, A):
    """save(File{format"SMS"}(path), A) for a CSR or a Triplet (reference src/SpaSM.jl:514-529)."""
    f = _libc.fopen(str(path).encode(), b"w")
    if not f:
        raise OSError(f"cannot open {path}")
    try:
        if isinstance(A, Triplet):
            _abi.lib().spasm_triplet_save(A.data, f)
        else:
            _abi.lib().spasm_csr_save(A.data, f)
    finally:
        _libc.fclose(f)


# ---------------------------------------------------------------------------------------------
# Rank certificates  (reference src/SpaSM.jl:345-353, :928-933)
# ---------------------------------------------------------------------------------------------
class RankCertificate:
    """RankCertificate{F}: r rows i and r columns j of A whose submatrix is shown non-singular by y * A[i, j] == x for the
    challenge x drawn from (hash, prime, r, i, j) -- a proof that rank(A) >= r (include/spasm_amd.h)."""

    def __init__(self, ptr, own=True):
        assert ptr
        self.data = ptr
        self._own = own

    def __del__(self):
        if getattr(self, "data", None) and self._own:
            try:
                _abi.lib().spasm_rank_certificate_free(self.data)
            except Exception:
                pass
            self.data = None

    r = property(lambda s: int(s.data.contents.r))
    prime = property(lambda s: int(s.data.contents.prime))
    hash = property(lambda s: bytes(s.data.contents.hash))

    def _arr(self, name):
        r = self.r
        return np.ctypeslib.as_array(getattr(self.data.contents, name), (max(r, 1),))[:r]

    i = property(lambda s: s._arr("i"))
    j = property(lambda s: s._arr("j"))
    x = property(lambda s: s._arr("x"))
    y = property(lambda s: s._arr("y"))


def _hash_arg(h):
    h = bytes(h)
    assert len(h) == 32, "the hash is the 32-byte SHA-256 digest load(..., get_hash=True) returns"
    return (C.c_uint8 * 32).from_buffer_copy(h)


def certificate_rank_create(A, hash, fact):
    """certificate_rank_create(A, hash, fact) (reference src/SpaSM.jl:928)."""
    ptr = _abi.lib().spasm_certificate_rank_create(A.data, _hash_arg(hash), fact.data)
    if not ptr:
        raise SpasmError("spasm_certificate_rank_create failed: " + _abi.last_error())
    return RankCertificate(ptr)


def certificate_rank_verify(A, hash, proof):
    """certificate_rank_verify(A, hash, proof) -> Bool (reference src/SpaSM.jl:930): host-side, O(nnz(A))."""
    return bool(_abi.lib().spasm_certificate_rank_verify(A.data, _hash_arg(hash), proof.data))


def rank_certificate_save(proof, path):
    """rank_certificate_save(proof, file) (reference src/SpaSM.jl:931)."""
    f = _libc.fopen(str(path).encode(), b"w")
    if not f:
        raise OSError(f"cannot open {path}")
    try:
        _abi.lib().spasm_rank_certificate_save(proof.data, f)
    finally:
        _libc.fclose(f)


def rank_certificate_load(path):
    """rank_certificate_load(file, proof) (reference src/SpaSM.jl:933): a RankCertificate, or None when the file does not parse."""
    f = _libc.fopen(str(path).encode(), b"r")
    if not f:
        raise OSError(f"cannot open {path}")
    libc = C.CDLL(None)
    libc.calloc.restype = C.c_void_p
    libc.calloc.argtypes = [C.c_size_t, C.c_size_t]
    mem = libc.calloc(1, C.sizeof(_abi.RankCertificateStruct))
    ptr = C.cast(mem, C.POINTER(_abi.RankCertificateStruct))
    try:
        ok = _abi.lib().spasm_rank_certificate_load(f, ptr)
    finally:
        _libc.fclose(f)
    if not ok:
        _abi.lib().spasm_rank_certificate_free(ptr)
        return None
    return RankCertificate(ptr)
