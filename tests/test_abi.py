"""CPU tests of the drop-in boundary: struct layouts, exported symbols, ownership helpers, generator.
No compute entry point is called here (there is no GPU on the CPU runner)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
from conftest import LM  # leftmost-entry pivots only: what these tests compare does not depend on how the rounds went then

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_struct_layouts_match_julia_mirrors(S):
    """Sizes/offsets implied by reference src/SpaSM.jl:51-56,126-134,262-270,325-343."""
    abi = S._abi
    for name, (size, offsets) in abi.EXPECTED_LAYOUT.items():
        T = getattr(abi, name)
        assert C.sizeof(T) == size, name
        for field, off in offsets.items():
            assert getattr(T, field).offset == off, (name, field)


def test_library_exports_every_declared_symbol(S):
    """Every function declared in include/spasm_amd.h is exported by libspasm_amd.so and bound."""
    header = open(os.path.join(ROOT, "include", "spasm_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(spasm_[a-z_0-9A-Z]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = S._abi.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in S._abi.SIGNATURES, f"{name} has no ctypes signature"
    for name in S._abi.DATA_SYMBOLS:
        C.c_void_p.in_dll(lib, name)


def test_alloc_nnz_free_roundtrip(S):
    A = S.CSR(np.array([[1, 2, 0, 0], [0, 0, 0, 0], [0, 0, 3, 4]]))
    assert A.shape == (4, 3)  # CSR(m) stores the transpose, reference src/SpaSM.jl:941-968
    assert S.nnz(A) == 4
    assert A.prime == 42013 and A._st.field.halfp == 21006 and A._st.field.mhalfp == -21006
    assert abs(A._st.field.dinvp - 1 / 42013) < 1e-18
    # sparse(CSR(m)) == ZZp.(m), reference test/runtests.jl:7-10
    assert (S.sparse(A).toarray() % 42013 == np.array([[1, 2, 0, 0], [0, 0, 0, 0], [0, 0, 3, 4]])).all()


def test_balanced_representatives(S):
    F = S.Field(42013)
    assert (F.halfp, F.mhalfp) == (21006, -21006)  # reference src/SpaSM.jl:73-76
    assert S.ZZp(F, 42012) == -1 and S.ZZp(F, 21006) == 21006 and S.ZZp(F, 21007) == -21006
    assert S.ZZp(3) == 3
    with pytest.raises(AssertionError):
        S.Field(2)
    with pytest.raises(AssertionError):
        S.Field(0xFFFFFFFC)


def test_values_are_reduced_and_zeros_dropped(S):
    A = S.CSR(np.array([[42013, 42014], [-1, 21007]]))  # 42013 = 0 is dropped, reference src/SpaSM.jl:955-959
    assert A.rows() == [[(1, -1)], [(0, 1), (1, -21006)]]


def test_init_opts_defaults(S):
    o = S.EchelonizeOpts()
    assert o.enable_greedy_pivot_search and o.enable_dense and o.enable_GPLU and not o.L and not o.complete
    assert o.max_round == 3 and o.min_pivot_proportion == 0.1 and o.dense_block_size == 1000
    with pytest.raises(AttributeError):
        S.EchelonizeOpts(no_such_field=1)
    assert S.EchelonizeOpts(max_round=7).max_round == 7  # kwargs override, reference src/SpaSM.jl:819-824


def test_synth_is_deterministic_and_well_formed(S):
    A = S.synth_csr(1, 500, 700, row_nnz=20, prime=65521, seed=0x5A5A0003)
    B = S.synth_csr(1, 500, 700, row_nnz=20, prime=65521, seed=0x5A5A0003)
    assert (A.p == B.p).all() and (A.j[:10000] == B.j[:10000]).all() and (A.x[:10000] == B.x[:10000]).all()
    assert (np.diff(A.p) == 20).all()
    for r in A.rows()[:50]:
        cols = [c for c, _ in r]
        assert len(set(cols)) == 20 and min(cols) >= 0 and max(cols) < 700
        assert all(v != 0 and -32760 <= v <= 32760 for _, v in r)
    C2 = S.synth_csr(0, 2000, 2000, density=1e-2, prime=42013, seed=0x5A5A0002)
    assert abs(S.nnz(C2) / (2000 * 2000) - 1e-2) < 1e-3
    # rows are not sorted by column
    j = A.j[: S.nnz(A)].reshape(500, 20)
    assert (np.diff(j, axis=1) < 0).any()


def test_hot_path_fails_loudly_without_gpu(S):
    if S._abi.lib().spasm_amd_device_count() > 0:
        pytest.skip("a GPU is present")
    A = S.CSR(np.array([[1, 2], [3, 6]]))
    with pytest.raises(S.SpasmError, match="no HIP device"):
        S.echelonize(A, **LM)
    with pytest.raises(S.SpasmError, match="no HIP device"):
        S.transpose(A)


def test_dense_shard_steps_out_of_order_are_errors_not_crashes():
    """The per-process steps of the dense finish over row shards (spasm_amd_dshard_*) without a handle: an error code and a message."""
    import ctypes as C

    from spasm_jl_amd import _abi

    lib = _abi.lib()
    assert lib.spasm_amd_dshard_block_begin(None) == -1 and "dense shard" in _abi.last_error()
    assert lib.spasm_amd_dshard_apply(None, 0, 0, 64, 64) == -1
    assert lib.spasm_amd_dshard_finish(None) == -1
    assert lib.spasm_amd_dshard_pack(None, 0, None) == -1
    assert not lib.spasm_amd_dshard_fetch_U(None, None, None, None)
    assert not lib.spasm_amd_dshard_open(None, 0, 2) and "null plan" in _abi.last_error()
    lib.spasm_amd_dshard_close(None)
