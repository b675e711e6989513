import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _leftmost_by_default(fn):
    """Most parity tests compare results (pivot columns, kernel basis, rref) that do not depend on how the rounds went only while
    every pivot is a leftmost entry: the engine and the oracle finish differently (dense / batched rounds vs sequential GPLU).
    They therefore run with enable_greedy_pivot_search=False unless they say otherwise; the tests of the "FL on columns" search
    pass enable_greedy_pivot_search=True themselves, with options under which both sides go through the same rounds."""
    import functools

    @functools.wraps(fn)
    def wrapped(A, *args, **kwargs):
        kwargs.setdefault("enable_greedy_pivot_search", False)
        return fn(A, *args, **kwargs)

    wrapped.__wrapped_leftmost__ = True
    return wrapped


@pytest.fixture(scope="session")
def S():
    import spasm_jl_amd
    from spasm_jl_amd import api

    if not getattr(api.echelonize, "__wrapped_leftmost__", False):
        api.echelonize = _leftmost_by_default(api.echelonize)  # (kernel(A), rank(A), blocks and sharded look it up there)
        spasm_jl_amd.echelonize = api.echelonize
    return spasm_jl_amd


@pytest.fixture(scope="session")
def O():
    import oracle_ffi

    oracle_ffi.lib()
    if not getattr(oracle_ffi.echelonize, "__wrapped_leftmost__", False):
        oracle_ffi.echelonize = _leftmost_by_default(oracle_ffi.echelonize)
    return oracle_ffi
