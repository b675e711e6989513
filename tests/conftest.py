import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Passed explicitly (`**LM`) by the tests whose comparisons hold entry for entry only while every pivot is a leftmost entry: rank,
# pivot columns, kernel basis and rref then do not depend on how the rounds went (the engine and the oracle finish differently:
# dense / batched rounds vs sequential GPLU).  The library's own default -- the reference's, enable_greedy_pivot_search = 1,
# src/SpaSM.jl:326 -- is what tests/test_gpu_default_options.py runs the same groups with.
LM = {"enable_greedy_pivot_search": False}


@pytest.fixture(scope="session")
def S():
    import spasm_jl_amd

    return spasm_jl_amd


@pytest.fixture(scope="session")
def O():
    import oracle_ffi

    oracle_ffi.lib()
    return oracle_ffi
