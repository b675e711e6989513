"""The device's GF(p) arithmetic (csrc/zp.hpp) against Python integers, edge values included.

Reference: src/SpaSM.jl:73-76 (field), :83-88 (balanced representatives), :383-390 (add, sub, mul, inverse, axpy by the
float-quotient formula).  Every result must be THE balanced residue in [p//2 - p + 1, p//2]."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PRIMES = [3, 127, 42013, 65521, 65537, 2**31 - 1, 0xFFFFFFFB]


def balanced(x, p):
    r = x % p
    return r - p if r > p // 2 else r


@pytest.mark.parametrize("p", PRIMES)
def test_device_field_arithmetic_edge_values(S, p):
    lib = S._abi.lib()
    half, mhalf = p // 2, p // 2 - p + 1
    edge = sorted({v for v in (0, 1, -1, 2, -2, half, half - 1, mhalf, mhalf + 1, half // 2, -(half // 2), 3, -3) if mhalf <= v <= half})
    rng = np.random.default_rng(p % 1000003)
    rnd = [int(v) for v in rng.integers(mhalf, half + 1, size=200)]
    vals = edge + rnd
    a, b, c = [], [], []
    for x in edge:
        for y in edge:
            a.append(x); b.append(y); c.append(edge[(len(a) * 7) % len(edge)])
    for i in range(0, len(rnd) - 2, 1):
        a.append(rnd[i]); b.append(rnd[i + 1]); c.append(rnd[i + 2])
    n = len(a)
    A = np.asarray(a, dtype=np.int32); B = np.asarray(b, dtype=np.int32); Cc = np.asarray(c, dtype=np.int32)
    out = np.empty(8 * n, dtype=np.int32)
    P = C.POINTER(C.c_int32)
    rc = lib.spasm_amd_zp_probe(p, n, A.ctypes.data_as(P), B.ctypes.data_as(P), Cc.ctypes.data_as(P), out.ctypes.data_as(P))
    assert rc == 0, S._abi.last_error()
    out = out.reshape(n, 8)
    for i in range(n):
        x, y, z = a[i], b[i], c[i]
        want = [balanced(x * y, p), balanced(x * y + z, p), balanced(x + y, p), balanced(x - y, p), balanced(-x, p),
                balanced(pow(x, -1, p), p) if x % p else 0, balanced(x * y, p), balanced(64 * x * y, p)]
        assert out[i].tolist() == want, (p, x, y, z, out[i].tolist(), want)
        assert all(mhalf <= v <= half for v in out[i].tolist())
