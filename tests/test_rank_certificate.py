"""Rank certificates (reference src/SpaSM.jl:345-353, :928-933; include/spasm_amd.h): verification is host-side and runs here,
against certificates built from the CPU oracle's factorization with an exact solve in Python; creation goes through the device
(echelonize with L + spasm_solve) and is tested on the GPU box."""
import hashlib
import os

import numpy as np
import pytest
from conftest import LM


def build_cert_by_hand(S, A, lu_p, lu_qinv, r, p, digest):
    """(i, j, x, y) of a certificate: x from the library's challenge (through a throw-away certificate file), y by a dense solve."""
    q = np.asarray(lu_qinv)
    jj = [int(np.nonzero(q == k)[0][0]) for k in range(r)]
    ii = [int(lu_p[k]) for k in range(r)]
    return ii, jj


def dense_solve_left(C, x, p):
    """y with y * C == x mod p (C r x r non-singular), exact integers."""
    r = len(x)
    M = [[int(C[k][c]) % p for k in range(r)] + [int(x[c]) % p] for c in range(r)]  # rows: equations sum_k y_k C[k][c] = x_c
    for c in range(r):
        piv = next(t for t in range(c, r) if M[t][c])
        M[c], M[piv] = M[piv], M[c]
        inv = pow(M[c][c], -1, p)
        M[c] = [(v * inv) % p for v in M[c]]
        for t in range(r):
            if t != c and M[t][c]:
                f = M[t][c]
                M[t] = [(a - f * b) % p for a, b in zip(M[t], M[c])]
    return [M[k][r] for k in range(r)]


@pytest.mark.parametrize("p", [127, 65521, 0xFFFFFFFB])
def test_verify_accepts_a_true_certificate_and_rejects_tampering(S, O, tmp_path, p):
    rng = np.random.default_rng(3)
    n, m = 40, 50
    D = ((rng.random((n, m)) < 0.15) * rng.integers(1, p, size=(n, m))).astype(np.int64)
    D[n - 1] = (D[0] + D[1]) % p
    A = S.CSR(D.T.copy(), prime=p)
    olu = O.echelonize(A, **LM)
    r = olu.r
    digest = hashlib.sha256(b"matrix file bytes").digest()
    ii, jj = build_cert_by_hand(S, A, olu.p, olu.qinv, r, p, digest)
    # the challenge: write a certificate file with x = y = 0, load it, and read the x the verifier expects from a failing verify?
    # -- simpler: the library exposes the challenge through certificate files only via creation, so compute y for the x that
    # spasm_certificate_rank_verify recomputes, by asking the library for it through a zero certificate and the ABI helper below.
    lib = S._abi.lib()
    import ctypes as C

    x = (C.c_int32 * max(r, 1))()
    ia = (C.c_int32 * max(r, 1))(*ii)
    ja = (C.c_int32 * max(r, 1))(*jj)
    lib.spasm_amd_certificate_challenge((C.c_uint8 * 32).from_buffer_copy(digest), p, r, ia, ja, x)
    xs = [int(v) for v in x[:r]]
    assert all(-(p // 2) - 1 <= v <= p // 2 for v in xs) and len(set(xs)) > 1
    Cm = [[int(D[i][j]) % p for j in jj] for i in ii]
    y = dense_solve_left(Cm, xs, p)
    bal = lambda v: v - p if v > p // 2 else v  # noqa: E731
    path = tmp_path / "cert.txt"
    with open(path, "w") as f:
        f.write("spasm-amd rank certificate v1\n%d %d\n%s\n" % (r, p, digest.hex()))
        for k in range(r):
            f.write("%d %d %d %d\n" % (ii[k], jj[k], xs[k], bal(y[k])))
    proof = S.rank_certificate_load(path)
    assert proof is not None and proof.r == r and proof.prime == p and proof.hash == digest
    assert S.certificate_rank_verify(A, digest, proof)
    # round trip through save
    path2 = tmp_path / "cert2.txt"
    S.rank_certificate_save(proof, path2)
    assert open(path).read() == open(path2).read()
    # tampering: another hash, a changed y, a repeated row, a wrong matrix
    assert not S.certificate_rank_verify(A, hashlib.sha256(b"other").digest(), proof)
    proof.y[0] = bal((int(proof.y[0]) + 1) % p)
    assert not S.certificate_rank_verify(A, digest, proof)
    proof.y[0] = bal(y[0])
    assert S.certificate_rank_verify(A, digest, proof)
    if r > 1:
        keep = int(proof.i[1])
        proof.i[1] = proof.i[0]
        assert not S.certificate_rank_verify(A, digest, proof)
        proof.i[1] = keep
    D2 = D.copy()
    D2[ii[0]][jj[0]] = (int(D2[ii[0]][jj[0]]) + 1) % p
    A2 = S.CSR(D2.T.copy(), prime=p)
    assert not S.certificate_rank_verify(A2, digest, proof)
    assert S.rank_certificate_load(os.devnull) is None


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,m,kw,p", [(1, 3000, 2500, dict(row_nnz=6), 65521), (0, 300, 400, dict(density=0.02), 0xFFFFFFFB), (2, 2000, 800, dict(row_nnz=30), 127)])
def test_certificate_created_on_the_device_verifies(S, O, tmp_path, kind, n, m, kw, p):
    A = S.synth_csr(kind, n, m, prime=p, seed=0xCE27, **kw)
    path = tmp_path / "A.sms"
    S.save(path, A)
    B, digest = S.load(path, prime=p, get_hash=True)
    assert digest == hashlib.sha256(open(path, "rb").read()).digest()
    fact = S.echelonize(B)  # default options: any factorization will do
    proof = S.certificate_rank_create(B, digest, fact)
    assert proof.r == fact.r == O.echelonize(B, **LM).r
    assert S.certificate_rank_verify(B, digest, proof)
    assert sorted(proof.j.tolist()) == sorted(np.nonzero(np.asarray(fact.qinv) >= 0)[0].tolist())
    S.rank_certificate_save(proof, tmp_path / "c.txt")
    again = S.rank_certificate_load(tmp_path / "c.txt")
    assert S.certificate_rank_verify(B, digest, again)
    # rank(A) <= r is the other half: the reference's self-check of the factorization
    assert S.factorization_verify(B, fact, 9)


def test_load_refuses_a_row_count_the_file_cannot_hold(S, tmp_path):
    """r is read from the file: a count that the rest of the file has no room for is refused before memory is set aside for it"""
    path = tmp_path / "huge.txt"
    with open(path, "w") as f:
        f.write("spasm-amd rank certificate v1\n%d %d\n%s\n" % (2_000_000_000, 42013, "00" * 32))
        f.write("0 0 1 1\n")
    assert S.rank_certificate_load(path) is None
