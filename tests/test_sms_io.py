"""The SMS wire format and the triplet container: host-side code, CPU tests.
Format: header "n m M", 1-based "i j v" lines, terminator "0 0 0" (reference src/SpaSM.jl:1029-1042, :1063-1086)."""
import hashlib

import numpy as np
import pytest


def test_triplet_push_compress(S):
    T = S.Triplet(0, 0, prime=42013)
    T.push(1, 1, 1).push(1, 2, 3).push(2, 1, 2).push(2, 2, 6)
    T.push(2, 2, 42013)  # reduces to zero: not stored
    assert T.nz == 4 and T.shape == (2, 2)
    A = T.compress()
    assert A.rows() == [[(0, 1), (1, 3)], [(0, 2), (1, 6)]]
    # repeated positions are summed, cancelled ones dropped
    T2 = S.Triplet(3, 3, prime=7)
    for (i, j, v) in [(1, 1, 3), (3, 2, 5), (1, 1, 4), (1, 3, 2), (3, 2, 1)]:
        T2.push(i, j, v)
    assert T2.compress().rows() == [[(2, 2)], [], [(1, -1)]]
    T2.transpose_()
    assert T2.shape == (3, 3) and T2.compress().rows() == [[], [(2, -1)], [(0, 2)]]


def test_sms_text_format_and_roundtrip(S, tmp_path):
    A = S.CSR.from_rows([[(0, 1), (3, -2)], [], [(2, 21006)]], 5, prime=42013)
    path = tmp_path / "a.sms"
    S.save(path, A)
    text = path.read_text()
    assert text == "3 5 M\n1 1 1\n1 4 -2\n3 3 21006\n0 0 0\n"
    B, digest = S.load(path, prime=42013, get_hash=True)
    assert B.shape == (3, 5) and B.rows() == A.rows()
    assert digest == hashlib.sha256(text.encode()).digest()
    # values are reduced into the field of the reader, unsigned residues are accepted (README prints 42012 for -1)
    path.write_text("2 2 M\n1 1 42012\n2 2 -42014\n2 1 84026\n0 0 0\n")
    assert S.load(path, prime=42013).rows() == [[(0, -1)], [(1, -1)]]


def test_sms_reader_rejects_malformed_input(S, tmp_path):
    path = tmp_path / "bad.sms"
    path.write_text("2 2 M\n1 1 5\n")  # no terminator
    with pytest.raises(S.SpasmError, match="terminator"):
        S.load(path)
    path.write_text("2 2 M\n3 1 5\n0 0 0\n")  # row out of range
    with pytest.raises(S.SpasmError, match="outside"):
        S.load(path)


def test_large_random_roundtrip(S, tmp_path):
    A = S.synth_csr(0, 300, 400, density=0.02, prime=65521, seed=5)
    path = tmp_path / "r.sms"
    S.save(path, A)
    B = S.load(path, prime=65521)
    assert B.shape == A.shape and B.rows() == A.rows()


# ---------------------------------------------------------------------------------------------------------------------
# SMS file -> engine, end to end on the GPU.  tests/golden/runtests_matrix.sms is the matrix of the
# reference's own tests, m = sparse([1,1,3,3],[1,2,3,4],[1,2,3,4]) (test/runtests.jl:3), byte for byte as the reference's
# writer puts it out (src/SpaSM.jl:1029-1042: header "rows cols M", one "i j v" line per entry in findnz order -- column
# by column --, 1-based, terminator "0 0 0").  A reader takes the file as the matrix it names (libspasm's
# spasm_triplet_load, :498-512); SpaSM.jl's CSR(::SparseMatrixCSC) hands libspasm the TRANSPOSE (:941-968), which is why
# runtests.jl's `sm` is the transpose of what the file says and its golden kernels (:20-23) pair up as below.
# ---------------------------------------------------------------------------------------------------------------------
GOLDEN_SMS = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "runtests_matrix.sms")


def test_reference_writer_format_fixture(S):
    text = open(GOLDEN_SMS).read()
    assert text == "3 4 M\n1 1 1\n1 2 2\n3 3 3\n3 4 4\n0 0 0\n"
    A = S.load(GOLDEN_SMS, prime=42013)
    assert A.shape == (3, 4) and A.rows() == [[(0, 1), (1, 2)], [], [(2, 3), (3, 4)]]


@pytest.mark.gpu
def test_sms_file_through_the_engine_to_the_golden_kernels(S, tmp_path):
    p = 42013
    A = S.load(GOLDEN_SMS, prime=p)  # the matrix the file names = transpose(sm) of runtests.jl
    fact = S.echelonize(A)
    assert fact.r == 2
    K = S.kernel(fact)
    # test/runtests.jl:23: sparse(kernel(transpose(sm))) == sparse([1,2,3,4],[1,1,2,2],[2,42012,28010,42012])
    assert [[(c, v % p) for c, v in r] for r in K.rows()] == [[(0, 2), (1, 42012)], [(2, 28010), (3, 42012)]]
    # and sm itself (what CSR(m) hands to libspasm): runtests.jl:21, sparse([2],[1],[42012],3,1)
    sm = S.transpose(A)
    Ks = S.kernel(S.echelonize(sm))
    assert [[(c, v % p) for c, v in r] for r in Ks.rows()] == [[(1, 42012)]]
    # written back by the engine's writer, the file reads the same
    out = tmp_path / "again.sms"
    S.save(out, A)
    assert S.load(out, prime=p).rows() == A.rows()
