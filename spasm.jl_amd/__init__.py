"""spasm.jl_amd -- MI355X-native sparse GF(p) echelonization behind SpaSM.jl's CSR / echelonize / kernel surface.

The directory name carries a dot, so import it as `spasm_jl_amd` (the loader module of that name
at the repository root registers this package).
"""
from . import _abi, blocks
from .blocks import Block
from .api import (
    CSR,
    LU,
    EchelonizeOpts,
    Field,
    SpasmError,
    Triplet,
    ZZp,
    RankCertificate,
    balanced,
    certificate_rank_create,
    certificate_rank_verify,
    rank_certificate_load,
    rank_certificate_save,
    echelonize,
    echelonize_multi,
    factorization_verify,
    gesv,
    kernel,
    last_rounds,
    load,
    nnz,
    prime0,
    rank,
    rref,
    save,
    scatter,
    solve,
    sparse,
    sparse_triangular_solve,
    sparse_triangular_solve_row,
    synth_csr,
    transpose,
)

__all__ = [
    "Block", "blocks", "CSR", "LU", "Triplet", "load", "save", "EchelonizeOpts", "Field", "SpasmError", "ZZp", "balanced", "RankCertificate", "certificate_rank_create", "certificate_rank_verify", "rank_certificate_save", "rank_certificate_load", "echelonize", "echelonize_multi", "factorization_verify", "gesv", "solve", "kernel",
    "last_rounds", "nnz", "prime0", "rank", "rref", "sparse", "sparse_triangular_solve", "sparse_triangular_solve_row", "scatter", "synth_csr", "transpose",
]
