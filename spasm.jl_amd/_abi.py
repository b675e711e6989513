"""ctypes mirrors of the C ABI in include/spasm_amd.h and the loader of libspasm_amd.so.

The struct layouts are the ones SpaSM.jl declares for its @ccall bindings
(reference src/SpaSM.jl:51-56 Field, :126-134 _CSR, :262-270 _LU, :325-343 EchelonizeOpts).
The library is REQUIRED: there is no Python or CPU fallback behind these bindings.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPASM_AMD_LIB") or os.path.join(_HERE, "libspasm_amd.so")  # env override: diagnostic builds only


class Field(C.Structure):  # reference src/SpaSM.jl:51-56
    _fields_ = [("p", C.c_int64), ("halfp", C.c_int64), ("mhalfp", C.c_int64), ("dinvp", C.c_double)]


class CsrStruct(C.Structure):  # reference src/SpaSM.jl:126-134
    _fields_ = [
        ("nzmax", C.c_int64),
        ("n", C.c_int32),
        ("m", C.c_int32),
        ("p", C.POINTER(C.c_int64)),
        ("j", C.POINTER(C.c_int32)),
        ("x", C.POINTER(C.c_int32)),
        ("field", Field),
    ]


class TripletStruct(C.Structure):  # reference src/SpaSM.jl:234-243
    _fields_ = [
        ("nzmax", C.c_int64),
        ("nz", C.c_int64),
        ("n", C.c_int32),
        ("m", C.c_int32),
        ("i", C.POINTER(C.c_int32)),
        ("j", C.POINTER(C.c_int32)),
        ("x", C.POINTER(C.c_int32)),
        ("field", Field),
    ]


class LuStruct(C.Structure):  # reference src/SpaSM.jl:262-270
    _fields_ = [
        ("r", C.c_int32),
        ("complete", C.c_bool),
        ("L", C.POINTER(CsrStruct)),
        ("U", C.POINTER(CsrStruct)),
        ("qinv", C.POINTER(C.c_int32)),
        ("p", C.POINTER(C.c_int32)),
        ("Ltmp", C.c_void_p),
    ]


class EchelonizeOptsStruct(C.Structure):  # reference src/SpaSM.jl:325-343
    _fields_ = [
        ("enable_greedy_pivot_search", C.c_bool),
        ("enable_tall_and_skinny", C.c_bool),
        ("enable_dense", C.c_bool),
        ("enable_GPLU", C.c_bool),
        ("L", C.c_bool),
        ("complete", C.c_bool),
        ("min_pivot_proportion", C.c_double),
        ("max_round", C.c_int32),
        ("sparsity_threshold", C.c_double),
        ("dense_block_size", C.c_int64),  # Julia declares Int at :339
        ("low_rank_ratio", C.c_double),
        ("tall_and_skinny_ratio", C.c_double),
        ("low_rank_start_weight", C.c_double),
    ]


class RankCertificateStruct(C.Structure):  # struct spasm_rank_certificate <-> RankCertificate{F} (reference src/SpaSM.jl:345-353)
    _fields_ = [
        ("r", C.c_int32),
        ("prime", C.c_int64),
        ("hash", C.c_uint8 * 32),
        ("i", C.POINTER(C.c_int32)),
        ("j", C.POINTER(C.c_int32)),
        ("x", C.POINTER(C.c_int32)),
        ("y", C.POINTER(C.c_int32)),
    ]


class RoundStats(C.Structure):  # struct spasm_amd_round_stats (engine extension)
    _fields_ = [
        ("round", C.c_int32),
        ("rows_in", C.c_int32),
        ("nnz_in", C.c_int64),
        ("npiv", C.c_int32),
        ("rows_out", C.c_int32),
        ("nnz_out", C.c_int64),
        ("nnz_reduced", C.c_int64),
        ("applications", C.c_int64),
        ("read_bytes", C.c_int64),
        ("ms_pivots", C.c_double),
        ("ms_solve", C.c_double),
        ("ms_scatter", C.c_double),
        ("ms_total", C.c_double),
        ("ms_class", C.c_double * 16),
        ("rows_class", C.c_int32 * 16),
        ("ent_class", C.c_int64 * 16),
        ("seg_class", C.c_int64 * 16),
        ("stream_fix", C.c_int64),
        ("stream_redo", C.c_int64),
        ("ms_uinv", C.c_double),
        ("ms_w", C.c_double),
        ("npiv_open", C.c_int64),
        ("ms_wbuild", C.c_double),
        ("w_levels", C.c_int64),
        ("w_entries", C.c_int64),
        ("w_long_rows", C.c_int64),
        ("npiv_greedy", C.c_int64),
        ("ms_fused", C.c_double),
        ("ms_fused_fix", C.c_double),
        ("rows_fused", C.c_int64),
        ("ent_fused", C.c_int64),
        ("seg_fused", C.c_int64),
        ("rows_rejected", C.c_int64),
        ("s_entries_used", C.c_int64),
        ("ms_levels", C.c_double),
        ("ms_w_sizing", C.c_double),
    ]

    def as_dict(self):
        d = {}
        for k, _ in self._fields_:
            v = getattr(self, k)
            d[k] = list(v) if hasattr(v, "__len__") else v
        return d


# sizes / offsets the Julia mirrors imply; checked in tests/test_abi.py
EXPECTED_LAYOUT = {
    "Field": (32, {"p": 0, "halfp": 8, "mhalfp": 16, "dinvp": 24}),
    "CsrStruct": (72, {"nzmax": 0, "n": 8, "m": 12, "p": 16, "j": 24, "x": 32, "field": 40}),
    "TripletStruct": (80, {"nzmax": 0, "nz": 8, "n": 16, "m": 20, "i": 24, "j": 32, "x": 40, "field": 48}),
    "LuStruct": (48, {"r": 0, "complete": 4, "L": 8, "U": 16, "qinv": 24, "p": 32, "Ltmp": 40}),
    "EchelonizeOptsStruct": (
        64,
        {
            "enable_greedy_pivot_search": 0,
            "enable_tall_and_skinny": 1,
            "enable_dense": 2,
            "enable_GPLU": 3,
            "L": 4,
            "complete": 5,
            "min_pivot_proportion": 8,
            "max_round": 16,
            "sparsity_threshold": 24,
            "dense_block_size": 32,
            "low_rank_ratio": 40,
            "tall_and_skinny_ratio": 48,
            "low_rank_start_weight": 56,
        },
    ),
}

# every function symbol include/spasm_amd.h declares: name -> (restype, argtypes)
_P = C.POINTER
SIGNATURES = {
    "spasm_wtime": (C.c_double, []),
    "spasm_nnz": (C.c_int64, [_P(CsrStruct)]),
    "spasm_csr_alloc": (_P(CsrStruct), [C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_bool]),
    "spasm_csr_realloc": (None, [_P(CsrStruct), C.c_int64]),
    "spasm_csr_resize": (None, [_P(CsrStruct), C.c_int32, C.c_int32]),
    "spasm_csr_free": (None, [_P(CsrStruct)]),
    "spasm_lu_free": (None, [_P(LuStruct)]),
    "spasm_get_num_threads": (C.c_int32, []),
    "spasm_get_thread_num": (C.c_int32, []),
    "spasm_field_init": (None, [C.c_int64, _P(Field)]),
    "spasm_transpose": (_P(CsrStruct), [_P(CsrStruct)]),
    "spasm_factorization_verify": (C.c_bool, [_P(CsrStruct), _P(LuStruct), C.c_uint64]),
    "spasm_gesv": (_P(CsrStruct), [_P(LuStruct), _P(CsrStruct), C.c_void_p]),
    "spasm_solve": (C.c_bool, [_P(LuStruct), C.c_void_p, C.c_void_p]),
    "spasm_rref": (_P(CsrStruct), [_P(LuStruct), _P(C.c_int32)]),
    "spasm_triplet_alloc": (_P(TripletStruct), [C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_bool]),
    "spasm_triplet_realloc": (None, [_P(TripletStruct), C.c_int64]),
    "spasm_triplet_free": (None, [_P(TripletStruct)]),
    "spasm_add_entry": (None, [_P(TripletStruct), C.c_int32, C.c_int32, C.c_int64]),
    "spasm_triplet_transpose": (None, [_P(TripletStruct)]),
    "spasm_compress": (_P(CsrStruct), [_P(TripletStruct)]),
    "spasm_triplet_load": (_P(TripletStruct), [C.c_void_p, C.c_int64, C.c_void_p]),
    "spasm_triplet_save": (None, [_P(TripletStruct), C.c_void_p]),
    "spasm_csr_save": (None, [_P(CsrStruct), C.c_void_p]),
    "spasm_echelonize_init_opts": (None, [_P(EchelonizeOptsStruct)]),
    "spasm_echelonize": (_P(LuStruct), [_P(CsrStruct), _P(EchelonizeOptsStruct)]),
    "spasm_kernel": (_P(CsrStruct), [_P(LuStruct)]),
    "spasm_amd_last_error": (C.c_char_p, []),
    "spasm_amd_device_count": (C.c_int32, []),
    "spasm_amd_set_device": (C.c_int32, [C.c_int32]),
    "spasm_amd_synth_csr": (_P(CsrStruct), [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int64, C.c_uint64]),
    "spasm_amd_schur_plan_create": (C.c_void_p, [_P(CsrStruct), C.c_int32, C.c_int32]),
    "spasm_amd_schur_plan_create_strided": (C.c_void_p, [_P(CsrStruct), C.c_int32, C.c_int32, C.c_int32]),
    "spasm_amd_schur_plan_run": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "spasm_amd_schur_plan_class_timing": (None, [C.c_void_p, C.c_int32]),
    "spasm_amd_schur_plan_fetch_U": (_P(CsrStruct), [C.c_void_p, _P(C.c_int32), _P(C.c_int32)]),
    "spasm_amd_schur_plan_stats": (C.c_int32, [C.c_void_p, _P(RoundStats)]),
    "spasm_amd_schur_plan_fetch": (_P(CsrStruct), [C.c_void_p, _P(C.c_int32)]),
    "spasm_amd_schur_plan_free": (None, [C.c_void_p]),
    "spasm_amd_last_rounds": (C.c_int32, [_P(RoundStats), C.c_int32]),
    "spasm_amd_rank": (C.c_int64, [_P(CsrStruct), _P(EchelonizeOptsStruct)]),
    "spasm_amd_zp_probe": (C.c_int32, [C.c_int64, C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "spasm_amd_shard_create": (C.c_void_p, [_P(CsrStruct), C.c_int32, C.c_int32]),
    "spasm_amd_shard_create_strided": (C.c_void_p, [_P(CsrStruct), C.c_int32, C.c_int32, C.c_int32]),
    "spasm_amd_shard_elect": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "spasm_amd_shard_set_keys": (C.c_int32, [C.c_void_p, C.c_void_p, _P(C.c_int32), _P(C.c_int64)]),
    "spasm_amd_shard_assign": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "spasm_amd_shard_open_step": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "spasm_amd_shard_finish_keys": (C.c_int32, [C.c_void_p, _P(C.c_int32), _P(C.c_int64)]),
    "spasm_amd_shard_export": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "spasm_amd_shard_import": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "spasm_amd_triangular_solve": (_P(CsrStruct), [_P(CsrStruct), _P(C.c_int32), _P(CsrStruct), _P(C.c_ubyte)]),
    "spasm_sparse_triangular_solve": (C.c_int32, [_P(CsrStruct), _P(CsrStruct), C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "spasm_scatter": (None, [_P(CsrStruct), C.c_int32, C.c_int32, _P(C.c_int32)]),
    "spasm_certificate_rank_create": (_P(RankCertificateStruct), [_P(CsrStruct), _P(C.c_uint8), _P(LuStruct)]),
    "spasm_certificate_rank_verify": (C.c_bool, [_P(CsrStruct), _P(C.c_uint8), _P(RankCertificateStruct)]),
    "spasm_rank_certificate_save": (None, [_P(RankCertificateStruct), C.c_void_p]),
    "spasm_rank_certificate_load": (C.c_bool, [C.c_void_p, _P(RankCertificateStruct)]),
    "spasm_rank_certificate_free": (None, [_P(RankCertificateStruct)]),
    "spasm_amd_echelonize_multi": (_P(LuStruct), [_P(CsrStruct), _P(EchelonizeOptsStruct), C.c_int32]),
    "spasm_amd_multi_last_finish": (C.c_int32, []),
    "spasm_amd_certificate_challenge": (None, [_P(C.c_uint8), C.c_int64, C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "spasm_amd_kernel_strided": (_P(CsrStruct), [_P(LuStruct), C.c_int32, C.c_int32]),
    "spasm_amd_schur_plan_advance": (C.c_void_p, [C.c_void_p, _P(C.c_int32), _P(C.c_int64)]),
    "spasm_amd_shard_fetch": (_P(CsrStruct), [C.c_void_p]),
    "spasm_amd_shard_free": (None, [C.c_void_p]),
    "spasm_amd_shard_import_U": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "spasm_amd_schur_plan_prepare": (C.c_int32, [C.c_void_p]),
    "spasm_amd_dshard_open": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_int32]),
    "spasm_amd_dshard_open_rows": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_int32]),
    "spasm_amd_dshard_flags": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "spasm_amd_dshard_density": (C.c_double, [C.c_void_p, C.c_void_p, C.c_int32, _P(C.c_int32)]),
    "spasm_amd_dshard_build": (C.c_int32, [C.c_void_p]),
    "spasm_amd_dshard_info": (C.c_int32, [C.c_void_p, _P(C.c_int32), _P(C.c_int32), _P(C.c_int64), _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "spasm_amd_dshard_block_begin": (C.c_int32, [C.c_void_p]),
    "spasm_amd_dshard_candidates": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "spasm_amd_dshard_elect": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "spasm_amd_dshard_pack": (C.c_int64, [C.c_void_p, C.c_int32, C.c_void_p]),
    "spasm_amd_dshard_unpack": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "spasm_amd_dshard_apply": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "spasm_amd_dshard_block_end": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "spasm_amd_dshard_finish": (C.c_int32, [C.c_void_p]),
    "spasm_amd_dshard_fetch_U": (_P(CsrStruct), [C.c_void_p, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "spasm_amd_dshard_close": (None, [C.c_void_p]),
}
DATA_SYMBOLS = ["logcallback"]

_lib = None


def lib():
    """Load libspasm_amd.so (once).  Raises if it has not been built: no fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The engine has no CPU fallback."
            )
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def last_error():
    msg = lib().spasm_amd_last_error()
    return msg.decode("utf-8", "replace") if msg else ""
