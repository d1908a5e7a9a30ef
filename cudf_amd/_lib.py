"""ctypes loader for libcudf_amd.so (the HIP/gfx950 library). There is NO CPU fallback: if the shared library
is missing or fails to load, importing the product fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CUDF_AMD_LIB", os.path.join(_HERE, "lib", "libcudf_amd.so"))  # override: A/B builds


class ColumnView(C.Structure):
    _fields_ = [("type_id", C.c_int32), ("size", C.c_int32), ("data", C.c_void_p), ("null_mask", C.c_void_p),
                ("null_count", C.c_int32), ("offset", C.c_int32), ("scale", C.c_int32)]


class AggregationRequest(C.Structure):
    _fields_ = [("values", ColumnView), ("kinds", C.POINTER(C.c_int32)), ("num_kinds", C.c_int32),
                ("params", C.POINTER(C.c_int32)), ("params2", C.POINTER(C.c_int32)), ("quantiles", C.POINTER(C.c_double)),
                ("quantile_offsets", C.POINTER(C.c_int32))]

    @classmethod
    def of(cls, values_view, aggregations, keep):
        """One request from Aggregation objects; `keep` receives the arrays that must outlive the call."""
        n = len(aggregations)
        kinds = (C.c_int32 * max(1, n))(*[int(a.kind()) for a in aggregations])
        params = (C.c_int32 * max(1, n))(*[a.param(1 if a.kind().name in ("VARIANCE", "STD") else 0) for a in aggregations])
        params2 = (C.c_int32 * max(1, n))(*[a.param2() for a in aggregations])
        offs, flat = [0], []
        for a in aggregations:
            flat += a.quantiles()
            offs.append(len(flat))
        quantiles = (C.c_double * max(1, len(flat)))(*flat)
        offsets = (C.c_int32 * (n + 1))(*offs)
        keep += [kinds, params, params2, quantiles, offsets]
        return cls(values_view, kinds, n, params, params2, quantiles, offsets)


# Every symbol declared in include/cudf_amd_c.h: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "cudf_amd_last_error": (C.c_char_p, []),
    "cudf_amd_version": (C.c_char_p, []),
    "cudf_amd_abi_version": (C.c_int32, []),
    "cudf_amd_malloc": (C.c_int, [C.POINTER(_P), C.c_size_t, _P]),
    "cudf_amd_free": (C.c_int, [_P, _P]),
    "cudf_amd_memcpy": (C.c_int, [_P, _P, C.c_size_t, C.c_int32, _P]),
    "cudf_amd_memset": (C.c_int, [_P, C.c_int32, C.c_size_t, _P]),
    "cudf_amd_stream_synchronize": (C.c_int, [_P]),
    "cudf_amd_memory_stats": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cudf_amd_profile_enable": (C.c_int, [C.c_int32]),
    "cudf_amd_profile_reset": (C.c_int, []),
    "cudf_amd_profile_report": (C.c_int, [C.c_char_p, C.c_size_t]),
    "cudf_amd_table_num_columns": (C.c_int32, [_P]),
    "cudf_amd_table_num_rows": (C.c_int32, [_P]),
    "cudf_amd_table_column": (C.c_int, [_P, C.c_int32, C.POINTER(ColumnView)]),
    "cudf_amd_table_column_num_children": (C.c_int32, [_P, C.c_int32]),
    "cudf_amd_table_column_child": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(ColumnView)]),
    "cudf_amd_table_free": (None, [_P]),
    "cudf_amd_groupby_aggregate": (C.c_int, [C.POINTER(ColumnView), C.c_int32, C.c_int32, C.c_int32,
                                             C.POINTER(AggregationRequest), C.c_int32, _P, C.POINTER(_P),
                                             C.POINTER(_P), C.POINTER(C.c_int32)]),
    "cudf_amd_join": (C.c_int, [C.POINTER(ColumnView), C.c_int32, C.POINTER(ColumnView), C.c_int32, C.c_int32, C.c_int32,
                                _P, C.POINTER(_P)]),
    "cudf_amd_hash_join_create": (C.c_int, [C.POINTER(ColumnView), C.c_int32, C.c_int32, C.c_int32, C.c_double, _P,
                                            C.POINTER(_P)]),
    "cudf_amd_hash_join_destroy": (None, [_P]),
    "cudf_amd_hash_join_probe": (C.c_int, [_P, C.POINTER(ColumnView), C.c_int32, C.c_int32, C.c_int64, _P,
                                           C.POINTER(_P)]),
    "cudf_amd_hash_join_size": (C.c_int, [_P, C.POINTER(ColumnView), C.c_int32, C.c_int32, _P, C.POINTER(C.c_uint64)]),
    "cudf_amd_from_arrow": (C.c_int, [_P, _P, _P, C.POINTER(_P)]),
    "cudf_amd_to_arrow_host": (C.c_int, [C.POINTER(ColumnView), C.c_int32, C.POINTER(C.c_char_p), _P, _P, _P]),
    "cudf_amd_hash_join_match_counts": (C.c_int, [_P, C.POINTER(ColumnView), C.c_int32, C.c_int32, _P, C.POINTER(_P)]),
    "cudf_amd_hash_join_probe_range": (C.c_int, [_P, C.POINTER(ColumnView), C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32,
                                                 _P, C.POINTER(_P)]),
    "cudf_amd_hash_join_finalize_full": (C.c_int, [C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_uint64), C.c_int32,
                                                   C.c_int32, C.c_int32, _P, C.POINTER(_P)]),
    "cudf_amd_hash_partition": (C.c_int, [C.POINTER(ColumnView), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                          C.c_uint32, _P, C.POINTER(_P), C.POINTER(C.c_int32)]),
    "cudf_amd_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "cudf_amd_comm_create": (C.c_int, [C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.POINTER(_P)]),
    "cudf_amd_comm_destroy": (None, [_P]),
    "cudf_amd_comm_create_loopback": (C.c_int, [C.c_int32, C.POINTER(_P)]),
    "cudf_amd_comm_set_max_message_bytes": (C.c_int, [_P, C.c_int64]),
    "cudf_amd_plan_exchange": (C.c_int, [C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64)]),
    "cudf_amd_shuffle_join": (C.c_int, [_P, C.POINTER(ColumnView), C.c_int32, C.POINTER(ColumnView), C.c_int32, C.c_int32, _P,
                                        C.POINTER(_P)]),
    "cudf_amd_range_partition": (C.c_int, [C.POINTER(ColumnView), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, _P,
                                           C.POINTER(_P), C.POINTER(C.c_int32)]),
    "cudf_amd_shuffle": (C.c_int, [_P, C.POINTER(ColumnView), C.c_int32, C.POINTER(C.c_int32), C.c_int32, _P, C.POINTER(_P)]),
    "cudf_amd_shuffle_groupby": (C.c_int, [_P, C.POINTER(ColumnView), C.c_int32, C.c_int32, C.POINTER(AggregationRequest),
                                           C.c_int32, _P, C.POINTER(_P), C.POINTER(_P)]),
    "cudf_amd_combine_groupby": (C.c_int, [_P, C.POINTER(ColumnView), C.c_int32, C.c_int32, C.POINTER(AggregationRequest),
                                           C.c_int32, _P, C.POINTER(_P), C.POINTER(_P)]),
    "cudf_amd_murmurhash3_x86_32": (C.c_int, [C.POINTER(ColumnView), C.c_int32, C.c_uint32, _P, C.POINTER(_P)]),
    "cudf_amd_gather": (C.c_int, [C.POINTER(ColumnView), C.c_int32, C.POINTER(ColumnView), C.c_int32, _P, C.POINTER(_P)]),
}

_lib = None


ABI_VERSION = 4  # include/cudf_amd_c.h CUDF_AMD_ABI_VERSION


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950). cudf_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if lib.cudf_amd_abi_version() != ABI_VERSION:  # struct layouts and signatures below were written for this version
            raise ImportError(f"{LIB_PATH} has ABI version {lib.cudf_amd_abi_version()}, cudf_amd/_lib.py expects {ABI_VERSION}: rebuild the library")
        _lib = lib
    return _lib


class CudfAmdError(RuntimeError):
    """cudf::logic_error / device errors (pylibcudf maps these to RuntimeError)."""


# status -> Python exception, as pylibcudf's exception_handler maps the C++ types
# (reference python/pylibcudf/pylibcudf/exception_handler.pxd:34-64)
def check(status):
    if status == 0:
        return
    msg = load().cudf_amd_last_error().decode(errors="replace")
    if status == 2:
        raise ValueError(msg)
    if status == 3:
        raise TypeError(msg)
    if status == 5:
        raise MemoryError(msg)
    if status == 7:
        raise IndexError(msg)
    raise CudfAmdError(msg)


def profile_enable(on: bool):
    check(load().cudf_amd_profile_enable(1 if on else 0))


def profile_reset():
    check(load().cudf_amd_profile_reset())


def profile_report():
    """{kernel name: (launches, total_ms)} since the last reset (HIP events on the launch stream)."""
    buf = C.create_string_buffer(1 << 16)
    check(load().cudf_amd_profile_report(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, ms = line.split()
        out[name] = (int(n), float(ms))
    return out
