"""TEST INFRASTRUCTURE ONLY — ctypes wrapper over oracle/liboracle.so (the CPU restatement in oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(cudf_amd/) never does. Columns are (numpy array, optional bool validity array) pairs; results come back as
numpy arrays plus validity arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# cudf::type_id values (reference cpp/include/cudf/types.hpp:185-217)
TYPE_ID = {
    "int8": 1, "int16": 2, "int32": 3, "int64": 4, "uint8": 5, "uint16": 6, "uint32": 7, "uint64": 8,
    "float32": 9, "float64": 10, "bool": 11,
    "timestamp_days": 12, "timestamp_s": 13, "timestamp_ms": 14, "timestamp_us": 15, "timestamp_ns": 16,
    "duration_days": 17, "duration_s": 18, "duration_ms": 19, "duration_us": 20, "duration_ns": 21,
    "decimal32": 25, "decimal64": 26,
}
NP_OF_TYPE_ID = {1: np.int8, 2: np.int16, 3: np.int32, 4: np.int64, 5: np.uint8, 6: np.uint16, 7: np.uint32,
                 8: np.uint64, 9: np.float32, 10: np.float64, 11: np.bool_, 12: np.int32, 13: np.int64,
                 14: np.int64, 15: np.int64, 16: np.int64, 17: np.int32, 18: np.int64, 19: np.int64,
                 20: np.int64, 21: np.int64, 25: np.int32, 26: np.int64}
# cudf::aggregation::Kind values (reference cpp/include/cudf/aggregation.hpp:78-121)
KIND = {"sum": 0, "sum_overflow": 1, "product": 2, "min": 3, "max": 4, "count_valid": 5, "count_all": 6, "sum_of_squares": 9,
        "mean": 10, "m2": 11, "variance": 12, "std": 13, "argmax": 16, "argmin": 17, "nth_element": 19}
JOIN_NO_MATCH = -2**31


class OracleError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class _Col(C.Structure):
    _fields_ = [("type_id", C.c_int32), ("size", C.c_int32), ("data", C.c_void_p), ("mask", C.c_void_p),
                ("null_count", C.c_int32), ("offset", C.c_int32)]


class _Req(C.Structure):
    _fields_ = [("values", _Col), ("kinds", C.POINTER(C.c_int32)), ("nkinds", C.c_int32)]


class _OutCol(C.Structure):
    _fields_ = [("type_id", C.c_int32), ("size", C.c_int32), ("data", C.c_void_p), ("mask", C.c_void_p),
                ("null_count", C.c_int32), ("aux", C.c_void_p)]


class _GbRes(C.Structure):
    _fields_ = [("nkeys", C.c_int32), ("keys", C.POINTER(_OutCol)), ("nresults", C.c_int32),
                ("results", C.POINTER(_OutCol))]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_last_error.restype = C.c_char_p
        _LIB.orc_murmur3_32.restype = C.c_uint32
        _LIB.orc_murmur3_32.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
    return _LIB


def _check(rc):
    if rc != 0:
        raise OracleError(rc, lib().orc_last_error().decode())


def pack_mask(valid, offset=0):
    """bool validity array -> LSB-first uint32 bitmask words, bit (offset+i) for element i."""
    n = len(valid) + offset
    bits = np.zeros(((n + 31) // 32) * 32, dtype=np.uint8)
    bits[offset:offset + len(valid)] = np.asarray(valid, dtype=np.uint8)
    return np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint32).copy()


def unpack_mask(words, n):
    if n == 0:
        return np.zeros(0, dtype=bool)
    b = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")
    return b[:n].astype(bool)


class HostColumn:
    """A host Arrow-layout column: data array, optional validity (bool array), logical type name."""

    def __init__(self, data, valid=None, type_name=None, offset=0):
        data = np.ascontiguousarray(data)
        if type_name is None:
            type_name = "bool" if data.dtype == np.bool_ else data.dtype.name
        self.type_id = TYPE_ID[type_name]
        self.data = data if data.dtype != np.bool_ else data.astype(np.uint8)
        self.offset = offset
        self.size = len(data) - offset
        self.valid = None if valid is None else np.asarray(valid, dtype=bool)
        self.mask = None if valid is None else pack_mask(self.valid, 0)
        self.null_count = 0 if valid is None else int((~self.valid[offset:offset + self.size]).sum())

    def c(self):
        return _Col(self.type_id, self.size, self.data.ctypes.data,
                    None if self.mask is None else self.mask.ctypes.data, self.null_count, self.offset)


def _as_cols(cols):
    hc = [c if isinstance(c, HostColumn) else HostColumn(*c) if isinstance(c, tuple) else HostColumn(c) for c in cols]
    arr = (_Col * max(1, len(hc)))(*[h.c() for h in hc])
    return hc, arr


def _take_out(oc):
    npt = NP_OF_TYPE_ID[oc.type_id]
    n = oc.size
    itemsize = np.dtype(npt).itemsize
    data = np.frombuffer(C.string_at(oc.data, max(n, 0) * itemsize), dtype=npt).copy() if n else np.zeros(0, npt)
    valid = None
    if oc.mask:
        words = np.frombuffer(C.string_at(oc.mask, ((n + 31) // 32) * 4), dtype=np.uint32)
        valid = unpack_mask(words, n)
    if oc.aux:  # SUM_OVERFLOW: struct {sum, overflow} -> tuple of the children, as Column.to_numpy() of a STRUCT column
        flags = np.frombuffer(C.string_at(oc.aux, max(n, 0)), dtype=np.uint8).astype(bool) if n else np.zeros(0, bool)
        return (data, flags), valid, oc.type_id
    return data, valid, oc.type_id


def groupby(keys, requests, include_null_keys=False):
    """keys: list of columns; requests: list of (values_column, [kind names]).
    Returns (key_cols, result_cols) each a list of (data, valid_or_None, type_id); result_cols is a list per
    request of lists per aggregation."""
    L = lib()
    kh, karr = _as_cols(keys)
    holders, reqs = [], []
    for vals, kinds in requests:
        vh = vals if isinstance(vals, HostColumn) else HostColumn(*vals) if isinstance(vals, tuple) else HostColumn(vals)
        ks = (C.c_int32 * max(1, len(kinds)))(*[KIND[k] if isinstance(k, str) else int(k) for k in kinds])
        holders.append((vh, ks))
        reqs.append(_Req(vh.c(), ks, len(kinds)))
    rarr = (_Req * max(1, len(reqs)))(*reqs)
    out = C.POINTER(_GbRes)()
    _check(L.orc_groupby(karr, len(kh), int(include_null_keys), rarr, len(reqs), C.byref(out)))
    try:
        res = out.contents
        kc = [_take_out(res.keys[i]) for i in range(res.nkeys)]
        flat = [_take_out(res.results[i]) for i in range(res.nresults)]
    finally:
        L.orc_groupby_free(out)
    rc, p = [], 0
    for _, kinds in requests:
        rc.append(flat[p:p + len(kinds)])
        p += len(kinds)
    return kc, rc


def join(left, right, nulls_equal=True, kind="inner"):
    L = lib()
    lh, larr = _as_cols(left)
    rh, rarr = _as_cols(right)
    pl, pr = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
    n = C.c_int64()
    _check(L.orc_join(larr, len(lh), rarr, len(rh), int(nulls_equal), {"inner": 0, "left": 1, "full": 2}[kind],
                      C.byref(pl), C.byref(pr), C.byref(n)))
    try:
        li = np.ctypeslib.as_array(pl, shape=(max(n.value, 1),))[:n.value].copy()
        ri = np.ctypeslib.as_array(pr, shape=(max(n.value, 1),))[:n.value].copy()
    finally:
        L.orc_free(pl)
        L.orc_free(pr)
    return li, ri


def join_size(left, right, nulls_equal=True, kind="inner"):
    L = lib()
    lh, larr = _as_cols(left)
    rh, rarr = _as_cols(right)
    n = C.c_uint64()
    _check(L.orc_join_size(larr, len(lh), rarr, len(rh), int(nulls_equal), {"inner": 0, "left": 1, "full": 2}[kind],
                           C.byref(n)))
    return n.value


def row_hash(cols, seed=0):
    L = lib()
    h, arr = _as_cols(cols)
    out = np.zeros(h[0].size if h else 0, dtype=np.uint32)
    _check(L.orc_row_hash(arr, len(h), C.c_uint32(seed), out.ctypes.data_as(C.c_void_p)))
    return out


def murmur3_32(data: bytes, seed=0):
    buf = C.create_string_buffer(data, len(data))
    return lib().orc_murmur3_32(buf, len(data), seed)


def hash_partition(cols, num_partitions, seed=0):
    L = lib()
    h, arr = _as_cols(cols)
    n = h[0].size if h else 0
    part = np.zeros(n, np.int32)
    offs = np.zeros(num_partitions, np.int32)
    order = np.zeros(n, np.int32)
    _check(L.orc_hash_partition(arr, len(h), num_partitions, C.c_uint32(seed), part.ctypes.data_as(C.c_void_p),
                                offs.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p)))
    return part, offs, order
