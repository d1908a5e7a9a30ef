"""pylibcudf.groupby mirror (reference python/pylibcudf/pylibcudf/groupby.pyx:40-215): GroupByRequest and
GroupBy.aggregate -> (keys Table, [results Table per request])."""
import ctypes as C
from enum import IntEnum

from . import _lib
from .column import Table, _stream_ptr
from .types import NullPolicy, Sorted


class HashPath(IntEnum):
    NONE = 0
    LDS_SINGLE_PASS = 1
    PARTITIONED_LDS = 2
    GLOBAL_TABLE = 3
    DENSE_DIRECT = 4
    SORT = 5


class GroupByRequest:
    def __init__(self, values, aggregations):
        self._values = values
        self._aggregations = list(aggregations)


class GroupBy:
    def __init__(self, keys: Table, null_handling: NullPolicy = NullPolicy.EXCLUDE,
                 keys_are_sorted: Sorted = Sorted.NO, column_order=None, null_precedence=None):
        self._keys = keys  # pinned, as pylibcudf does (groupby.pyx:143-145)
        self._null_handling = NullPolicy(null_handling)
        self._keys_are_sorted = Sorted(keys_are_sorted)
        self.last_path = HashPath.NONE

    def aggregate(self, requests, stream=None, mr=None):
        lib = _lib.load()
        reqs, keep = [], []
        for r in requests:
            reqs.append(_lib.AggregationRequest.of(r._values._view(), r._aggregations, keep))
        rarr = (_lib.AggregationRequest * max(1, len(reqs)))(*reqs)
        out_keys, out_res, path = C.c_void_p(), C.c_void_p(), C.c_int32(0)
        _lib.check(lib.cudf_amd_groupby_aggregate(self._keys._views(), self._keys.num_columns(),
                                                  int(self._null_handling), int(self._keys_are_sorted), rarr,
                                                  len(reqs), _stream_ptr(stream), C.byref(out_keys),
                                                  C.byref(out_res), C.byref(path)))
        self.last_path = HashPath(path.value)
        keys = Table._from_handle(out_keys, stream)
        flat = Table._from_handle(out_res, stream, ragged=True).columns()
        results, p = [], 0
        for r in requests:
            results.append(Table(flat[p:p + len(r._aggregations)], ragged=True))
            p += len(r._aggregations)
        return keys, results
