"""pylibcudf.join mirror (reference python/pylibcudf/pylibcudf/join.pyx:67-112 and join.pyi): index-pair joins on key
tables, plus a HashJoin object for cudf::hash_join (build once, probe many)."""
import ctypes as C

from . import _lib
from .column import Table, _stream_ptr
from .types import NullEquality

JOIN_NO_MATCH = -2**31
_KIND = {"inner": 0, "left": 1, "full": 2}


def _join(left_keys: Table, right_keys: Table, nulls_equal, kind, stream):
    out = C.c_void_p()
    _lib.check(_lib.load().cudf_amd_join(left_keys._views(), left_keys.num_columns(), right_keys._views(),
                                         right_keys.num_columns(), 1 if nulls_equal == NullEquality.EQUAL else 0,
                                         _KIND[kind], _stream_ptr(stream), C.byref(out)))
    cols = Table._from_handle(out).columns()
    return cols[0], cols[1]


def inner_join(left_keys: Table, right_keys: Table, nulls_equal: NullEquality = NullEquality.EQUAL, stream=None, mr=None):
    return _join(left_keys, right_keys, nulls_equal, "inner", stream)


def left_join(left_keys: Table, right_keys: Table, nulls_equal: NullEquality = NullEquality.EQUAL, stream=None, mr=None):
    return _join(left_keys, right_keys, nulls_equal, "left", stream)


def full_join(left_keys: Table, right_keys: Table, nulls_equal: NullEquality = NullEquality.EQUAL, stream=None, mr=None):
    return _join(left_keys, right_keys, nulls_equal, "full", stream)


class HashJoin:
    """cudf::hash_join(right, [has_nulls], compare_nulls, [load_factor]). `has_nulls`: True / False / None
    (None selects the two-argument constructor, which assumes nulls may be present)."""

    def __init__(self, right_keys: Table, nulls_equal: NullEquality = NullEquality.EQUAL, has_nulls=None,
                 load_factor: float = 0.5, stream=None):
        self._right = right_keys  # keep the build table alive
        self._h = C.c_void_p()
        hn = -1 if has_nulls is None else (1 if has_nulls else 0)
        _lib.check(_lib.load().cudf_amd_hash_join_create(right_keys._views(), right_keys.num_columns(), hn,
                                                         1 if nulls_equal == NullEquality.EQUAL else 0,
                                                         float(load_factor), _stream_ptr(stream), C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                _lib.load().cudf_amd_hash_join_destroy(self._h)
        except Exception:
            pass

    def _probe(self, left_keys, kind, output_size, stream):
        out = C.c_void_p()
        _lib.check(_lib.load().cudf_amd_hash_join_probe(self._h, left_keys._views(), left_keys.num_columns(), _KIND[kind],
                                                        -1 if output_size is None else int(output_size),
                                                        _stream_ptr(stream), C.byref(out)))
        cols = Table._from_handle(out).columns()
        return cols[0], cols[1]

    def _size(self, left_keys, kind, stream):
        n = C.c_uint64()
        _lib.check(_lib.load().cudf_amd_hash_join_size(self._h, left_keys._views(), left_keys.num_columns(), _KIND[kind],
                                                       _stream_ptr(stream), C.byref(n)))
        return n.value

    def inner_join(self, left_keys, output_size=None, stream=None):
        return self._probe(left_keys, "inner", output_size, stream)

    def left_join(self, left_keys, output_size=None, stream=None):
        return self._probe(left_keys, "left", output_size, stream)

    def full_join(self, left_keys, output_size=None, stream=None):
        return self._probe(left_keys, "full", output_size, stream)

    def inner_join_size(self, left_keys, stream=None):
        return self._size(left_keys, "inner", stream)

    def left_join_size(self, left_keys, stream=None):
        return self._size(left_keys, "left", stream)

    def full_join_size(self, left_keys, stream=None):
        return self._size(left_keys, "full", stream)
