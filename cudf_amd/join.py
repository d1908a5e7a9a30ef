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
    cols = Table._from_handle(out, stream).columns()
    return cols[0], cols[1]


def inner_join(left_keys: Table, right_keys: Table, nulls_equal: NullEquality = NullEquality.EQUAL, stream=None, mr=None):
    return _join(left_keys, right_keys, nulls_equal, "inner", stream)


def left_join(left_keys: Table, right_keys: Table, nulls_equal: NullEquality = NullEquality.EQUAL, stream=None, mr=None):
    return _join(left_keys, right_keys, nulls_equal, "left", stream)


def full_join(left_keys: Table, right_keys: Table, nulls_equal: NullEquality = NullEquality.EQUAL, stream=None, mr=None):
    return _join(left_keys, right_keys, nulls_equal, "full", stream)


class JoinMatchContext:
    """cudf::join_match_context (reference join.hpp:81-108): the left table of a probe and its per-row match counts."""

    def __init__(self, left_table: Table, match_counts):
        self._left_table = left_table
        self._match_counts = match_counts  # INT32 Column, one entry per left row


class JoinPartitionContext:
    """cudf::join_partition_context (reference join.hpp:120-125): rows [left_start_idx, left_end_idx) of a match context."""

    def __init__(self, left_table_context: JoinMatchContext, left_start_idx: int, left_end_idx: int):
        self.left_table_context = left_table_context
        self.left_start_idx = int(left_start_idx)
        self.left_end_idx = int(left_end_idx)


def _sync_stream(stream):
    """hipStreamSynchronize of a torch stream / raw hipStream_t / the default stream (None), through the library."""
    _lib.check(_lib.load().cudf_amd_stream_synchronize(_stream_ptr(stream)))


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
                # probes return without synchronising their stream and the object's tables go back to the pool on the BUILD stream:
                # the probe streams used since the last synchronisation are drained first (include/cudf/join/hash_join.hpp, Lifetime)
                for s in getattr(self, "_probe_streams", ()):
                    _sync_stream(s)
                _lib.load().cudf_amd_hash_join_destroy(self._h)
        except Exception:
            pass

    def _note_stream(self, stream):
        if not hasattr(self, "_probe_streams"):
            self._probe_streams = []
        if not any(s is stream for s in self._probe_streams):
            self._probe_streams.append(stream)

    def _probe(self, left_keys, kind, output_size, stream):
        self._note_stream(stream)
        out = C.c_void_p()
        _lib.check(_lib.load().cudf_amd_hash_join_probe(self._h, left_keys._views(), left_keys.num_columns(), _KIND[kind],
                                                        -1 if output_size is None else int(output_size),
                                                        _stream_ptr(stream), C.byref(out)))
        cols = Table._from_handle(out, stream).columns()
        return cols[0], cols[1]

    def _size(self, left_keys, kind, stream):
        n = C.c_uint64()
        _lib.check(_lib.load().cudf_amd_hash_join_size(self._h, left_keys._views(), left_keys.num_columns(), _KIND[kind],
                                                       _stream_ptr(stream), C.byref(n)))
        return n.value

    def _match_context(self, left_keys, kind, stream):
        self._note_stream(stream)
        out = C.c_void_p()
        _lib.check(_lib.load().cudf_amd_hash_join_match_counts(self._h, left_keys._views(), left_keys.num_columns(),
                                                               _KIND[kind], _stream_ptr(stream), C.byref(out)))
        return JoinMatchContext(left_keys, Table._from_handle(out, stream).columns()[0])

    def inner_join_match_context(self, left_keys, stream=None):
        return self._match_context(left_keys, "inner", stream)

    def left_join_match_context(self, left_keys, stream=None):
        return self._match_context(left_keys, "left", stream)

    def full_join_match_context(self, left_keys, stream=None):
        return self._match_context(left_keys, "full", stream)

    def _partitioned(self, context, kind, stream):
        ctx = context.left_table_context
        left = ctx._left_table if ctx is not None else None
        if left is None:
            raise ValueError("join_partition_context has no match context")
        counts = ctx._match_counts
        out = C.c_void_p()
        _lib.check(_lib.load().cudf_amd_hash_join_probe_range(
            self._h, left._views(), left.num_columns(), None if counts is None else C.c_void_p(counts.data_ptr()),
            _KIND[kind], context.left_start_idx, context.left_end_idx, _stream_ptr(stream), C.byref(out)))
        cols = Table._from_handle(out, stream).columns()
        return cols[0], cols[1]

    def partitioned_inner_join(self, context, stream=None):
        return self._partitioned(context, "inner", stream)

    def partitioned_left_join(self, context, stream=None):
        return self._partitioned(context, "left", stream)

    def partitioned_full_join(self, context, stream=None):
        return self._partitioned(context, "full", stream)

    @staticmethod
    def finalize_partitioned_full_join(left_partials, right_partials, left_table_num_rows, right_table_num_rows, stream=None):
        n = len(left_partials)
        if n != len(right_partials):
            raise ValueError("left and right partial results differ in number")
        lp = (C.c_void_p * max(n, 1))(*[c.data_ptr() for c in left_partials])
        rp = (C.c_void_p * max(n, 1))(*[c.data_ptr() for c in right_partials])
        for a, b in zip(left_partials, right_partials):
            if a.size() != b.size():
                raise ValueError("partial index vectors differ in size")
        sz = (C.c_uint64 * max(n, 1))(*[c.size() for c in left_partials])
        out = C.c_void_p()
        _lib.check(_lib.load().cudf_amd_hash_join_finalize_full(lp, rp, sz, n, int(left_table_num_rows),
                                                                int(right_table_num_rows), _stream_ptr(stream), C.byref(out)))
        cols = Table._from_handle(out, stream).columns()
        return cols[0], cols[1]

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
