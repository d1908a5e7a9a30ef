"""pylibcudf.partitioning / hashing / copying mirrors for the pieces either side of the hot path:
hash_partition (reference python/pylibcudf/pylibcudf/partitioning.pyx), hashing.murmurhash3_x86_32, copying.gather."""
import ctypes as C

from . import _lib
from .column import Column, Table, _stream_ptr
from .types import OutOfBoundsPolicy


def hash_partition(input: Table, columns_to_hash, num_partitions: int, seed: int = 0, stream=None, mr=None):
    """-> (partitioned Table, num_partitions + 1 row offsets: partition i = rows [offsets[i], offsets[i+1]))."""
    cols = (C.c_int32 * max(1, len(columns_to_hash)))(*columns_to_hash)
    offs = (C.c_int32 * (max(0, num_partitions) + 1))()
    out = C.c_void_p()
    _lib.check(_lib.load().cudf_amd_hash_partition(input._views(), input.num_columns(), cols, len(columns_to_hash),
                                                   num_partitions, seed, _stream_ptr(stream), C.byref(out), offs))
    return Table._from_handle(out, stream), list(offs)[:max(num_partitions, 0) + 1]


def murmurhash3_x86_32(input: Table, seed: int = 0, stream=None) -> Column:
    out = C.c_void_p()
    _lib.check(_lib.load().cudf_amd_murmurhash3_x86_32(input._views(), input.num_columns(), seed, _stream_ptr(stream),
                                                       C.byref(out)))
    return Table._from_handle(out, stream).columns()[0]


def gather(source_table: Table, gather_map: Column, bounds_policy: OutOfBoundsPolicy = OutOfBoundsPolicy.DONT_CHECK,
           stream=None, mr=None) -> Table:
    out = C.c_void_p()
    gm = gather_map._view()
    _lib.check(_lib.load().cudf_amd_gather(source_table._views(), source_table.num_columns(), C.byref(gm),
                                           1 if bounds_policy == OutOfBoundsPolicy.NULLIFY else 0, _stream_ptr(stream),
                                           C.byref(out)))
    return Table._from_handle(out, stream)
