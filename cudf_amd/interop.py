"""Arrow C Data Interface bridge (pylibcudf.interop counterpart): pyarrow Table / RecordBatch <-> cudf_amd.Table
through `cudf::from_arrow` / `cudf::to_arrow_host` (include/cudf/interop.hpp; reference cpp/include/cudf/interop.hpp
:685-689, :618-621). Fixed-width columns only (the types of the hash-groupby / hash-join path)."""
import ctypes as C

from . import _lib
from .column import Table


class _ArrowSchema(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
                ("n_children", C.c_int64), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p),
                ("private_data", C.c_void_p)]


class _ArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                ("n_children", C.c_int64), ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p),
                ("release", C.c_void_p), ("private_data", C.c_void_p)]


def _release(struct):
    if struct.release:
        C.CFUNCTYPE(None, C.c_void_p)(struct.release)(C.addressof(struct))


def from_arrow(obj, stream=None) -> Table:
    """pyarrow.Table or RecordBatch (host memory) -> device Table (copies; bit-packed booleans become BOOL8)."""
    import pyarrow as pa
    if isinstance(obj, pa.Table):
        batches = obj.combine_chunks().to_batches()
        batch = batches[0] if batches else pa.RecordBatch.from_pylist([], schema=obj.schema)
    else:
        batch = obj
    schema, array = _ArrowSchema(), _ArrowArray()
    batch._export_to_c(C.addressof(array), C.addressof(schema))
    try:
        out = C.c_void_p()
        sp = C.c_void_p(int(stream.cuda_stream)) if stream is not None and hasattr(stream, "cuda_stream") else C.c_void_p(stream or 0)
        _lib.check(_lib.load().cudf_amd_from_arrow(C.addressof(schema), C.addressof(array), sp, C.byref(out)))
        return Table._from_handle(out)
    finally:  # the exported structs are ours to release; the library copied what it needed
        _release(array)
        _release(schema)


def to_arrow(table: Table, names=None, stream=None):
    """device Table -> pyarrow.RecordBatch (host copy)."""
    import pyarrow as pa
    n = table.num_columns()
    names = list(names) if names is not None else [f"c{i}" for i in range(n)]
    if len(names) != n:
        raise ValueError("one name per column is required")
    c_names = (C.c_char_p * max(n, 1))(*[s.encode() for s in names])
    schema, array = _ArrowSchema(), _ArrowArray()
    sp = C.c_void_p(int(stream.cuda_stream)) if stream is not None and hasattr(stream, "cuda_stream") else C.c_void_p(stream or 0)
    _lib.check(_lib.load().cudf_amd_to_arrow_host(table._views(), n, c_names, sp, C.addressof(schema), C.addressof(array)))
    return pa.RecordBatch._import_from_c(C.addressof(array), C.addressof(schema))  # takes ownership of both
