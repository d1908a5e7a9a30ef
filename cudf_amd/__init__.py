"""cudf_amd — MI355X-native (gfx950, hand-written HIP) hash-groupby / hash-join behind the libcudf API.

Python layer = a pylibcudf-shaped host mirror (Column, Table, groupby.GroupBy, join.inner_join, ...) over the
C ABI of cudf_amd/lib/libcudf_amd.so. The shared library is the product; importing this package without it
fails (no CPU fallback)."""
from . import _lib, aggregation, groupby, join, partitioning, types  # noqa: F401
from .column import Column, Table  # noqa: F401
from .types import DataType, NullEquality, NullPolicy, Sorted, TypeId  # noqa: F401

__version__ = "0.1.0"
