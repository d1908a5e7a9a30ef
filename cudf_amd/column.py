"""Device columns and tables: host-side mirror of pylibcudf.Column / pylibcudf.Table
(reference python/pylibcudf/pylibcudf/column.pyx, table.pyx) over the C ABI. A Column is either a
non-owning view over caller memory (a torch tensor, a raw device pointer) or a view into an owning result
table handle kept alive by reference."""
import ctypes as C

import numpy as np

from . import _lib
from .types import DataType, TypeId


def _stream_ptr(stream):
    if stream is None:
        return None
    if isinstance(stream, int):
        return C.c_void_p(stream)
    if hasattr(stream, "cuda_stream"):  # torch.cuda.Stream
        return C.c_void_p(stream.cuda_stream)
    raise TypeError("stream must be None, an int hipStream_t, or a torch.cuda.Stream")


class _DeviceAlloc:
    """Owning device allocation from the library's device resource."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = nbytes
        _lib.check(_lib.load().cudf_amd_malloc(C.byref(self.ptr), max(nbytes, 1), None))

    def __del__(self):
        try:
            if self.ptr:
                _lib.load().cudf_amd_free(self.ptr, None)
        except Exception:
            pass


class _TableHandle:
    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            if self.handle:
                _lib.load().cudf_amd_table_free(self.handle)
        except Exception:
            pass


class Column:
    def __init__(self, dtype: DataType, size: int, data_ptr, mask_ptr=None, null_count: int = 0, offset: int = 0,
                 owner=None, children=None, stream=None):
        self._dtype = dtype
        self._size = int(size)
        self._data = int(data_ptr or 0)
        self._mask = int(mask_ptr or 0)
        self._null_count = int(null_count)
        self._offset = int(offset)
        self._owner = owner  # keeps the underlying memory alive
        self._children = list(children or [])  # STRUCT columns (SUM_OVERFLOW: [sum, overflow])
        # The stream that produced this column (results of groupby / join / gather are stream-ordered, like every libcudf
        # result): host copies and synchronisations below run on it. None = the default stream (caller-provided memory).
        self._stream = stream

    def stream(self):
        return self._stream

    # ---- pylibcudf.Column accessors
    def type(self) -> DataType:
        return self._dtype

    def size(self) -> int:
        return self._size

    def null_count(self) -> int:
        return self._null_count

    def offset(self) -> int:
        return self._offset

    def data_ptr(self) -> int:
        return self._data

    def null_mask_ptr(self) -> int:
        return self._mask

    def nullable(self) -> bool:
        return self._mask != 0

    def num_children(self) -> int:
        return len(self._children)

    def child(self, i: int) -> "Column":
        return self._children[i]

    def children(self):
        return list(self._children)

    def _view(self) -> _lib.ColumnView:
        return _lib.ColumnView(int(self._dtype.id()), self._size, self._data or None, self._mask or None,
                               self._null_count, self._offset, self._dtype.scale())

    # ---- constructors
    @staticmethod
    def from_numpy(data, valid=None, dtype: DataType = None, offset: int = 0) -> "Column":
        """Copies a host array (and optional boolean validity array) to the device. With `offset`, the column
        is a slice [offset:] of the uploaded buffers (Arrow offset semantics)."""
        data = np.ascontiguousarray(data)
        if dtype is None:
            dtype = DataType.from_numpy(data.dtype)
        raw = data.astype(np.uint8) if data.dtype == np.bool_ else data
        lib = _lib.load()
        dbuf = _DeviceAlloc(raw.nbytes)
        _lib.check(lib.cudf_amd_memcpy(dbuf.ptr, raw.ctypes.data, raw.nbytes, 0, None))
        owner = [dbuf]
        mask_ptr, nulls = None, 0
        if valid is not None:
            valid = np.asarray(valid, dtype=bool)
            assert len(valid) == len(data)
            nbits = len(valid)
            padded = np.zeros(((nbits + 511) // 512) * 512 + 512, dtype=np.uint8)
            padded[:nbits] = valid
            words = np.packbits(padded.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).copy()
            mbuf = _DeviceAlloc(words.nbytes)
            _lib.check(lib.cudf_amd_memcpy(mbuf.ptr, words.ctypes.data, words.nbytes, 0, None))
            owner.append(mbuf)
            mask_ptr = mbuf.ptr.value
            nulls = int((~valid[offset:]).sum())
        _lib.check(lib.cudf_amd_stream_synchronize(None))
        return Column(dtype, len(data) - offset, dbuf.ptr.value, mask_ptr, nulls, offset, owner)

    @staticmethod
    def from_torch(tensor, mask_tensor=None, null_count: int = 0, dtype: DataType = None) -> "Column":
        """Zero-copy view over a contiguous 1-D torch tensor on the GPU (and an optional int32 bitmask tensor)."""
        import torch

        assert tensor.is_cuda and tensor.is_contiguous() and tensor.dim() == 1
        if dtype is None:
            dtype = DataType.from_numpy(torch.empty(0, dtype=tensor.dtype).numpy().dtype)
        mptr = None
        if mask_tensor is not None:
            assert mask_tensor.is_cuda and mask_tensor.is_contiguous()
            mptr = mask_tensor.data_ptr()
        return Column(dtype, tensor.numel(), tensor.data_ptr(), mptr, null_count, 0, (tensor, mask_tensor))

    # ---- host transfer
    def to_numpy(self):
        """Returns (data, valid_or_None) on the host. A STRUCT column returns the tuple of its children's data."""
        lib = _lib.load()
        sp = _stream_ptr(self._stream)
        if self._children:
            data = tuple(c.to_numpy()[0] for c in self._children)
            valid = None
            if self._mask:
                valid = Column(DataType(TypeId.INT8), self._size, 0, self._mask, self._null_count, self._offset, self._owner,
                               stream=self._stream)._valid_bits()
            return data, valid
        npdt = self._dtype.numpy_dtype()
        raw_dt = np.uint8 if npdt == np.bool_ else npdt
        out = np.empty(self._size, dtype=raw_dt)
        if self._size:
            src = self._data + self._offset * out.itemsize
            _lib.check(lib.cudf_amd_memcpy(out.ctypes.data, C.c_void_p(src), out.nbytes, 1, sp))
        valid = None
        if self._mask:
            nwords = (self._offset + self._size + 31) // 32
            words = np.empty(max(nwords, 1), dtype=np.uint32)
            if nwords:
                _lib.check(lib.cudf_amd_memcpy(words.ctypes.data, C.c_void_p(self._mask), nwords * 4, 1, sp))
        _lib.check(lib.cudf_amd_stream_synchronize(sp))
        if self._mask:
            bits = np.unpackbits(words.view(np.uint8), bitorder="little")
            valid = bits[self._offset:self._offset + self._size].astype(bool)
        if npdt == np.bool_:
            out = out != 0
        return out, valid

    def _valid_bits(self):
        lib = _lib.load()
        sp = _stream_ptr(self._stream)
        nwords = (self._offset + self._size + 31) // 32
        words = np.empty(max(nwords, 1), dtype=np.uint32)
        if nwords:
            _lib.check(lib.cudf_amd_memcpy(words.ctypes.data, C.c_void_p(self._mask), nwords * 4, 1, sp))
        _lib.check(lib.cudf_amd_stream_synchronize(sp))
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")
        return bits[self._offset:self._offset + self._size].astype(bool)

    # ---- zero-copy interop (torch.as_tensor / cupy / numba read this protocol)
    @property
    def __cuda_array_interface__(self):
        npdt = self._dtype.numpy_dtype()
        if npdt == np.bool_:
            npdt = np.dtype(np.uint8)
        cai = {"shape": (self._size,), "typestr": npdt.str, "version": 3,
               "data": (self._data + self._offset * npdt.itemsize, False)}
        sp = _stream_ptr(self._stream)
        if sp is not None and sp.value:
            cai["stream"] = int(sp.value)  # the consumer orders its reads behind the producing stream
        return cai

    def to_torch(self):
        """Zero-copy torch tensor over the column's data (validity is not carried); keeps the column alive."""
        import torch

        if self._size == 0:
            return torch.empty(0, dtype=torch.from_numpy(np.empty(0, self._dtype.numpy_dtype())).dtype, device="cuda")
        sp = _stream_ptr(self._stream)
        # produced on another stream than torch's current one: the tensor is handed out only once it is complete
        # (work queued on torch's current stream is ordered behind the producer by the stream itself)
        if sp is not None and sp.value and int(sp.value) != int(torch.cuda.current_stream().cuda_stream):
            _lib.check(_lib.load().cudf_amd_stream_synchronize(sp))
        t = torch.as_tensor(self, device="cuda")
        t._cudf_amd_owner = self
        return t

    def __repr__(self):
        return f"Column({self._dtype}, size={self._size}, nulls={self._null_count})"


class Table:
    def __init__(self, columns, ragged=False):
        """ragged: the columns of one groupby request's results - a QUANTILE aggregation answers groups x quantiles values in one
        column (reference group_quantiles.cu:84-87), next to columns of one value per group."""
        self._columns = list(columns)
        if self._columns and not ragged:
            n = self._columns[0].size()
            assert all(c.size() == n for c in self._columns), "Column size mismatch."

    def columns(self):
        return list(self._columns)

    def num_columns(self):
        return len(self._columns)

    def num_rows(self):
        return self._columns[0].size() if self._columns else 0

    def _views(self):
        arr = (_lib.ColumnView * max(1, len(self._columns)))(*[c._view() for c in self._columns])
        return arr

    @staticmethod
    def _from_handle(handle, stream=None, ragged=False) -> "Table":
        lib = _lib.load()
        owner = _TableHandle(handle)
        cols = []
        for i in range(lib.cudf_amd_table_num_columns(handle)):
            v = _lib.ColumnView()
            _lib.check(lib.cudf_amd_table_column(handle, i, C.byref(v)))
            children = []
            for j in range(lib.cudf_amd_table_column_num_children(handle, i)):
                cv = _lib.ColumnView()
                _lib.check(lib.cudf_amd_table_column_child(handle, i, j, C.byref(cv)))
                children.append(Column(DataType(TypeId(cv.type_id), cv.scale), cv.size, cv.data, cv.null_mask, cv.null_count,
                                       cv.offset, owner, stream=stream))
            cols.append(Column(DataType(TypeId(v.type_id), v.scale), v.size, v.data, v.null_mask, v.null_count,
                               v.offset, owner, children, stream=stream))
        return Table(cols, ragged)
