"""Type vocabulary mirroring pylibcudf.types (reference python/pylibcudf/pylibcudf/types.pyi) with the enum
values of cpp/include/cudf/types.hpp:99-217."""
from enum import IntEnum

import numpy as np


class TypeId(IntEnum):
    EMPTY = 0
    INT8 = 1
    INT16 = 2
    INT32 = 3
    INT64 = 4
    UINT8 = 5
    UINT16 = 6
    UINT32 = 7
    UINT64 = 8
    FLOAT32 = 9
    FLOAT64 = 10
    BOOL8 = 11
    TIMESTAMP_DAYS = 12
    TIMESTAMP_SECONDS = 13
    TIMESTAMP_MILLISECONDS = 14
    TIMESTAMP_MICROSECONDS = 15
    TIMESTAMP_NANOSECONDS = 16
    DURATION_DAYS = 17
    DURATION_SECONDS = 18
    DURATION_MILLISECONDS = 19
    DURATION_MICROSECONDS = 20
    DURATION_NANOSECONDS = 21
    DICTIONARY32 = 22
    STRING = 23
    LIST = 24
    DECIMAL32 = 25
    DECIMAL64 = 26
    DECIMAL128 = 27
    STRUCT = 28


class NullPolicy(IntEnum):
    EXCLUDE = 0
    INCLUDE = 1


class NullEquality(IntEnum):
    EQUAL = 0
    UNEQUAL = 1


class Sorted(IntEnum):
    NO = 0
    YES = 1


class Order(IntEnum):
    ASCENDING = 0
    DESCENDING = 1


class NullOrder(IntEnum):
    AFTER = 0
    BEFORE = 1


class OutOfBoundsPolicy(IntEnum):
    NULLIFY = 0
    DONT_CHECK = 1


_NP_OF = {
    TypeId.INT8: np.int8, TypeId.INT16: np.int16, TypeId.INT32: np.int32, TypeId.INT64: np.int64,
    TypeId.UINT8: np.uint8, TypeId.UINT16: np.uint16, TypeId.UINT32: np.uint32, TypeId.UINT64: np.uint64,
    TypeId.FLOAT32: np.float32, TypeId.FLOAT64: np.float64, TypeId.BOOL8: np.bool_,
    TypeId.TIMESTAMP_DAYS: np.int32, TypeId.TIMESTAMP_SECONDS: np.int64, TypeId.TIMESTAMP_MILLISECONDS: np.int64,
    TypeId.TIMESTAMP_MICROSECONDS: np.int64, TypeId.TIMESTAMP_NANOSECONDS: np.int64,
    TypeId.DURATION_DAYS: np.int32, TypeId.DURATION_SECONDS: np.int64, TypeId.DURATION_MILLISECONDS: np.int64,
    TypeId.DURATION_MICROSECONDS: np.int64, TypeId.DURATION_NANOSECONDS: np.int64,
    TypeId.DECIMAL32: np.int32, TypeId.DECIMAL64: np.int64,
}
_ID_OF_NP = {np.dtype(np.int8): TypeId.INT8, np.dtype(np.int16): TypeId.INT16, np.dtype(np.int32): TypeId.INT32,
             np.dtype(np.int64): TypeId.INT64, np.dtype(np.uint8): TypeId.UINT8, np.dtype(np.uint16): TypeId.UINT16,
             np.dtype(np.uint32): TypeId.UINT32, np.dtype(np.uint64): TypeId.UINT64,
             np.dtype(np.float32): TypeId.FLOAT32, np.dtype(np.float64): TypeId.FLOAT64,
             np.dtype(np.bool_): TypeId.BOOL8}


class DataType:
    def __init__(self, id: TypeId, scale: int = 0):
        self._id = TypeId(id)
        self._scale = scale

    def id(self) -> TypeId:
        return self._id

    def scale(self) -> int:
        return self._scale

    def numpy_dtype(self):
        return np.dtype(_NP_OF[self._id])

    def __eq__(self, other):
        return isinstance(other, DataType) and other._id == self._id and other._scale == self._scale

    def __hash__(self):
        return hash((self._id, self._scale))

    def __repr__(self):
        return f"DataType({self._id.name})"

    @staticmethod
    def from_numpy(dtype) -> "DataType":
        return DataType(_ID_OF_NP[np.dtype(dtype)])
