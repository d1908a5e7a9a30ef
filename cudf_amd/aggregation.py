"""Aggregation factories mirroring pylibcudf.aggregation (reference python/pylibcudf/pylibcudf/aggregation.pyi:16-114);
Kind values are those of cudf::aggregation::Kind (cpp/include/cudf/aggregation.hpp:78-121)."""
from enum import IntEnum

from .types import NullPolicy


class Kind(IntEnum):
    SUM = 0
    SUM_OVERFLOW = 1
    PRODUCT = 2
    MIN = 3
    MAX = 4
    COUNT_VALID = 5
    COUNT_ALL = 6
    ANY = 7
    ALL = 8
    SUM_OF_SQUARES = 9
    MEAN = 10
    M2 = 11
    VARIANCE = 12
    STD = 13
    MEDIAN = 14
    QUANTILE = 15
    ARGMAX = 16
    ARGMIN = 17
    NUNIQUE = 18
    NTH_ELEMENT = 19


class Interpolation(IntEnum):  # cudf::interpolation (reference cpp/include/cudf/types.hpp:173-180)
    LINEAR = 0
    LOWER = 1
    HIGHER = 2
    MIDPOINT = 3
    NEAREST = 4
    NEAREST_HALF_UP = 5


class Aggregation:
    def __init__(self, kind: Kind, param: int = None, param2: int = None, quantiles=None):
        self._kind = Kind(kind)
        self._param = param  # ddof of VARIANCE / STD, n of NTH_ELEMENT (cudf_amd_aggregation_request.params)
        self._param2 = param2  # null policy of NTH_ELEMENT / NUNIQUE, interpolation of QUANTILE (.params2)
        self._quantiles = [float(q) for q in quantiles] if quantiles is not None else []

    def param2(self) -> int:
        if self._param2 is not None:
            return int(self._param2)
        return 1 if self._kind == Kind.NTH_ELEMENT else 0  # the factories' defaults (aggregation.hpp:320-367)

    def quantiles(self):
        return list(self._quantiles)

    def kind(self) -> Kind:
        return self._kind

    def param(self, default: int = 0) -> int:
        return default if self._param is None else int(self._param)

    def __repr__(self):
        return f"Aggregation({self._kind.name})" if self._param is None else f"Aggregation({self._kind.name}, {self._param})"


def sum():
    return Aggregation(Kind.SUM)


def sum_with_overflow():
    # pylibcudf.aggregation.sum_with_overflow: struct {sum, overflow} (aggregation.pyi; cudf::make_sum_overflow_aggregation)
    return Aggregation(Kind.SUM_OVERFLOW)


def product():
    return Aggregation(Kind.PRODUCT)


def min():
    return Aggregation(Kind.MIN)


def max():
    return Aggregation(Kind.MAX)


def count(null_handling: NullPolicy = NullPolicy.INCLUDE):
    # pylibcudf's default is INCLUDE (aggregation.pyi:76); EXCLUDE -> COUNT_VALID, INCLUDE -> COUNT_ALL
    return Aggregation(Kind.COUNT_ALL if null_handling == NullPolicy.INCLUDE else Kind.COUNT_VALID)


def sum_of_squares():
    return Aggregation(Kind.SUM_OF_SQUARES)


def mean():
    return Aggregation(Kind.MEAN)


def variance(ddof: int = 1):
    return Aggregation(Kind.VARIANCE, int(ddof))


def std(ddof: int = 1):
    return Aggregation(Kind.STD, int(ddof))


def median():
    return Aggregation(Kind.MEDIAN)


def argmax():
    return Aggregation(Kind.ARGMAX)


def argmin():
    return Aggregation(Kind.ARGMIN)


def nth_element(n: int, null_handling: NullPolicy = NullPolicy.INCLUDE):
    return Aggregation(Kind.NTH_ELEMENT, int(n), int(null_handling))


def nunique(null_handling: NullPolicy = NullPolicy.EXCLUDE):
    return Aggregation(Kind.NUNIQUE, None, int(null_handling))


def quantile(quantiles, interp: Interpolation = Interpolation.LINEAR):
    # pylibcudf.aggregation.quantile(quantiles, interp) (aggregation.pyi; cudf::make_quantile_aggregation)
    return Aggregation(Kind.QUANTILE, None, int(interp), quantiles)


def m2():
    return Aggregation(Kind.M2)
