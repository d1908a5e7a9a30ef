"""Arrow C Data Interface in and out (include/cudf/interop.hpp, cudf_amd/interop.py): round trips of every supported
type with nulls and offsets, then the north-star shape end to end — an Arrow table goes in, the groupby / join result
comes back as Arrow — checked against pyarrow's own CPU group_by / join on the same input."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
pa = pytest.importorskip("pyarrow")


@pytest.fixture(scope="module")
def I(gpu):
    from cudf_amd import interop
    return interop


def _all_types_batch(n, nulls, seed=5):
    rng = np.random.default_rng(seed)
    cols, names = [], []
    for name, typ, gen in [
        ("i8", pa.int8(), lambda: rng.integers(-100, 100, n)), ("u8", pa.uint8(), lambda: rng.integers(0, 200, n)),
        ("i16", pa.int16(), lambda: rng.integers(-30000, 30000, n)), ("u16", pa.uint16(), lambda: rng.integers(0, 60000, n)),
        ("i32", pa.int32(), lambda: rng.integers(-2**31, 2**31 - 1, n)), ("u32", pa.uint32(), lambda: rng.integers(0, 2**32 - 1, n)),
        ("i64", pa.int64(), lambda: rng.integers(-2**62, 2**62, n)), ("u64", pa.uint64(), lambda: rng.integers(0, 2**63, n).astype(np.uint64)),
        ("f32", pa.float32(), lambda: rng.random(n).astype(np.float32)), ("f64", pa.float64(), lambda: rng.random(n)),
        ("b", pa.bool_(), lambda: rng.random(n) > 0.5),
        ("ts_ms", pa.timestamp("ms"), lambda: rng.integers(0, 2**40, n)), ("ts_ns", pa.timestamp("ns"), lambda: rng.integers(0, 2**60, n)),
        ("d32", pa.date32(), lambda: rng.integers(0, 20000, n).astype(np.int32)), ("dur_us", pa.duration("us"), lambda: rng.integers(0, 2**50, n)),
    ]:
        mask = (rng.random(n) < 0.2) if nulls else None  # True = null (pyarrow convention)
        cols.append(pa.array(gen(), type=typ, mask=mask))
        names.append(name)
    return pa.RecordBatch.from_arrays(cols, names=names)


@pytest.mark.parametrize("nulls", [False, True])
@pytest.mark.parametrize("n", [0, 1, 1000, 4099])
def test_round_trip_all_types(I, n, nulls):
    batch = _all_types_batch(n, nulls)
    back = I.to_arrow(I.from_arrow(batch), names=batch.schema.names)
    assert back.num_rows == n and back.schema.names == batch.schema.names
    for name in batch.schema.names:
        assert back.column(name).type == batch.column(name).type, name
        assert back.column(name).equals(batch.column(name)), name


@pytest.mark.parametrize("offset", [1, 7, 8, 33, 129])
def test_sliced_input_offsets(I, offset):
    """A sliced RecordBatch exports arrays with a non-zero offset: data pointers and both bitmaps (validity, booleans)
    have to be read from that bit."""
    batch = _all_types_batch(1500, True, seed=6).slice(offset, 1000)
    back = I.to_arrow(I.from_arrow(batch), names=batch.schema.names)
    for name in batch.schema.names:
        assert back.column(name).equals(batch.column(name)), name


def test_table_with_chunks_and_errors(I):
    t = pa.concat_tables([pa.table({"k": [1, 2], "v": [0.5, None]}), pa.table({"k": [3], "v": [1.5]})])
    back = I.to_arrow(I.from_arrow(t), names=["k", "v"])
    assert back.column("k").to_pylist() == [1, 2, 3] and back.column("v").to_pylist() == [0.5, None, 1.5]
    with pytest.raises(TypeError):  # cudf::data_type_error: strings are not on this path
        I.from_arrow(pa.table({"s": ["a", "b"]}))


def test_arrow_in_groupby_arrow_out_matches_pyarrow(I):
    """Arrow table in -> cudf::groupby SUM/COUNT/MEAN/MIN/MAX -> Arrow out, against pyarrow's group_by on the same table."""
    import cudf_amd
    from cudf_amd import aggregation as agg, groupby as gb
    from cudf_amd.types import NullPolicy
    rng = np.random.default_rng(8)
    n = 200_000
    k = rng.integers(0, 5000, n)
    v = rng.random(n)
    t = pa.table({"k": pa.array(k, pa.int64()), "v": pa.array(v, pa.float64(), mask=rng.random(n) < 0.1)})
    dev = I.from_arrow(t)
    kc, vc = dev.columns()
    g = gb.GroupBy(cudf_amd.Table([kc]))
    keys, res = g.aggregate([gb.GroupByRequest(vc, [agg.sum(), agg.count(NullPolicy.EXCLUDE), agg.mean(), agg.min(), agg.max()])])
    got = pa.Table.from_batches([I.to_arrow(cudf_amd.Table(keys.columns() + (res[0].columns() if hasattr(res[0], "columns") else list(res[0]))), names=["k", "sum", "count", "mean", "min", "max"])]).sort_by("k")
    exp = t.group_by("k").aggregate([("v", "sum"), ("v", "count"), ("v", "mean"), ("v", "min"), ("v", "max")]).sort_by("k")
    assert got.column("k").equals(exp.column("k"))
    assert got.column("count").to_pylist() == exp.column("v_count").to_pylist()
    assert got.column("min").equals(exp.column("v_min")) and got.column("max").equals(exp.column("v_max"))
    for a, b in (("sum", "v_sum"), ("mean", "v_mean")):
        x, y = got.column(a).to_numpy(zero_copy_only=False), exp.column(b).to_numpy(zero_copy_only=False)
        assert np.array_equal(np.isnan(x), np.isnan(y))
        assert np.allclose(x[~np.isnan(x)], y[~np.isnan(y)], rtol=1e-12, atol=0.0)


def test_arrow_in_join_arrow_out_matches_pyarrow(I):
    """Arrow tables in -> cudf::inner_join + gather of the payload columns -> Arrow out, against pyarrow's join."""
    import cudf_amd
    from cudf_amd import join as J, partitioning as P
    rng = np.random.default_rng(9)
    nl, nr = 50_000, 8_000
    left = pa.table({"k": pa.array(rng.integers(0, 12_000, nl), pa.int64(), mask=rng.random(nl) < 0.05), "lp": pa.array(rng.random(nl))})
    right = pa.table({"k": pa.array(rng.permutation(12_000)[:nr], pa.int64(), mask=rng.random(nr) < 0.05), "rp": pa.array(rng.random(nr))})
    dl, dr = I.from_arrow(left), I.from_arrow(right)
    from cudf_amd.types import NullEquality
    li, ri = J.inner_join(cudf_amd.Table([dl.columns()[0]]), cudf_amd.Table([dr.columns()[0]]), NullEquality.UNEQUAL)
    gl, gr = P.gather(dl, li), P.gather(dr, ri)
    got = pa.Table.from_batches([I.to_arrow(cudf_amd.Table(gl.columns() + [gr.columns()[1]]), names=["k", "lp", "rp"])])
    exp = left.join(right, keys="k", join_type="inner")  # pyarrow: null keys do not match
    key = lambda t: sorted(zip(t.column("k").to_pylist(), t.column("lp").to_pylist(), t.column("rp").to_pylist()))
    assert got.num_rows == exp.num_rows and key(got) == key(exp)
