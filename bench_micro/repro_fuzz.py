#!/usr/bin/env python3
"""Replays one seed of tests/test_groupby_gpu.py::test_fuzz_against_oracle outside pytest and reports, per aggregation,
how many groups differ from the oracle (run on the GPU box from the repo root: python bench_micro/repro_fuzz.py SEED)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kat
import gpu_backend as G
from oracle import oracle
from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
from test_groupby_gpu import _FUZZ_KEY_TYPES, _FUZZ_VAL_TYPES, _FUZZ_AGGS

seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
n = int(rng.choice([0, 1, 37, 5_000, 80_000, 700_000]))
if seed % 4 == 3:
    os.environ["CUDF_AMD_GB_LDS_KB"] = str(int(rng.choice([16, 24, 32])))


def column(tname, distinct, nullable, offset):
    npt = NP_OF_TYPE_ID[TYPE_ID[tname]]
    m = n + offset
    if tname == "bool":
        data = rng.integers(0, 2, m).astype(npt)
    elif np.dtype(npt).kind == "f":
        data = (rng.integers(0, distinct, m) - distinct // 2).astype(npt) * npt(0.25)
    else:
        info = np.iinfo(npt)
        data = rng.integers(0, min(distinct, int(info.max) - 1), m).astype(npt)
    valid = (rng.random(m) > 0.15) if nullable else None
    return HostColumn(data, valid, tname, offset=offset) if offset else HostColumn(data[:n] if offset == 0 else data, valid, tname)


nkeys = int(rng.integers(1, 4))
spread = int(rng.choice([3, 40, 3000]))
keys = [column(str(rng.choice(_FUZZ_KEY_TYPES)), spread, bool(rng.random() < 0.4), int(rng.choice([0, 0, 5]))) for _ in range(nkeys)]
requests = []
for _ in range(int(rng.integers(1, 3))):
    vt = str(rng.choice(_FUZZ_VAL_TYPES))
    vals = column(vt, 9, bool(rng.random() < 0.5), int(rng.choice([0, 0, 3])))
    kinds = [str(k) for k in rng.choice(_FUZZ_AGGS, size=int(rng.integers(1, 5)), replace=False)]
    requests.append((vals, kinds))
include = bool(rng.random() < 0.5)
print(f"seed {seed}: n={n} keys={[(k.type_id, k.offset, k.valid is not None) for k in keys]} "
      f"requests={[(v.type_id, v.offset, v.valid is not None, kk) for v, kk in requests]} include={include}", flush=True)


def run(reqs, label):
    got = kat.sort_groups(*G.groupby(keys, reqs, include_null_keys=include))
    exp = kat.sort_groups(*oracle.groupby(keys, reqs, include_null_keys=include))
    out = [f"groups {len(got[0][0][0])}/{len(exp[0][0][0])} path={G.last_path.name}"]
    for (vals, kinds), ra, re_ in zip(reqs, got[1], exp[1]):
        for kind, a, e in zip(kinds, ra, re_):
            av = np.ones(len(a[0]), bool) if a[1] is None else a[1]
            ev = np.ones(len(e[0]), bool) if e[1] is None else e[1]
            if len(av) != len(ev):
                out.append(f"{kind}: SIZE {len(av)} vs {len(ev)}")
                continue
            dv = int((av != ev).sum())
            both = av & ev
            dd = int((~np.isclose(a[0][both].astype(np.float64), e[0][both].astype(np.float64), rtol=1e-9, atol=1e-9, equal_nan=True)).sum())
            out.append(f"{kind}: validity diffs {dv}, data diffs {dd}")
    print(f"{label:40s} " + " | ".join(out), flush=True)


run(requests, "as fuzzed")
for vals, kinds in requests:
    for k in kinds:
        run([(vals, [k])], f"alone: {k}")
run([(vals, ["count_valid", "count_all", "sum"]) for vals, _ in requests], "count_valid+count_all+sum")
for var, val in (("CUDF_AMD_GB_EXACT", "1"), ("CUDF_AMD_GB_P", "256"), ("CUDF_AMD_GB_LDS_KB", "64")):
    os.environ[var] = val
    run(requests, f"{var}={val}")
    del os.environ[var]
