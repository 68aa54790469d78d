"""include/mpr_fmath.h: accuracy of the shared float functions against float64 numpy."""
import numpy as np
import pytest


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    ulp = np.maximum(ulp, np.float64(np.finfo(np.float32).tiny) * 2 ** -23)
    return np.abs(got.astype(np.float64) - ref64) / ulp


CASES = [
    ("sin", np.sin, lambda r: r.uniform(-100, 100, 200000)),
    ("cos", np.cos, lambda r: r.uniform(-100, 100, 200000)),
    ("sin", np.sin, lambda r: r.uniform(-1e4, 1e4, 100000)),
    ("asin", np.arcsin, lambda r: r.uniform(-1, 1, 200000)),
    ("acos", np.arccos, lambda r: r.uniform(-1, 1, 200000)),
    ("atan", np.arctan, lambda r: np.concatenate([r.uniform(-10, 10, 100000), r.standard_cauchy(100000) * 100])),
    ("exp", np.exp, lambda r: r.uniform(-87, 88, 200000)),
    ("log", np.log, lambda r: np.exp(r.uniform(-80, 80, 200000))),
]


@pytest.mark.parametrize("name,ref,gen", CASES)
def test_accuracy(orc, name, ref, gen):
    x = gen(np.random.default_rng(0)).astype(np.float32)
    got = orc.fmath(name, x)
    err = ulp_err(got, ref(x.astype(np.float64)))
    assert np.isfinite(err).all()
    # acos near +1 amplifies the argument's rounding; everything else stays within 2 ulp
    assert err.max() <= (4.0 if name == "acos" else 2.0), (name, err.max(), x[err.argmax()])


def test_special_values(orc):
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    assert orc.fmath("exp", [inf, -inf, 0.0, 89.0, -104.0]).tolist() == [inf, 0.0, 1.0, inf, 0.0]
    assert np.isnan(orc.fmath("exp", [nan])).all()
    lg = orc.fmath("log", [0.0, -0.0, -1.0, inf, 1.0])
    assert lg[0] == -inf and lg[1] == -inf and np.isnan(lg[2]) and lg[3] == inf and lg[4] == 0.0
    assert np.isnan(orc.fmath("asin", [1.5, -1.5])).all() and np.isnan(orc.fmath("acos", [1.5, -1.5])).all()
    assert np.isnan(orc.fmath("sin", [inf, nan])).all() and np.isnan(orc.fmath("cos", [-inf])).all()
    at = orc.fmath("atan", [inf, -inf, 0.0])
    assert abs(at[0] - np.pi / 2) < 1e-6 and abs(at[1] + np.pi / 2) < 1e-6 and at[2] == 0.0
    # subnormal results / arguments
    tiny = orc.fmath("exp", [-100.0])[0]
    assert 0 < tiny < np.finfo(np.float32).tiny and abs(tiny / np.exp(-100.0) - 1) < 5e-2
    assert abs(orc.fmath("log", [1e-42])[0] - np.log(np.float64(np.float32(1e-42)))) < 1e-4
