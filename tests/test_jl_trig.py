"""Julia Base's elementary functions of the step path — sin, cos (fresnel_coefficients), acos (angle3d), tan and atan(y, x) (gauss_parameters) —
restated twice from the algorithms Base ports (FreeBSD msun / fdlibm): `oracle/jl_trig.hpp` for the oracle, `csrc/bmo_jlmath.hpp` for the engine
(the same header runs on the device and in the host emulator; `bmo_selftest` compares device and host).  Julia is not available here, so the pins are:
  * faithful rounding (< 1 ulp) against 40-digit references on dense samples — what a wrong constant or a swapped branch breaks;
  * the reference's own exact assertion runtests.jl:157, `real(rp) == 0` at Brewster's angle, which no C library's sin satisfies;
  * oracle == lane code, bit for bit, on 10^5 arguments per function."""
import math

import numpy as np
import pytest

from parity import emu_jl_trig

mp = pytest.importorskip("mpmath")
mp.mp.dps = 40

FUNCS = {"sin": (mp.sin, -7.0, 7.0), "cos": (mp.cos, -7.0, 7.0), "tan": (mp.tan, -1.5707, 1.5707), "acos": (mp.acos, -1.0, 1.0), "atan": (mp.atan, -40.0, 40.0)}
WHICH = {"sin": 0, "cos": 1, "tan": 2, "acos": 3, "atan": 4, "atan2": 5}


def _ulp_error(got, x, f):
    t = f(mp.mpf(float(x)))
    tf = float(t)
    return float(abs((mp.mpf(float(got)) - t) / mp.mpf(math.ulp(tf) if tf != 0.0 else 5e-324)))


@pytest.mark.parametrize("name", sorted(FUNCS))
def test_faithful_rounding_against_high_precision(oracle, name):
    f, lo, hi = FUNCS[name]
    rng = np.random.default_rng(20251005)
    xs = np.concatenate([rng.uniform(lo, hi, 6000), rng.uniform(-0.8, 0.8, 1500) if name != "atan" else rng.uniform(-3, 3, 1500)])
    got = oracle.jl_trig(name, xs)
    worst = max(_ulp_error(g, x, f) for g, x in zip(got, xs))
    assert worst < 0.95, (name, worst)  # fdlibm's own bounds are < 1 ulp; a wrong coefficient or branch shows as several
    exact = sum(float(f(mp.mpf(float(x)))) == g for g, x in zip(got, xs))
    assert exact < len(xs)  # ... and they are NOT correctly rounded everywhere: that is the point of restating them


def test_near_multiples_of_half_pi(oracle):
    """The three-constant Cody-Waite reduction (arguments whose high word is pi/2's): sin / cos / tan stay faithful where x - n pi/2 cancels."""
    xs = []
    for q in (1, 2, 3, 4):
        x0 = q * (math.pi / 2)
        for e in range(-40, 41):
            xs.append(x0 + e * math.ulp(x0))
        xs += [x0 + d for d in (1e-13, -1e-13, 1e-9, -1e-9, 1e-5, -1e-5)]
    xs = np.array(xs)
    for name, f in (("sin", mp.sin), ("cos", mp.cos)):
        got = oracle.jl_trig(name, xs)
        assert max(_ulp_error(g, x, f) for g, x in zip(got, xs)) < 0.95, name
    got = oracle.jl_trig("tan", xs)
    assert max(_ulp_error(g, x, mp.tan) for g, x in zip(got, xs)) < 1.5  # k_tan.c's own bound next to the poles is below 1 ulp; margin for -1 / tan


def test_special_values(oracle):
    assert oracle.jl_trig("sin", [0.0, -0.0, 1e-300]).tolist() == [0.0, -0.0, 1e-300] and math.copysign(1, oracle.jl_trig("sin", [-0.0])[0]) == -1
    assert oracle.jl_trig("tan", [-0.0])[0] == 0.0 and math.copysign(1, oracle.jl_trig("tan", [-0.0])[0]) == -1
    assert oracle.jl_trig("cos", [0.0, 1e-9]).tolist() == [1.0, 1.0]
    assert oracle.jl_trig("acos", [1.0, -1.0, 0.0]).tolist() == [0.0, math.pi, math.pi / 2]
    assert oracle.jl_trig("sin", [math.pi])[0] == 1.2246467991473532e-16  # the residue the reference's rotation matrices carry
    assert oracle.jl_trig("cos", [math.pi / 2])[0] == 6.123233995736766e-17
    assert oracle.jl_trig("atan", [1.5])[0] == 0.982793723247329  # a table entry of s_atan.c: atan(3/2)
    assert np.isnan(oracle.jl_trig("sin", [np.inf, np.nan, 1e300])).all()  # Base throws for Inf; beyond 2^20 pi/2 Payne-Hanek is not restated
    # atan(1, x) as gauss_parameters calls it (Gaussian.jl:345): x = 0 -> pi/2, x = Inf -> 0, x = 1 -> atan(1)
    got = oracle.jl_trig("atan2", [0.0, np.inf, 1.0, 1e-30, 2.0], [1.0] * 5)
    assert got[0] == math.pi / 2 and got[1] == 0.0 and got[2] == math.pi / 4 and got[3] == math.pi / 2
    assert _ulp_error(got[4], 0.5, mp.atan) < 0.95


def test_brewster_zero_is_exact(oracle):
    """runtests.jl:153-157: `real(rp) ≈ 0` is `== 0`.  Base's sin(atan(1.5)) is one unit below the correctly rounded value; with it the
    numerator -n² cos θ + sqrt(n² - sin² θ) cancels exactly, with the C library's it is -2.2e-16."""
    thb = math.atan(1.5)
    s = oracle.jl_trig("sin", [thb])[0]
    assert s == 0.8320502943378436 and math.sin(thb) == 0.8320502943378437
    assert oracle.jl_trig("cos", [thb])[0] == 0.5547001962252291
    rs, rp, ts, tp = oracle.fresnel_coefficients(thb, 1.5)
    assert rp.real == 0.0 and rp.imag == 0.0


@pytest.mark.parametrize("name", sorted(WHICH))
def test_lane_code_equals_oracle_bit_for_bit(oracle, name):
    rng = np.random.default_rng(7)
    n = 100000
    if name == "acos":
        xs = np.concatenate([rng.uniform(-1, 1, n - 4), [1.0, -1.0, 0.5, -0.5]])
    elif name in ("atan", "atan2"):
        xs = rng.standard_normal(n) * np.exp(rng.uniform(-40, 40, n))
    else:
        xs = np.concatenate([rng.uniform(-8, 8, n // 2), rng.uniform(-1.6, 1.6, n // 2)])
    ys = np.where(rng.random(n) < 0.5, 1.0, rng.standard_normal(n)) if name == "atan2" else None
    a = oracle.jl_trig(name, xs, ys)
    b = emu_jl_trig(WHICH[name], xs, ys)
    assert np.array_equal(a.view(np.int64), b.view(np.int64)), (name, xs[np.flatnonzero(a.view(np.int64) != b.view(np.int64))[:5]])


def test_engine_library_export_equals_oracle(oracle):
    """`bmo_jl_trig` (include/bmo.h): the host build of csrc/bmo_jlmath.hpp inside the engine library — what `bmo_selftest` compares the device with
    and what a maintainer compares with Base (julia/GPUSystem.jl `check_elementary_functions`).  Loads without a GPU."""
    import bmo_amd as bmo

    lib = bmo.abi.load_engine()
    rng = np.random.default_rng(11)
    for name, which in WHICH.items():
        xs = rng.uniform(-1, 1, 4000) if name == "acos" else rng.uniform(-7, 7, 4000)
        ys = np.ones(4000) if name == "atan2" else None
        want = oracle.jl_trig(name, xs, ys)
        got = np.array([lib.bmo_jl_trig(which, float(x), 1.0) for x in xs])
        assert np.array_equal(want.view(np.int64), got.view(np.int64)), name
    assert math.isnan(lib.bmo_jl_trig(9, 0.5, 0.0))
