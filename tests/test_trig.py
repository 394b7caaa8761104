"""The build's own sincos / atan2 (gridworld_amd/csrc/igw_trig.h, compiled for the host by
oracle/Makefile): correctly rounded vs mpmath on random arguments, consistent with the walking LUT,
and how often glibc (what the Python reference calls) differs from it."""
import math
import re

import mpmath
import numpy as np

from oracle import oracle as O

mpmath.mp.prec = 300


def _cr(f, *a):
    return float(f(*[mpmath.mpf(float(v)) for v in a]))


def _sincos(x):
    T = O.trig_host()
    x = np.ascontiguousarray(x, np.float64)
    s, c = np.zeros_like(x), np.zeros_like(x)
    T.igw_host_sincos_array(x.ctypes.data, s.ctypes.data, c.ctypes.data, len(x))
    return s, c


def _atan2(y, x):
    T = O.trig_host()
    y, x = np.ascontiguousarray(y, np.float64), np.ascontiguousarray(x, np.float64)
    o = np.zeros_like(x)
    T.igw_host_atan2_array(y.ctypes.data, x.ctypes.data, o.ctypes.data, len(x))
    return o


def test_sincos_correctly_rounded():
    rng = np.random.RandomState(1)
    # the arguments flying mode produces: radians of angles in [-720, 720] degrees, plus tiny ones
    deg = np.concatenate([rng.uniform(-720, 720, 20000), rng.uniform(-90, 90, 10000),
                          np.float32(rng.uniform(-5, 5, 5000)).astype(np.float64), rng.uniform(-1e-6, 1e-6, 1000)])
    x = deg * (math.pi / 180.0)
    s, c = _sincos(x)
    bad = 0
    for xi, si, ci in zip(x, s, c):
        bad += (si != _cr(mpmath.sin, xi)) + (ci != _cr(mpmath.cos, xi))
    assert bad == 0, f'{bad} of {2 * len(x)} results not correctly rounded'
    # glibc vs ours: informational bound (SURVEY F12 measured ~0.15 %), never more than one ulp
    ls = np.array([math.sin(v) for v in x])  # CPython math == glibc (numpy has its own SIMD loops)
    lc = np.array([math.cos(v) for v in x])
    diff = np.mean(s != ls) + np.mean(c != lc)
    print('glibc differs from the correctly rounded value on %.3f %% of sin/cos calls' % (50 * diff))
    assert diff < 0.01
    assert (np.abs(s - ls) <= np.spacing(np.abs(s))).all()
    assert (np.abs(c - lc) <= np.spacing(np.abs(c))).all()


def test_atan2_correctly_rounded():
    rng = np.random.RandomState(2)
    y = np.concatenate([np.float32(rng.uniform(-1, 1, 20000)).astype(np.float64), rng.uniform(-1e-3, 1e-3, 2000),
                        rng.uniform(-10, 10, 3000)])
    x = np.concatenate([np.float32(rng.uniform(-1, 1, 20000)).astype(np.float64), rng.uniform(-1, 1, 2000),
                        rng.uniform(-1e-3, 1e-3, 3000)])
    a = _atan2(y, x)
    bad = sum(ai != _cr(mpmath.atan2, yi, xi) for yi, xi, ai in zip(y, x, a))
    assert bad == 0, f'{bad} of {len(x)} results not correctly rounded'
    libm = np.array([math.atan2(yi, xi) for yi, xi in zip(y, x)])  # CPython math == glibc (numpy has its own SIMD loops)
    assert np.mean(a != libm) < 0.01
    assert (np.abs(a - libm) <= np.spacing(np.abs(a))).all()


def _variant(tmp_path, scale):
    """The host trig library with the quick evaluations' error bound E scaled (tests only)."""
    import ctypes
    import subprocess
    so = str(tmp_path / ('trig_e%g.so' % scale))
    subprocess.check_call(['g++', '-O2', '-fPIC', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-shared',
                           '-DIGW_QUICK_E_SCALE=(%r)' % float(scale), '-o', so, O._HERE + '/igw_trig_host.cpp', '-lm'])
    return ctypes.CDLL(so)


def test_quick_evaluations_never_accept_a_wrong_rounding(tmp_path):
    """igw_sincos / igw_atan2 = a quick evaluation with an error bound E (Ziv's strategy) and the full
    double-double one when hi + (lo +- E) straddles a rounding boundary.  Whenever the quick one accepts, its
    result must be the accurate one's -- also with E shrunk 16 times (the margin of the bound) -- and it must
    accept almost always (it is the fast path of the flying kernel)."""
    import ctypes
    rng = np.random.RandomState(7)
    n = 400000
    deg = np.concatenate([rng.uniform(-720, 720, n), rng.uniform(-90, 90, n),
                          np.float32(rng.uniform(-5, 5, n // 4)).astype(np.float64), rng.uniform(-1e-3, 1e-3, n // 8),
                          90 * rng.randint(-8, 9, n // 4) + rng.uniform(-1.5, 1.5, n // 4), rng.uniform(-1e-9, 1e-9, n // 8)])
    x = deg * (math.pi / 180.0)
    # and the doubles around multiples of pi/2 (where the quick reduction cancels), small and large multiples
    near = []
    for k in np.concatenate([np.arange(-64, 65), rng.randint(-12000, 12000, 2000), rng.randint(-(1 << 19), 1 << 19, 2000)]):
        v = float(mpmath.mpf(int(k)) * mpmath.pi / 2)
        near += [v, np.nextafter(v, np.inf), np.nextafter(v, -np.inf), v * (1 + 2.0 ** -45), v * (1 - 2.0 ** -38), v + 2.0 ** -41, v - 2.0 ** -39]
    x = x[x != 0]
    n_typical = len(x)   # (the acceptance rate is asserted on these: the directed ones are MEANT to be refused)
    x = np.concatenate([x, np.array(near), rng.uniform(-1e6, 1e6, n // 8) * (math.pi / 180.0)])
    x = np.ascontiguousarray(x[x != 0])
    f32 = lambda a: np.float32(a).astype(np.float64)  # noqa: E731
    ya = np.concatenate([f32(rng.uniform(-1, 1, n)), rng.uniform(-1, 1, n), rng.uniform(-1e-3, 1e-3, n // 4), rng.uniform(-1e-9, 1e-9, n // 8)])
    xa = np.concatenate([f32(rng.uniform(-1, 1, n)), rng.uniform(-1, 1, n), rng.uniform(-1, 1, n // 4), rng.uniform(-1, 1, n // 8)])
    keep = (ya != 0) & (xa != 0)
    ya, xa = np.ascontiguousarray(ya[keep]), np.ascontiguousarray(xa[keep])
    for scale, min_accept in ((1.0, 0.9999), (1.0 / 16, 0.99999)):
        T = _variant(tmp_path, scale)
        vp, lg = ctypes.c_void_p, ctypes.c_long
        T.igw_host_sincos_quick_array.argtypes = [vp, vp, vp, vp, lg]
        T.igw_host_sincos_accurate_array.argtypes = [vp, vp, vp, lg]
        T.igw_host_atan2_quick_array.argtypes = [vp, vp, vp, vp, lg]
        T.igw_host_atan2_accurate_array.argtypes = [vp, vp, vp, lg]
        s, c, sa, ca = (np.zeros_like(x) for _ in range(4))
        ok = np.zeros(len(x), np.uint8)
        T.igw_host_sincos_quick_array(x.ctypes.data, s.ctypes.data, c.ctypes.data, ok.ctypes.data, len(x))
        T.igw_host_sincos_accurate_array(x.ctypes.data, sa.ctypes.data, ca.ctypes.data, len(x))
        acc = ok.astype(bool)
        assert not (acc & ((s != sa) | (c != ca))).any(), f'sincos: wrong acceptance at E x {scale}'
        assert acc[:n_typical].mean() >= min_accept, (scale, acc[:n_typical].mean())
        q, qa = np.zeros_like(xa), np.zeros_like(xa)
        ok = np.zeros(len(xa), np.uint8)
        T.igw_host_atan2_quick_array(ya.ctypes.data, xa.ctypes.data, q.ctypes.data, ok.ctypes.data, len(xa))
        T.igw_host_atan2_accurate_array(ya.ctypes.data, xa.ctypes.data, qa.ctypes.data, len(xa))
        acc = ok.astype(bool)
        assert not (acc & ((q != qa) | (np.signbit(q) != np.signbit(qa)))).any(), f'atan2: wrong acceptance at E x {scale}'
        assert acc.mean() >= min_accept, (scale, acc.mean())
    # and the accurate evaluations alone are correctly rounded (the public functions are tested above)
    idx = rng.choice(len(x), 4000, replace=False)
    assert all(sa[i] == _cr(mpmath.sin, x[i]) and ca[i] == _cr(mpmath.cos, x[i]) for i in idx)
    idx = rng.choice(len(xa), 4000, replace=False)
    assert all(qa[i] == _cr(mpmath.atan2, ya[i], xa[i]) for i in idx)


def test_sincos_near_multiples_of_half_pi_and_large_arguments():
    """The quick evaluation's reduction (33-bit pieces of pi/2, one two_sum) leaves arguments within 2^-40 of a
    multiple of pi/2 to the accurate evaluation and widens its error bound with |kd|: results stay correctly
    rounded at the doubles around k pi/2 up to the largest k the camera bound allows (1e6 degrees) and beyond."""
    rng = np.random.RandomState(11)
    ks = np.concatenate([np.arange(-40, 41), rng.randint(-12000, 12000, 300), rng.randint(-(1 << 19), 1 << 19, 300)])
    xs = []
    for k in ks:
        c = float(mpmath.mpf(int(k)) * mpmath.pi / 2)
        v = c
        for _ in range(4):
            v = np.nextafter(v, -np.inf)
        for _ in range(9):
            xs.append(v)
            v = np.nextafter(v, np.inf)
    xs += list(rng.uniform(-1e6, 1e6, 3000) * (math.pi / 180.0))        # the whole camera range
    xs += list(rng.uniform(-1.6e6, 1.6e6, 1000))                          # up to 2^20 radians
    x = np.array([v for v in xs if v != 0.0])
    s, c = _sincos(x)
    bad = [(xi, si, ci) for xi, si, ci in zip(x, s, c) if si != _cr(mpmath.sin, xi) or ci != _cr(mpmath.cos, xi)]
    assert not bad, bad[:5]


def test_special_values():
    T = O.trig_host()
    assert T.igw_host_sin(0.0) == 0.0 and math.copysign(1, T.igw_host_sin(-0.0)) == -1 and T.igw_host_cos(-0.0) == 1.0
    for y, x in ((0.0, 1.0), (-0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (0.5, -0.0), (-1.0, -1.0),
                 (1.0, 1.0), (1e-300, 1.0), (1.0, 1e-300), (-0.25, 0.75)):
        got, want = T.igw_host_atan2(y, x), math.atan2(y, x)
        assert got == want and math.copysign(1, got) == math.copysign(1, want), (y, x, got, want)


def test_lut_consistent_with_general_path():
    """Every walking-LUT entry equals the general sincos at the same argument (both correctly rounded),
    so poses that leave the 5-degree lattice continue seamlessly."""
    src = open(O._CSRC + '/igw_trig_lut.h').read()
    rows = re.findall(r'\{(-?0x[0-9a-fp.+-]+), (-?0x[0-9a-fp.+-]+)\}, // (-?\d+) deg', src)
    assert len(rows) == 127
    x = np.array([int(d) * (math.pi / 180.0) for _, _, d in rows])
    s, c = _sincos(x)
    for (ch, sh, d), si, ci in zip(rows, s, c):
        assert float.fromhex(ch) == ci == math.cos(math.radians(int(d)))
        assert float.fromhex(sh) == si == math.sin(math.radians(int(d)))


def test_oracle_device_trig_mode_switch():
    """The oracle's trig hooks: libm by default (= reference), product trig on request."""
    import golden_replay as GR
    fx = GR.load_fixture('s4_fly_cdm')
    try:
        O.use_device_trig(True)
        env = O.OracleEnv(**fx['kwargs'])
        env.set_task(fx['targets'][0], fx['starts'][0])
        env.reset()
        for t in range(50):
            env.step(dict(movement=fx['act_movement'][0, t], camera=fx['act_camera'][0, t],
                          inventory=fx['act_inventory'][0, t], placement=fx['act_placement'][0, t]))
        a = env.internal()
    finally:
        O.use_device_trig(False)
    # same trajectory up to last-bit trig differences
    assert np.allclose(a[:6], fx['internal'][0, 49][:6], rtol=0, atol=1e-9)


def test_oracle_device_trig_replays_cr_libm_reference_bit_for_bit():
    """"CR-libm oracle" (SURVEY.md section 8a): the Python reference run with correctly rounded sin/cos/atan2.
    The build's trig is correctly rounded, so the oracle with that trig reproduces the whole trajectory --
    float64 internals included -- which pins igw_trig.h to the reference's call sites, not just to mpmath."""
    import golden_replay as GR
    try:
        O.use_device_trig(True)
        for name in GR.CRLIBM_FIXTURES:
            fx = GR.load_fixture(name)
            assert GR.replay(fx, GR.OracleDriver(fx), check_internal=True) == fx['done'].size
    finally:
        O.use_device_trig(False)
