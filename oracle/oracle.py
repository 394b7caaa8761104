"""ctypes/numpy front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product package `gridworld_amd` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libigw_oracle.so')
CELLS = 1089
WALKING, FLYING = 0, 1


class Config(C.Structure):
    _fields_ = [('action_space', C.c_int32), ('select_and_place', C.c_int32),
                ('size_reward', C.c_int32), ('max_steps', C.c_int32),
                ('right_placement_scale', C.c_double), ('wrong_placement_scale', C.c_double)]


class BatchOut(C.Structure):
    _fields_ = [('agentPos', C.c_void_p), ('inventory', C.c_void_p), ('compass', C.c_void_p),
                ('reward', C.c_void_p), ('done', C.c_void_p), ('grid', C.c_void_p)]


_TRIG_PATH = os.path.join(_HERE, 'libigw_trig_host.so')
_CSRC = os.path.join(_HERE, '..', 'gridworld_amd', 'csrc')


def _stale(lib, srcs):
    return not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in srcs)


def build(force=False):
    """Compiles oracle/libigw_oracle.so and oracle/libigw_trig_host.so (oracle/Makefile)."""
    import fcntl
    src = [os.path.join(_HERE, f) for f in ('igw_oracle.c', 'igw_oracle.h', 'Makefile')]
    tsrc = [os.path.join(_HERE, 'igw_trig_host.cpp'), os.path.join(_CSRC, 'igw_trig.h'),
            os.path.join(_CSRC, 'igw_trig_tables.h')]
    if not (force or _stale(_LIB_PATH, src) or _stale(_TRIG_PATH, tsrc)):
        return _LIB_PATH
    with open(os.path.join(_HERE, '.build.lock'), 'w') as lock:  # several processes may start at once
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            for lib_path, deps, target in ((_LIB_PATH, src, 'libigw_oracle.so'),
                                           (_TRIG_PATH, tsrc, 'libigw_trig_host.so')):
                if force or _stale(lib_path, deps):  # re-checked under the lock
                    subprocess.run(['make', '-C', _HERE, '-B', target], check=True, stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB_PATH


_trig = None


def trig_host():
    """Host compile of the product's igw_trig.h (sin / cos / atan2 as the HIP kernels compute them)."""
    global _trig
    if _trig is None:
        build()
        T = C.CDLL(_TRIG_PATH)
        for f in (T.igw_host_sin, T.igw_host_cos):
            f.restype = C.c_double
            f.argtypes = [C.c_double]
        T.igw_host_atan2.restype = C.c_double
        T.igw_host_atan2.argtypes = [C.c_double, C.c_double]
        T.igw_host_sincos_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        T.igw_host_atan2_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        _trig = T
    return _trig


def use_device_trig(on):
    """Oracle trig mode: libm (= the Python reference, default) or the product's own sincos/atan2
    ("device-trig" mode: the oracle then matches the HIP kernels bit-for-bit in flying mode too)."""
    L = lib()
    L.igo_set_trig.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    if on:
        T = trig_host()
        L.igo_set_trig(C.cast(T.igw_host_sin, C.c_void_p), C.cast(T.igw_host_cos, C.c_void_p),
                       C.cast(T.igw_host_atan2, C.c_void_p))
    else:
        L.igo_set_trig(None, None, None)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.igo_create.restype = C.c_void_p
        L.igo_create.argtypes = [C.POINTER(Config)]
        L.igo_destroy.argtypes = [C.c_void_p]
        L.igo_set_task.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.igo_set_initial_pose.argtypes = [C.c_void_p, C.c_void_p]
        L.igo_reset.argtypes = [C.c_void_p]
        L.igo_step_walking.argtypes = [C.c_void_p, C.c_int]
        L.igo_step_flying.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.igo_step_walking_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.igo_batch_step_walking_dict.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                  C.POINTER(BatchOut)]
        L.igo_get_obs.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.igo_get_reward.restype = C.c_double
        L.igo_get_reward.argtypes = [C.c_void_p]
        L.igo_get_done.argtypes = [C.c_void_p]
        L.igo_get_internal.argtypes = [C.c_void_p, C.c_void_p]
        L.igo_get_task_state.argtypes = [C.c_void_p, C.c_void_p]
        L.igo_task_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7
        L.igo_batch_step_walking.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                             C.POINTER(BatchOut)]
        L.igo_batch_step_flying.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int, C.c_int, C.POINTER(BatchOut)]
        L.igo_batch_rollout_walking.restype = C.c_int64
        L.igo_batch_rollout_walking.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64,
                                                C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.igo_batch_rollout_flying.restype = C.c_int64
        L.igo_batch_rollout_flying.argtypes = L.igo_batch_rollout_walking.argtypes
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dense(a):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.int8)
    assert a.size == CELLS
    return a


def sparse_to_dense(blocks):
    """reference Tasks.to_dense for a sparse list (tasks/task.py:168-175)."""
    d = np.zeros((9, 11, 11), np.int8)
    for x, y, z, c in blocks:
        d[int(y) + 1, int(x) + 5, int(z) + 5] = int(c)
    return d


class OracleEnv:
    """One reference-semantics env (gym.make('IGLUGridworld-v0', vector_state=True, render=False))."""

    def __init__(self, action_space='walking', select_and_place=True, size_reward=True, max_steps=250,
                 right_placement_scale=1., wrong_placement_scale=0.1, discretize=True):
        self.flying = action_space == 'flying'
        self.discretize = discretize
        self.cfg = Config(FLYING if self.flying else WALKING, int(select_and_place), int(size_reward),
                          int(max_steps), float(right_placement_scale), float(wrong_placement_scale))
        self.h = C.c_void_p(lib().igo_create(C.byref(self.cfg)))

    def __del__(self):
        if getattr(self, 'h', None) and _lib is not None:
            _lib.igo_destroy(self.h)
            self.h = None

    def set_task(self, target, start=None, full_grid=None, invariant=True):
        """target dense [9,11,11]; start: dense array, sparse list or None."""
        if start is not None and not isinstance(start, np.ndarray):
            start = sparse_to_dense(start)
        t, s, f = _dense(target), _dense(start), _dense(full_grid)
        lib().igo_set_task(self.h, _p(t), _p(s), _p(f), int(invariant))

    def set_initial_pose(self, pose5):
        p = np.ascontiguousarray(pose5, dtype=np.float64)
        lib().igo_set_initial_pose(self.h, _p(p))

    def reset(self):
        lib().igo_reset(self.h)
        return self.obs()

    def step(self, action):
        if isinstance(action, dict) and 'buttons' in action:  # walking, discretize=False
            b = np.ascontiguousarray(action['buttons'], dtype=np.uint8)
            cam = np.ascontiguousarray(action['camera'], dtype=np.float64)
            lib().igo_step_walking_dict(self.h, _p(b), _p(cam))
        elif self.flying:
            mv = np.ascontiguousarray(action['movement'], dtype=np.float64)
            cam = np.ascontiguousarray(action['camera'], dtype=np.float64)
            lib().igo_step_flying(self.h, _p(mv), _p(cam), int(action['inventory']),
                                  int(action['placement']))
        else:
            lib().igo_step_walking(self.h, int(action))
        return self.obs(), lib().igo_get_reward(self.h), bool(lib().igo_get_done(self.h)), {}

    def obs(self):
        ap = np.zeros(5, np.float32)
        inv = np.zeros(6, np.float32)
        comp = np.zeros(1, np.float32)
        grid = np.zeros(CELLS, np.int8)
        lib().igo_get_obs(self.h, _p(ap), _p(inv), _p(comp), _p(grid))
        return {'agentPos': ap, 'inventory': inv, 'compass': comp,
                'grid': grid.reshape(9, 11, 11).astype(np.int32)}

    def internal(self):
        o = np.zeros(8, np.float64)
        lib().igo_get_internal(self.h, _p(o))
        return o

    def task_state(self):
        o = np.zeros(6, np.int32)
        lib().igo_get_task_state(self.h, _p(o))
        return dict(zip(('syn_max_int', 'syn_prev_size', 'syn_target_size', 'env_max_int', 'step_no',
                         'size'), (int(v) for v in o)))


def task_eval(target, grid=None, full_grid=None, invariant=True):
    t, g, f = _dense(target), _dense(grid), _dense(full_grid)
    ts = np.zeros(1, np.int32)
    cnt = np.zeros(4, np.int32)
    mask = np.zeros((4, 21, 21), np.uint8)
    rot = np.zeros((4, 9, 11, 11), np.int8)
    mi = np.zeros(1, np.int32)
    am = np.zeros(3, np.int32)
    lib().igo_task_eval(_p(t), _p(f), int(invariant), _p(g), _p(ts), _p(cnt), _p(mask), _p(rot),
                        _p(mi), _p(am))
    return dict(target_size=int(ts[0]), adm_count=cnt, adm_mask=mask, rot=rot, max_int=int(mi[0]),
                argmax=am)


class OracleBatch:
    """N independent oracle envs stepped by the C batch drivers (pthreads over shards)."""

    def __init__(self, n, **kw):
        self.envs = [OracleEnv(**kw) for _ in range(n)]
        self.n = n
        self.flying = self.envs[0].flying
        self.handles = (C.c_void_p * n)(*[e.h for e in self.envs])
        self.agentPos = np.zeros((n, 5), np.float32)
        self.inventory = np.zeros((n, 6), np.float32)
        self.compass = np.zeros((n,), np.float32)
        self.reward = np.zeros((n,), np.float32)
        self.done = np.zeros((n,), np.uint8)
        self.grid = np.zeros((n, CELLS), np.int8)
        self._out = BatchOut(*[a.ctypes.data for a in (self.agentPos, self.inventory, self.compass,
                                                       self.reward, self.done, self.grid)])

    def set_tasks(self, targets, starts=None, full_grids=None, invariant=True):
        for i, e in enumerate(self.envs):
            e.set_task(targets[i], None if starts is None else starts[i],
                       None if full_grids is None else full_grids[i], invariant=invariant)

    def set_initial_pose(self, poses):
        for i, e in enumerate(self.envs):
            e.set_initial_pose(poses[i])

    def reset(self, mask=None):
        for i, e in enumerate(self.envs):
            if mask is not None and not mask[i]:
                continue
            o = e.reset()
            self.agentPos[i], self.inventory[i], self.compass[i] = o['agentPos'], o['inventory'], o['compass'][0]
            self.grid[i] = o['grid'].reshape(-1)

    def step_walking(self, actions, autoreset=False, nthreads=1):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        lib().igo_batch_step_walking(self.handles, self.n, _p(a), int(autoreset), nthreads,
                                     C.byref(self._out))

    def step_flying(self, movement, camera, inventory, placement, autoreset=False, nthreads=1):
        mv = np.ascontiguousarray(movement, dtype=np.float32)
        cam = np.ascontiguousarray(camera, dtype=np.float32)
        inv = np.ascontiguousarray(inventory, dtype=np.int32)
        pl = np.ascontiguousarray(placement, dtype=np.int32)
        lib().igo_batch_step_flying(self.handles, self.n, _p(mv), _p(cam), _p(inv), _p(pl),
                                    int(autoreset), nthreads, C.byref(self._out))

    def step_walking_dict(self, buttons, camera, autoreset=False, nthreads=1):
        b = np.ascontiguousarray(buttons, dtype=np.uint8)
        cam = np.ascontiguousarray(camera, dtype=np.float32)
        lib().igo_batch_step_walking_dict(self.handles, self.n, _p(b), _p(cam), int(autoreset), nthreads,
                                          C.byref(self._out))

    def rollout_walking(self, T, seed, env_offset=0, autoreset=True, nthreads=1):
        changed = np.zeros(1, np.int64)
        steps = lib().igo_batch_rollout_walking(self.handles, self.n, T, seed, env_offset,
                                                int(autoreset), nthreads, _p(changed))
        return int(steps), int(changed[0])

    def rollout_flying(self, T, seed, env_offset=0, autoreset=True, nthreads=1):
        changed = np.zeros(1, np.int64)
        steps = lib().igo_batch_rollout_flying(self.handles, self.n, T, seed, env_offset,
                                               int(autoreset), nthreads, _p(changed))
        return int(steps), int(changed[0])

    def internals(self):
        return np.stack([e.internal() for e in self.envs])
