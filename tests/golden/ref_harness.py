"""Import harness for the Python reference (runs ONLY in the build container).

The reference at /root/reference is pure Python.  This module imports it through
two tiny stand-ins (tests/golden/shims: `gym` class shells and `numba.jit` =
identity; both packages are absent from this image) and records trajectories of
`gym.make('IGLUGridworld-v0', vector_state=True, render=False)` as plain numpy
arrays.  Nothing here travels to the GPU box except the .npz files it writes.

Hygiene rules followed:
  * tasks are built with `starting_grid=[]` or a sparse list, never None
    (DUMMY_TASK itself crashes in step(): env.py:290 `grid - None`);
  * walking actions are Python ints, flying actions Python floats widened from
    float32 (so the reference keeps float64 internals under NumPy >= 2);
  * agent internals (float64 position / rotation / dy, time_int_steps,
    active_block) are recorded next to the float32 observations.
"""
import os
import sys

import numpy as np

REFERENCE_ROOT = '/root/reference'
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shims')
_loaded = {}


def load_reference():
    """Returns (gym, Task, Tasks) with the reference importable."""
    if _loaded:
        return _loaded['gym'], _loaded['Task'], _loaded['Tasks']
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError('reference tree not present (only exists in the build container)')
    sys.dont_write_bytecode = True
    sys.path[:0] = [_SHIMS, REFERENCE_ROOT]
    if not hasattr(np, 'int'):
        np.int = int  # removed in NumPy 1.24; the reference uses it (tasks/task.py:171)
    import gym  # the shim
    import gridworld.env  # noqa: F401  registers the env ids
    from gridworld.tasks.task import Task, Tasks
    _loaded.update(gym=gym, Task=Task, Tasks=Tasks)
    return gym, Task, Tasks


class _CRMath:
    """Drop-in for the `math` module inside gridworld.core.world with CORRECTLY ROUNDED sin / cos / atan2
    (mpmath, 300 bits -> nearest double); everything else is the real `math`.  "CR-libm oracle" mode of
    SURVEY.md section 8a: against such a reference a correctly rounded implementation matches on every bit."""

    def __init__(self):
        import math
        import mpmath
        self._math, self._mp = math, mpmath
        mpmath.mp.prec = 300

    def __getattr__(self, name):
        return getattr(self._math, name)

    def sin(self, x):
        return x if x == 0 else float(self._mp.sin(self._mp.mpf(float(x))))

    def cos(self, x):
        return float(self._mp.cos(self._mp.mpf(float(x))))

    def atan2(self, y, x):
        if y == 0 or x == 0:
            return self._math.atan2(y, x)  # exact special values (signed zeros, +-pi/2, +-pi)
        return float(self._mp.atan2(self._mp.mpf(float(y)), self._mp.mpf(float(x))))


class cr_libm:
    """with cr_libm(): ... -- the reference's world module computes with correctly rounded trig."""

    def __enter__(self):
        load_reference()
        import gridworld.core.world as w
        self._w, self._old = w, w.math
        w.math = _CRMath()

    def __exit__(self, *a):
        self._w.math = self._old


def dense_to_sparse(dense):
    """dense [9,11,11] -> reference-style sparse list [(x, y, z, id)] (tasks/task.py:178-187)."""
    out = []
    ys, xs, zs = np.nonzero(dense)
    for y, x, z in zip(ys, xs, zs):
        out.append((int(x) - 5, int(y) - 1, int(z) - 5, int(dense[y, x, z])))
    return out


def _internals(env):
    a = env.unwrapped.agent
    return [float(a.position[0]), float(a.position[1]), float(a.position[2]),
            float(a.rotation[0]), float(a.rotation[1]), float(a.dy),
            float(a.time_int_steps), float(a.active_block)]


def flying_action(move, cam, inv, place):
    """float32 inputs -> Python floats (exact widening), ints -> Python ints."""
    return {'movement': [float(np.float32(v)) for v in move],
            'camera': [float(np.float32(v)) for v in cam],
            'inventory': int(inv), 'placement': int(place)}


def walking_dict_action(buttons, cam):
    """discretize=False walking action (env.py:60-70); camera as Python floats widened from float32."""
    b = [int(v) for v in buttons]
    return {'forward': b[0], 'back': b[1], 'left': b[2], 'right': b[3], 'jump': b[4], 'attack': b[5],
            'use': b[6], 'hotbar': b[7], 'camera': [float(np.float32(v)) for v in cam]}


def run_batch(kwargs, targets, starts, actions, reset_on_done=True, task_kwargs=None,
              init_pose=None):
    """Runs E independent reference envs for T steps each.

    kwargs   : extra gym.make kwargs (action_space, size_reward, max_steps, ...)
    targets  : int array [E,9,11,11]
    starts   : list of E sparse lists [(x,y,z,id)] (may be [])
    actions  : walking: int array [E,T];
               flying: dict(movement f32[E,T,3], camera f32[E,T,2], inventory int[E,T], placement int[E,T])
    init_pose: optional [E,5] (x,y,z,yaw,pitch) applied through initialize_world
    Returns a dict of numpy arrays.
    """
    gym, Task, Tasks = load_reference()
    walkdict = isinstance(actions, dict) and 'buttons' in actions
    flying = isinstance(actions, dict) and not walkdict
    E = len(targets)
    T = (actions['buttons'] if walkdict else actions['inventory'] if flying else actions).shape[1]
    out = dict(
        agentPos=np.zeros((E, T, 5), np.float32), inventory=np.zeros((E, T, 6), np.float32),
        compass=np.zeros((E, T), np.float32), reward=np.zeros((E, T), np.float64),
        done=np.zeros((E, T), np.uint8), grid=np.zeros((E, T, 9, 11, 11), np.int8),
        internal=np.zeros((E, T, 8), np.float64), reset_before=np.zeros((E, T), np.uint8),
        reset_agentPos=np.zeros((E, 5), np.float32), reset_inventory=np.zeros((E, 6), np.float32),
        reset_compass=np.zeros((E,), np.float32), reset_grid=np.zeros((E, 9, 11, 11), np.int8),
        reset_internal=np.zeros((E, 8), np.float64),
        syn_max_int=np.zeros((E, T), np.int32), env_max_int=np.zeros((E,), np.int32),
    )
    for e in range(E):
        env = gym.make('IGLUGridworld-v0', vector_state=True, render=False, **kwargs)
        task = Task('', np.asarray(targets[e]).astype(np.int32), starting_grid=list(starts[e]),
                    **(task_kwargs or {}))
        env.set_task(task)
        if init_pose is not None:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                env.initialize_world(list(starts[e]), [float(v) for v in init_pose[e]])
        obs = env.reset()
        out['reset_agentPos'][e] = obs['agentPos']
        out['reset_inventory'][e] = obs['inventory']
        out['reset_compass'][e] = obs['compass'][0]
        out['reset_grid'][e] = obs['grid']
        out['reset_internal'][e] = _internals(env)
        out['env_max_int'][e] = env.unwrapped.max_int
        done = False
        for t in range(T):
            if done and reset_on_done:
                env.reset()
                out['reset_before'][e, t] = 1
            if walkdict:
                a = walking_dict_action(actions['buttons'][e, t], actions['camera'][e, t])
            elif flying:
                a = flying_action(actions['movement'][e, t], actions['camera'][e, t],
                                  actions['inventory'][e, t], actions['placement'][e, t])
            else:
                a = int(actions[e, t])
            obs, reward, done, _ = env.step(a)
            out['agentPos'][e, t] = obs['agentPos']
            out['inventory'][e, t] = obs['inventory']
            out['compass'][e, t] = obs['compass'][0]
            out['reward'][e, t] = float(reward)
            out['done'][e, t] = bool(done)
            out['grid'][e, t] = obs['grid']
            out['internal'][e, t] = _internals(env)
            out['syn_max_int'][e, t] = env.unwrapped._synthetic_task.max_int
    return out


def task_vectors(targets, grids, full_grids=None, invariant=True):
    """Pure Task vectors: per target the 4 admissible sets (as bbox + count) and, per
    (target, grid) pair, maximal_intersection / argmax_intersection (tasks/task.py:121-161)."""
    gym, Task, Tasks = load_reference()
    P, G = len(targets), len(grids)
    n_rot = 4 if invariant else 1
    adm_count = np.zeros((P, 4), np.int32)
    adm_mask = np.zeros((P, 4, 21, 21), np.uint8)
    target_size = np.zeros((P,), np.int32)
    max_int = np.zeros((P, G), np.int32)
    argmax = np.zeros((P, G, 3), np.int32)
    rot = np.zeros((P, 4, 9, 11, 11), np.int8)
    for p in range(P):
        fg = None if full_grids is None else np.asarray(full_grids[p]).astype(np.int32)
        task = Task('', np.asarray(targets[p]).astype(np.int32), starting_grid=[], full_grid=fg,
                    invariant=invariant)
        target_size[p] = task.target_size
        for i in range(n_rot):
            adm_count[p, i] = len(task.admissible[i])
            for dx, dz in task.admissible[i]:
                adm_mask[p, i, dx + 10, dz + 10] = 1
        for i in range(4):
            rot[p, i] = task.target_grids[i]
        for g in range(G):
            grid = np.asarray(grids[g]).astype(np.int32)
            max_int[p, g] = task.maximal_intersection(grid)
            argmax[p, g] = task.argmax_intersection(grid)
    return dict(adm_count=adm_count, adm_mask=adm_mask, target_size=target_size,
                max_int=max_int, argmax=argmax, rot=rot)


def load_cdm_goals():
    """skills/goals.pkl: {name: sparse [(x,y,z,id)]} -> {name: dense int8[9,11,11]}."""
    import pickle
    with open(os.path.join(REFERENCE_ROOT, 'skills', 'goals.pkl'), 'rb') as f:
        goals = pickle.load(f)
    out = {}
    for name, blocks in goals.items():
        d = np.zeros((9, 11, 11), np.int8)
        for x, y, z, c in blocks:
            d[int(y) + 1, int(x) + 5, int(z) + 5] = int(c)
        out[name] = d
    return out
