"""Randomised lock-step scenarios of the gym-shaped env API against the CPU oracle (test infrastructure).

One scenario generator, two backends:

  * the imported Python REFERENCE (tests/golden/fuzz_ref_vs_oracle.py, build container only): pins the oracle;
  * the PRODUCT's 1-env facade, gridworld_amd.make(...) on the GPU (tests/test_gpu_facade_fuzz.py): the same seeds
    draw the same scenarios -- create_env kwargs, task sources (Task / Subtasks / CustomTasks / RandomTasks with seeded
    np.random, which the product's generators consume exactly like the reference's), starting grids, full_grid,
    initialize_world poses, all three action spaces, action mixes -- so what both sides agree with the oracle on, they
    agree on with each other.

Every step compares float32 observations (bit patterns), inventory, grid, reward (the Python value), done, the
float64 internals (bit patterns), the synthetic task's max_int / prev_grid_size, GridWorld.max_int and step_no; every
reset its observation and internals.

A backend provides: name; make(kwargs) -> env; Task / Subtasks / CustomTasks / RandomTasks; goals() (dense CDM
structures, sorted C1..C157); internals(env) -> 8 floats; syn(env) -> (max_int, prev_grid_size) of the synthetic task;
trig(crlibm) -> context manager around the scenario; oracle_device_trig(crlibm) -> bool; float32_actions (the env
takes continuous action values as float32)."""
import contextlib
import time
import warnings

import numpy as np


def dense_to_sparse(dense):
    """dense [9,11,11] -> reference-style sparse list [(x, y, z, id)] (tasks/task.py:178-187)."""
    out = []
    ys, xs, zs = np.nonzero(dense)
    for y, x, z in zip(ys, xs, zs):
        out.append((int(x) - 5, int(y) - 1, int(z) - 5, int(dense[y, x, z])))
    return out


# ------------------------------------------------------------------------------------------------ scenario pieces
def rt20(rng):
    g = np.zeros((9, 11, 11), np.int8)
    bx, bz = rng.randint(2, 9, size=2)
    cells = [(bx + dx, bz + dz) for dx in range(-2, 3) for dz in range(-2, 3)]
    for i in rng.permutation(25)[:rng.randint(1, 21)]:
        g[0, cells[i][0], cells[i][1]] = rng.randint(1, 7)
    return g


def scattered(rng, n, levels=9):
    g = np.zeros((levels * 121,), np.int8)
    g[rng.permutation(g.size)[:n]] = rng.randint(1, 7, size=n)
    out = np.zeros((9, 11, 11), np.int8)
    out.reshape(-1)[:g.size] = g
    return out


def draw_target(rng, goals):
    k = rng.randint(8)
    if k == 0:
        return rt20(rng)
    if k == 1:
        return scattered(rng, rng.randint(1, 41))
    if k in (2, 3):
        return goals[rng.randint(len(goals))].copy()
    if k == 4:      # a few low levels, denser (towers, walls)
        return scattered(rng, rng.randint(5, 80), levels=rng.randint(1, 4))
    if k == 5:      # DUMMY_TASK's target (tasks/task_set.py:160)
        g = np.zeros((9, 11, 11), np.int8)
        g[8, 10, 10] = 1
        return g
    if k == 6:      # one or two blocks near the spawn point: completions happen
        g = np.zeros((9, 11, 11), np.int8)
        for _ in range(rng.randint(1, 3)):
            g[0, rng.randint(4, 7), rng.randint(3, 6)] = rng.randint(1, 7)
        return g
    return np.zeros((9, 11, 11), np.int8) if rng.rand() < 0.3 else rt20(rng)


def draw_start(rng, target):
    """sparse starting grid: [], a subset of the target, foreign blocks, or a wide one-colour floor."""
    k = rng.randint(6)
    sp = dense_to_sparse(target)
    if k <= 1 or (k == 2 and not sp):
        return []
    out = []
    if k in (2, 3):
        m = rng.randint(0, len(sp) + 1)
        out = [sp[i] for i in rng.permutation(len(sp))[:m]]
    if k in (3, 4):
        for _ in range(rng.randint(1, 6)):
            x, y, z = int(rng.randint(-5, 6)), int(rng.randint(-1, 4)), int(rng.randint(-5, 6))
            if not any(b[:3] == (x, y, z) for b in out):
                c = int(rng.randint(1, 7))
                if target[y + 1, x + 5, z + 5] == c and rng.rand() < 0.5:
                    continue
                out.append((x, y, z, c))
    if k == 5:      # many blocks of one colour -> negative inventory (env.py:243-246)
        c = int(rng.randint(1, 7))
        n = int(rng.randint(15, 60))
        cells = [(x, -1, z) for x in range(-5, 6) for z in range(-5, 6)]
        out = [(*cells[i], c) for i in rng.permutation(121)[:n]]
    return out


def draw_pose(rng):
    q = rng.rand() < 0.4
    x, z = rng.uniform(-9, 9, size=2)
    y = rng.uniform(-0.25, 8.5)
    yaw = rng.uniform(0, 360)
    pitch = rng.uniform(-90, 90)
    if q:
        x, y, z = round(x * 4) / 4, round(y * 4) / 4 - 0.25, round(z * 4) / 4
        yaw, pitch = float(5 * int(yaw / 5)), float(5 * int(pitch / 5))
    return [float(x), float(y), float(z), float(yaw), float(pitch)]


def structure_seq(rng, dense, n_turns):
    blocks = dense_to_sparse(dense)
    if not blocks:
        blocks = [(0, -1, 0, 1)]
    order = sorted(range(len(blocks)), key=lambda i: (blocks[i][1], rng.rand()))
    blocks = [blocks[i] for i in order]
    n_turns = max(1, min(n_turns, len(blocks)))
    cuts = sorted(set(int(round(len(blocks) * (k + 1) / n_turns)) for k in range(n_turns)))
    return [blocks[:c] for c in cuts if c > 0]


WALK_BUILDER_P = np.array([2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 2, 2, 4, 1, 3, 8], np.float64)


def draw_kwargs(rng):
    space = ('walking', 'walking_dict', 'flying')[rng.choice(3, p=[0.4, 0.25, 0.35])]
    kw = dict(size_reward=bool(rng.rand() < 0.5), select_and_place=bool(rng.rand() < 0.7),
              right_placement_scale=[1, 2., 0.5, 1.5][rng.randint(4)],
              wrong_placement_scale=[0.1, 1., 0.25, 0.][rng.randint(4)],
              max_steps=int([7, 30, 100, 250, 1000][rng.randint(5)]))
    if space == 'flying':
        kw['action_space'] = 'flying'
    elif space == 'walking_dict':
        kw['discretize'] = False
    return space, kw


class ActionSource:
    """One scenario's action stream (as the reference takes it) plus the same action for the oracle."""

    def __init__(self, rng, space, float32_only=False):
        self.rng, self.space = rng, space
        self.builder = rng.rand() < 0.5
        self.raw_f64 = rng.rand() < 0.2       # continuous values as raw doubles instead of widened float32
        if float32_only:                      # (drawn all the same: the scenario streams of the backends stay aligned)
            self.raw_f64 = False
        self.zero_p = rng.choice([0.0, 0.1, 0.4])

    def _cont(self, lo, hi, n):
        v = self.rng.uniform(lo, hi, size=n)
        if not self.raw_f64:
            v = v.astype(np.float32).astype(np.float64)
        z = self.rng.rand(n) < self.zero_p
        v[z] = 0.0
        if self.rng.rand() < 0.05:
            v[self.rng.randint(n)] = [lo, hi][self.rng.randint(2)]
        return [float(x) for x in v]

    def draw(self):
        rng = self.rng
        if self.space == 'walking':
            a = int(rng.choice(18, p=WALK_BUILDER_P / WALK_BUILDER_P.sum())) if self.builder else int(rng.randint(18))
            return a, a
        if self.space == 'walking_dict':
            p = 0.35 if self.builder else 0.2
            b = [int(v) for v in (rng.rand(7) < p)]
            hot = int(rng.randint(0, 7)) if rng.rand() < 0.3 else 0
            cam = self._cont(-5, 5, 2)
            if self.builder and rng.rand() < 0.3:
                cam[1] = -abs(cam[1])   # pitch up is negative-looking-down? keep variety either way
            ref = {'forward': b[0], 'back': b[1], 'left': b[2], 'right': b[3], 'jump': b[4], 'attack': b[5],
                   'use': b[6], 'hotbar': hot, 'camera': cam}
            return ref, {'buttons': b + [hot], 'camera': cam}
        mv = self._cont(-1, 1, 3)
        cam = self._cont(-5, 5, 2)
        inv = int(rng.randint(7))
        pl = int(rng.choice(3, p=[0.2, 0.5, 0.3])) if self.builder else int(rng.randint(3))
        a = {'movement': mv, 'camera': cam, 'inventory': inv, 'placement': pl}
        return a, a


# ------------------------------------------------------------------------------------------------ one scenario
FIELDS = ('agentPos', 'compass', 'inventory', 'grid', 'reward', 'done', 'internal', 'syn_max_int',
          'syn_prev_size', 'env_max_int', 'step_no')


def _b32(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _b64(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


class _Mismatch(Exception):
    pass


def _push_task(env, orc):
    """Hands the task the env currently holds (and its effective starting grid) to the oracle."""
    u = env.unwrapped
    task = u._task
    start = u.starting_grid if u.starting_grid is not None else []
    full = getattr(task, 'full_grid', None)
    invariant = getattr(task, 'invariant', None)
    if invariant is None:
        invariant = len(task.admissible) == 4                  # tasks/task.py:58-59: [[(0, 0)]] otherwise
    orc.set_task(np.asarray(task.target_grid), [tuple(int(v) for v in b) for b in start],
                 None if full is None else np.asarray(full), invariant=bool(invariant))


def run_scenario(seed, backend, max_T=None):
    """Returns dict(steps, resets, mismatch=None | {...}, tags...)."""
    from oracle import oracle as O
    B = backend
    rng = np.random.RandomState(seed)
    np.random.seed((seed * 7919 + 13) % (2 ** 32))            # the task generators use the global stream
    space, kw = draw_kwargs(rng)
    continuous = space != 'walking'
    crlibm = bool(rng.rand() < (0.35 if continuous else 0.1))
    source = ('task', 'subtasks', 'custom', 'random')[rng.choice(4, p=[0.55, 0.15, 0.15, 0.15])]
    T = int(rng.randint(60, 400))
    if crlibm:
        T = min(T, 200)                                        # mpmath trig is slow
    if max_T:
        T = min(T, max_T)
    info = dict(seed=int(seed), space=space, source=source, crlibm=crlibm, kwargs=kw, T=T, backend=B.name)
    orc = O.OracleEnv(**kw)
    O.use_device_trig(B.oracle_device_trig(crlibm))
    goals = B.goals()
    try:
        with B.trig(crlibm), warnings.catch_warnings():
            warnings.simplefilter('ignore')
            env = B.make(kw)
            pose = None
            if source == 'task':
                target = draw_target(rng, goals)
                start = draw_start(rng, target)
                tkw = {}
                if rng.rand() < 0.2:
                    full = target.copy()
                    extra = rng.permutation(1089)[:rng.randint(1, 8)]
                    f = full.reshape(-1)
                    f[extra] = np.where(f[extra] == 0, rng.randint(1, 7, size=len(extra)), f[extra])
                    tkw['full_grid'] = full.astype(np.int32)
                if rng.rand() < 0.2:
                    tkw['invariant'] = False
                env.set_task(B.Task('chat', target.astype(np.int32), starting_grid=start, **tkw))
            elif source == 'subtasks':
                seq = structure_seq(rng, draw_target(rng, goals), int(rng.randint(1, 6)))
                dialog = [['<A> turn %d' % i, '<B> ok'] for i in range(len(seq))]
                env.set_task_generator(B.Subtasks(dialog, seq))
            elif source == 'custom':
                n = int(rng.randint(1, 5))
                tg = [draw_target(rng, goals) for _ in range(n)]
                goals_ = [('c%d' % i, dense_to_sparse(g) if (rng.rand() < 0.3 and g.any()) else g.astype(np.int32))
                          for i, g in enumerate(tg)]
                tkw = {'starting_grid': draw_start(rng, tg[0])}
                if rng.rand() < 0.3:
                    tkw['invariant'] = False
                env.set_task_generator(B.CustomTasks(goals_, task_kwargs=tkw))
            else:
                d = int(rng.randint(1, 4))
                gen = B.RandomTasks(max_blocks=int(rng.randint(1, min(6, (d + 1) ** 2) + 1)),
                                    height_levels=int(rng.randint(1, 4)), max_dist=d,
                                    num_colors=int(rng.randint(1, 7)), max_cache=int(rng.choice([0, 0, 3])))
                env.set_task_generator(gen)
                pose = [0., 0., 0., 0., 0.] if rng.rand() < 0.5 else draw_pose(rng)
            if pose is None and rng.rand() < 0.25:
                pose = draw_pose(rng)
            if pose is not None:
                # RandomTasks' tasks have starting_grid None (the reference's step() would raise, env.py:290): they need
                # the overwrite; for the others it replaces the task's own starting grid
                u = env.unwrapped
                base = u._task.target_grid if source != 'random' else np.zeros((9, 11, 11), np.int8)
                ow = draw_start(rng, np.asarray(base)) if (source == 'random' or rng.rand() < 0.5) \
                    else list(u.starting_grid or [])
                env.initialize_world(ow, pose)
                orc.set_initial_pose(pose)
            info['pose'] = pose
            acts = ActionSource(rng, space, float32_only=B.float32_actions)
            steps = resets = 0

            def check(field, ok, t, detail):
                if not ok:
                    raise _Mismatch(dict(field=field, step=t, detail=detail))

            def do_reset(t):
                nonlocal resets
                obs = env.reset()
                _push_task(env, orc)
                o = orc.reset()
                resets += 1
                check('reset.inventory', np.array_equal(obs['inventory'], o['inventory']), t,
                      [obs['inventory'].tolist(), o['inventory'].tolist()])
                check('reset.grid', np.array_equal(obs['grid'], o['grid']), t, 'grid')
                check('reset.agentPos', np.array_equal(_b32(obs['agentPos']), _b32(o['agentPos'])), t, 'agentPos')
                check('reset.compass', np.array_equal(_b32(obs['compass']), _b32(o['compass'])), t, 'compass')
                if getattr(B, 'strict_float64', True):   # (agent.dy, time_int_steps survive a reset: SURVEY F7)
                    check('reset.internal', np.array_equal(_b64(B.internals(env)), _b64(orc.internal())), t,
                          [B.internals(env), orc.internal().tolist()])
                st = orc.task_state()
                check('reset.env_max_int', int(env.unwrapped.max_int) == st['env_max_int'], t,
                      [int(env.unwrapped.max_int), st['env_max_int']])
                check('reset.syn_max_int', int(B.syn(env)[0]) == st['syn_max_int'], t, '')

            try:
                do_reset(0)
                done = False
                for t in range(T):
                    if done:
                        do_reset(t)
                    ra, oa = acts.draw()
                    obs, reward, done, _ = env.step(ra)
                    o, orew, odone, _ = orc.step(oa)
                    steps += 1
                    u = env.unwrapped
                    check('done', bool(done) == bool(odone), t, [bool(done), bool(odone)])
                    check('reward', float(reward) == float(orew), t, [float(reward), float(orew)])
                    check('grid', np.array_equal(obs['grid'], o['grid']), t, 'grid')
                    check('inventory', np.array_equal(obs['inventory'], o['inventory']), t,
                          [obs['inventory'].tolist(), o['inventory'].tolist()])
                    check('agentPos', np.array_equal(_b32(obs['agentPos']), _b32(o['agentPos'])), t,
                          [obs['agentPos'].tolist(), o['agentPos'].tolist()])
                    check('compass', np.array_equal(_b32(obs['compass']), _b32(o['compass'])), t,
                          [obs['compass'].tolist(), o['compass'].tolist()])
                    ri, oi = B.internals(env), orc.internal()
                    if getattr(B, 'strict_float64', True):
                        check('internal', np.array_equal(_b64(ri), _b64(oi)), t, [ri, oi.tolist()])
                    else:   # the A-fly contract (glibc on the checker's side): float64 internals may part in the last bits
                        dev = float(np.abs(np.asarray(ri, np.float64) - np.asarray(oi, np.float64)).max())
                        check('internal (|difference| < 1e-9)', dev < 1e-9, t, [ri, oi.tolist()])
                        if not np.array_equal(_b64(ri), _b64(oi)):
                            info['float64_diff_steps'] = info.get('float64_diff_steps', 0) + 1
                            info['float64_max_dev'] = max(info.get('float64_max_dev', 0.0), dev)
                    st = orc.task_state()
                    smi, sps = B.syn(env)
                    check('syn_max_int', int(smi) == st['syn_max_int'], t, [int(smi), st['syn_max_int']])
                    check('syn_prev_size', int(sps) == st['syn_prev_size'], t, [int(sps), st['syn_prev_size']])
                    check('env_max_int', int(u.max_int) == st['env_max_int'], t, [int(u.max_int), st['env_max_int']])
                    check('step_no', int(u.step_no) == st['step_no'], t, [int(u.step_no), st['step_no']])
                info['mismatch'] = None
            except _Mismatch as m:
                info['mismatch'] = m.args[0]
            info.update(steps=steps, resets=resets)
    finally:
        O.use_device_trig(False)
    return info


def run_scenario_safe(seed, backend, max_T=None):
    try:
        return run_scenario(seed, backend, max_T)
    except Exception:  # a crash of either side is a finding, not a reason to lose the run
        import traceback
        return dict(seed=int(seed), steps=0, resets=0, mismatch=dict(field='exception', step=-1,
                                                                     detail=traceback.format_exc()[-1500:]),
                    space='?', source='?', crlibm=False, backend=backend.name)


def summarize(results, wall):
    by = {}
    for r in results:
        for key in ('space:' + r['space'], 'source:' + r['source'], 'trig:' + ('cr_libm' if r['crlibm'] else 'glibc')):
            d = by.setdefault(key, dict(scenarios=0, steps=0, mismatches=0))
            d['scenarios'] += 1
            d['steps'] += r['steps']
            d['mismatches'] += r['mismatch'] is not None
    bad = [r for r in results if r['mismatch'] is not None]
    extra = {}
    if any('float64_diff_steps' in r for r in results):
        extra = dict(scenarios_with_a_float64_difference=sum(1 for r in results if r.get('float64_diff_steps')),
                     steps_with_a_float64_difference=int(sum(r.get('float64_diff_steps', 0) for r in results)),
                     max_abs_float64_deviation=max(r.get('float64_max_dev', 0.0) for r in results))
    return dict(**extra, scenarios=len(results), env_steps=int(sum(r['steps'] for r in results)),
                reference_env_steps=int(sum(r['steps'] for r in results)),
                resets=int(sum(r['resets'] for r in results)), mismatches=len(bad), by=by,
                first_mismatches=bad[:10], wall_s=round(wall, 1),
                compared_every_step=list(FIELDS))


class ProductBackend:
    """gridworld_amd.make(...): the 1-env gym facade on the GPU.  Its trig is the product's own (correctly rounded)
    in every mode, so the oracle always runs the host compile of the same header; continuous action values enter
    the kernels as float32."""
    name = 'product'
    float32_actions = True

    def __init__(self):
        import os
        import gridworld_amd as G
        self.G = G
        self.Task, self.Subtasks, self.CustomTasks, self.RandomTasks = G.Task, G.Subtasks, G.CustomTasks, G.RandomTasks
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cdm_goals.npz'))
        self._goals = [g for g in z['dense']]        # sorted C1..C157 by tests/golden/gen_cdm_goals.py

    def make(self, kw):
        return self.G.make('IGLUGridworld-v0', vector_state=True, render=False, **kw)

    def goals(self):
        return self._goals

    def trig(self, crlibm):
        return contextlib.nullcontext()

    def oracle_device_trig(self, crlibm):
        return True

    def internals(self, env):
        a = env.unwrapped.agent
        return [*a.position, *a.rotation, a.dy, float(a.time_int_steps), float(a.active_block)]

    def syn(self, env):
        return env.unwrapped._counters


class ProductBackendGlibc(ProductBackend):
    """The same facade against the oracle computing with GLIBC sin / cos / atan2 (what the Python reference calls):
    nothing of the product's trig on the checker's side.  The A-fly contract applies (tests/test_gpu_flying.py): every
    integer output and every float32 observation bit-exact on every step; the float64 internals may part in the last
    bits where glibc misrounds (counted and bounded, not asserted equal)."""
    name = 'product-vs-glibc-oracle'
    strict_float64 = False

    def oracle_device_trig(self, crlibm):
        return False


def run(seeds, backend, max_T=None):
    t0 = time.time()
    results = [run_scenario_safe(int(s), backend, max_T) for s in seeds]
    return summarize(results, time.time() - t0)


if __name__ == '__main__':   # product side on the GPU box: python tests/scenario_fuzz.py [scenarios] [seed0] [out.json]
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    glibc = len(sys.argv) > 4 and sys.argv[4] == 'glibc'     # 4th argument: the oracle computes with GLIBC trig (A-fly contract)
    s = run(range(seed0, seed0 + n), ProductBackendGlibc() if glibc else ProductBackend())
    s.update(seed0=seed0, backend='product: gridworld_amd.make(...) on the GPU vs oracle.OracleEnv (%s)' % ('glibc trig; integers + float32 bit-exact, float64 within last bits' if glibc else 'device-trig mode'),
             command=' '.join(sys.argv))
    s.pop('reference_env_steps', None)
    if len(sys.argv) > 3:
        with open(sys.argv[3], 'w') as f:
            json.dump(s, f, indent=1)
    print(json.dumps({k: v for k, v in s.items() if k not in ('by', 'first_mismatches')}))
    sys.exit(1 if s['mismatches'] else 0)
