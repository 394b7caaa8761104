"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors recorded from the
Python reference and against the CPU oracle on the same seeded inputs.  Bit-exact everywhere
(integer outputs, float32 observations and the float64 agent internals)."""
import numpy as np
import pytest
import torch

import golden_replay as GR

pytestmark = pytest.mark.gpu


def _hip(fx, **kw):
    from hip_driver import HipDriver
    return HipDriver(fx, **kw)


@pytest.mark.parametrize('name', GR.WALK_FIXTURES)
def test_walking_fixture(name):
    fx = GR.load_fixture(name)
    assert GR.replay(fx, _hip(fx)) == fx['done'].size


@pytest.mark.parametrize('gs', [64, 32, 16, 8, 4, 2, 1])
@pytest.mark.parametrize('name', ['s2_walk_cdm', 's3_walk_rt20', 's5_scripted', 's5_init_pose', 's12_wide_inventory'])
def test_walking_fixture_lane_groups(name, gs):
    fx = GR.load_fixture(name)
    GR.replay(fx, _hip(fx, lanes_per_env=gs), max_steps=260)


def test_task_vectors():
    """igw_task_eval vs Task.maximal_intersection / argmax_intersection of the reference."""
    from gridworld_amd import task_eval
    z = np.load(GR.GOLDEN_DIR + '/s6_task_vectors.npz')
    targets, grids, fulls = z['targets'], z['grids'], z['full_grids']
    P, G = len(targets), len(grids)
    tt = np.repeat(targets, G, axis=0)
    gg = np.tile(grids, (P, 1, 1, 1))
    ff = np.repeat(fulls, G, axis=0)
    mi, am, ts = task_eval(tt, gg)
    assert np.array_equal(mi.reshape(P, G), z['max_int'])
    assert np.array_equal(am.reshape(P, G, 3), z['argmax'])
    assert np.array_equal(ts.reshape(P, G)[:, 0], z['target_size'])
    mi, am, _ = task_eval(tt, gg, invariant=False)
    assert np.array_equal(mi.reshape(P, G), z['ni_max_int'])
    assert np.array_equal(am.reshape(P, G, 3), z['ni_argmax'])
    mi, am, _ = task_eval(tt, gg, full_grids=ff)
    assert np.array_equal(mi.reshape(P, G), z['fg_max_int'])
    assert np.array_equal(am.reshape(P, G, 3), z['fg_argmax'])


@pytest.mark.parametrize('gs', [0, 64, 4, 1])
@pytest.mark.parametrize('name', GR.SUBTASK_FIXTURES)
def test_subtasks_env_fixture(name, gs):
    """env.set_task_generator(Subtasks(...)) episodes recorded from the reference, through
    set_tasks(targets, starts, full_grids=...): igw_prepare_tasks' full_grid -> env_max_int path
    (tasks/task.py:63-72, 260-286; env.py:227-241) and the SizeReward it feeds at the first step (env.py:325-331)."""
    from hip_driver import HipDriver
    fx = GR.load_subtasks_fixture(name)
    E, R = fx['ep_targets'].shape[:2]
    drv = HipDriver(dict(targets=fx['full_grids'], kwargs=fx['kwargs']), lanes_per_env=gs, num_tasks=E * R)
    assert GR.replay_subtasks(fx, drv) == fx['done'].size
    # every row of the table (also the ones no episode used), not only the rows the replay touched
    meta = drv.env.task_meta.cpu().numpy()
    got = meta[:, 42:44].copy().view(np.int16)[:, 0].reshape(E, R)
    used = np.arange(R)[None, :] < fx['n_episodes'][:, None]
    assert np.array_equal(got[used], fx['ep_env_max_int'][used])
    assert (fx['ep_env_max_int'][used] > 0).sum() > 20
    _check_index(drv.env, rows=np.nonzero(used.reshape(-1))[0][::7])   # synthetic targets with negative ids


def test_env_max_int_with_full_grid_equals_task_eval():
    """task_meta.env_max_int (prepare_tasks_kernel) == Task(target, full_grid).maximal_intersection(start)
    through the separately pinned igw_task_eval and through the oracle, for random rows incl. invariant=False."""
    from gridworld_amd import VecGridWorld, task_eval
    from oracle import oracle as O
    rng = np.random.RandomState(4242)
    n = 300
    fg = np.zeros((n, 9, 11, 11), np.int8)
    tg = np.zeros_like(fg)
    st = np.zeros_like(fg)
    for e in range(n):
        k = rng.randint(2, 30)
        cells = rng.permutation(1089)[:k]
        lv = rng.randint(1, 4)
        cells = (cells % (121 * lv))
        x0, z0 = rng.randint(0, 6), rng.randint(0, 6)   # a structure that does not fill the zone: translations exist
        for c in cells:
            y, r = divmod(int(c), 121)
            fg[e, y, x0 + (r // 11) % 6, z0 + (r % 11) % 6] = rng.randint(1, 7)
        nz = np.argwhere(fg[e] != 0)
        order = rng.permutation(len(nz))
        a, b = sorted(rng.randint(0, len(nz) + 1, size=2))
        for i in order[:b]:
            tg[e][tuple(nz[i])] = fg[e][tuple(nz[i])]
        for i in order[:a]:
            st[e][tuple(nz[i])] = fg[e][tuple(nz[i])]
        if rng.rand() < 0.3:      # the start was built somewhere else / rotated: max_int finds it
            st[e] = np.roll(np.rot90(st[e], k=rng.randint(4), axes=(1, 2)), (rng.randint(-1, 2), rng.randint(-1, 2)), axis=(1, 2))
    inv = (rng.rand(n) < 0.8).astype(np.uint8)
    env = VecGridWorld(n, size_reward=True)
    env.set_tasks(tg, st, full_grids=fg, invariant=inv)
    torch.cuda.synchronize()
    got = env.task_meta.cpu().numpy()[:, 42:44].copy().view(np.int16)[:, 0]
    for flag in (0, 1):
        m = inv == flag
        mi, _, _ = task_eval(tg[m], st[m], full_grids=fg[m], invariant=bool(flag))
        assert np.array_equal(got[m], mi)
    want = np.array([O.task_eval(tg[e], st[e], full_grid=fg[e], invariant=bool(inv[e]))['max_int'] for e in range(n)])
    assert np.array_equal(got, want) and (want > 0).sum() > 50


def test_env_max_int_at_reset():
    """GridWorld.max_int (env.py:241) lands in the task metadata and drives SizeReward."""
    from gridworld_amd import VecGridWorld
    fx = GR.load_fixture('s2_walk_cdm')
    env = VecGridWorld(len(fx['targets']), size_reward=True)
    env.set_tasks(fx['targets'], fx['starts'])
    torch.cuda.synchronize()
    meta = env.task_meta.cpu().numpy()
    got = meta[:, 42:44].copy().view(np.int16)[:, 0]
    assert np.array_equal(got, fx['env_max_int'])


def _rt20_targets(n, seed):
    rng = np.random.RandomState(seed)
    out = np.zeros((n, 9, 11, 11), np.int8)
    for e in range(n):
        bx, bz = rng.randint(2, 9, size=2)
        cells = [(bx + dx, bz + dz) for dx in range(-2, 3) for dz in range(-2, 3)]
        for i in rng.permutation(25)[:20]:
            out[e, 0, cells[i][0], cells[i][1]] = rng.randint(1, 7)
    return out


def _compare(env, ob, where):
    torch.cuda.synchronize()
    assert np.array_equal(env.done.cpu().numpy(), ob.done), where + ' done'
    assert np.array_equal(env.reward.cpu().numpy().view(np.uint32), ob.reward.view(np.uint32)), where + ' reward'
    assert np.array_equal(env.grid.cpu().numpy().reshape(env.num_envs, -1), ob.grid), where + ' grid'
    assert np.array_equal(env.inventory.cpu().numpy(), ob.inventory), where + ' inventory'
    assert np.array_equal(env.agent_pos.cpu().numpy().view(np.uint32), ob.agentPos.view(np.uint32)), where + ' agentPos'
    assert np.array_equal(env.compass.cpu().numpy().view(np.uint32), ob.compass.view(np.uint32)), where + ' compass'


@pytest.mark.parametrize('gs', [64, 16, 4, 1])
def test_config2_4096_envs_vs_oracle(gs):
    """BASELINE config 2: 4,096 parallel envs, walking, DUMMY-equivalent task, bit-exact vs the CPU
    path on counter-RNG action streams, auto-reset inside step (max_steps=50 so resets happen)."""
    from gridworld_amd import VecGridWorld
    from oracle import oracle as O
    n, T = 4096, 120
    kw = dict(size_reward=False, max_steps=50)
    tg = np.zeros((1, 9, 11, 11), np.int8)
    tg[0, 8, 10, 10] = 1
    env = VecGridWorld(n, autoreset=True, num_tasks=1, lanes_per_env=gs, **kw)
    env.set_tasks(tg, invariant=False)
    env.reset()
    ob = O.OracleBatch(n, **kw)
    ob.set_tasks(np.repeat(tg, n, axis=0), invariant=False)
    ob.reset()
    acts = env.fill_actions(T, seed=1234)
    a_np = acts.cpu().numpy()
    for t in range(T):
        env.step(acts[t])
        ob.step_walking(a_np[t], autoreset=True, nthreads=8)
        if t % 10 == 9 or t == T - 1:
            _compare(env, ob, f'step {t}')
    assert env.stats()['resets'] == n * (T // 50)


@pytest.mark.parametrize('gs', [0, 8, 2, 1])
def test_rt20_autoreset_vs_oracle(gs):
    """rt20 targets (full maximal_intersection reward), 1,000 envs (ragged last block), per-env tasks,
    vs the oracle; the occupancy bitmap must stay in sync with the int8 grid."""
    from gridworld_amd import VecGridWorld
    from oracle import oracle as O
    n, T = 1000, 300
    kw = dict(size_reward=False)
    tg = _rt20_targets(n, 5)
    env = VecGridWorld(n, autoreset=True, lanes_per_env=gs, **kw)
    env.set_tasks(tg)
    env.reset()
    ob = O.OracleBatch(n, **kw)
    ob.set_tasks(tg)
    ob.reset()
    acts = env.fill_actions(T, seed=99)
    a_np = acts.cpu().numpy()
    changed = 0
    for t in range(T):
        env.step(acts[t])
        ob.step_walking(a_np[t], autoreset=True, nthreads=8)
        if t % 25 == 24 or t == T - 1:
            _compare(env, ob, f'step {t}')
    st = env.stats()
    assert st['resets'] >= n and 0.01 < st['changed'] / (n * T) < 0.3
    _check_occ(env)
    _check_hist(env, tg, sample=range(0, n, 7))
    _check_index(env, rows=range(0, n, 11))


def _expected_hist(target_syn, grid_syn):
    """The vote histogram the kernels keep per env, from first principles: for every rotation
    (tasks/task.py:47-56) and every admissible translation (bounding-box rule == tasks/task.py:62-72)
    the intersection count of tasks/task.py:138-145, laid out [rot][dx - dxlo][dz - dzlo] with pitch 11."""
    out = np.zeros(512, np.uint16)
    t = target_syn.astype(np.int32)
    g = grid_syn.astype(np.int32)
    if not t.any():
        return out
    for r in range(4):
        tr = np.rot90(t, k=-r, axes=(1, 2))
        _, xs, zs = np.nonzero(tr)
        dxlo, dxhi, dzlo, dzhi = xs.max() - 10, xs.min(), zs.max() - 10, zs.min()
        for dx in range(dxlo, dxhi + 1):
            for dz in range(dzlo, dzhi + 1):
                st = tr[:, max(dx, 0):11 + min(dx, 0), max(dz, 0):11 + min(dz, 0)]
                sg = g[:, max(-dx, 0):11 + min(-dx, 0), max(-dz, 0):11 + min(-dz, 0)]
                out[r * 121 + (dx - dxlo) * 11 + (dz - dzlo)] = ((st == sg) & (st != 0)).sum()
    return out


def _check_hist(env, targets, starts=None, sample=None):
    """Persistent histogram == histogram recomputed from scratch; max_int == its maximum unless the
    reference itself holds a stale cached value (dirty bit)."""
    torch.cuda.synchronize()
    grid = env.grid.cpu().numpy().astype(np.int32)
    hist = env.hist_buf.cpu().numpy().view(np.uint16)
    ts = env.task_state()
    idx = range(env.num_envs) if sample is None else sample
    for e in idx:
        st = np.zeros((9, 11, 11), np.int32) if starts is None else starts[e].astype(np.int32)
        want = _expected_hist(targets[e].astype(np.int32) - st, grid[e] - st)
        assert np.array_equal(hist[e], want), f'env {e}: histogram differs from a fresh recount'
        if not ts['dirty'][e]:
            assert ts['max_int'][e] == want.max(), f'env {e}: max_int {ts["max_int"][e]} vs {want.max()}'


def _expected_index(target_syn, bbox16):
    """The colour index igw_prepare_tasks / the RandomTasks generator derive from a synthetic target (include/igw.h):
    per level the rotation boxes, the class offsets and the (x << 4 | z) cell list sorted by colour class."""
    out = np.zeros((9, 160), np.uint8)
    for y in range(9):
        out[y, :16] = bbox16
        lvl = target_syn[y].reshape(-1).astype(np.int32)
        pos = 0
        for k in range(14):
            colour = k - 7 if k < 7 else k - 6
            out[y, 16 + k] = pos
            for c in np.nonzero(lvl == colour)[0]:
                out[y, 32 + pos] = (c // 11) << 4 | (c % 11)
                pos += 1
        out[y, 16 + 14] = out[y, 16 + 15] = pos   # (offs[15] repeats offs[14], present only where the level has cells)
    return out.reshape(-1)


def _check_index(env, rows=None):
    """task_index == the index recomputed on the host from task_target and the metadata's boxes."""
    torch.cuda.synchronize()
    tt = env.task_target.cpu().numpy()[:, :1089].reshape(-1, 9, 11, 11)
    meta = env.task_meta.cpu().numpy()
    got = env.task_index.cpu().numpy()
    for r in (range(len(tt)) if rows is None else rows):
        assert np.array_equal(got[r], _expected_index(tt[r], meta[r, 48:64])), f'task row {r}: colour index'


def _check_occ(env):
    """HBM occupancy rows == the grid in the padded 9 x 13 x 13 bit layout of include/igw.h."""
    torch.cuda.synchronize()
    n = env.num_envs
    g = env.grid.cpu().numpy().reshape(n, 9, 11, 11) != 0
    box = np.zeros((n, 9, 13, 13), bool)
    box[:, :, 1:12, 1:12] = g
    bits = np.zeros((n, 48 * 32), bool)
    bits[:, :9 * 169] = box.reshape(n, -1)
    want = np.packbits(bits.reshape(n, 48, 32), axis=-1, bitorder='little').view(np.uint32)[:, :, 0]
    got = env.occ_buf.cpu().numpy().view(np.uint32)
    assert got.shape == want.shape and np.array_equal(got, want), 'occupancy bitmap out of sync with the grid'


@pytest.mark.parametrize('gs', [0, 16, 1])
def test_fused_rollout_vs_oracle(gs):
    """igw_rollout_walking (T steps in one launch, counter RNG, auto-reset) == T single steps == oracle."""
    from gridworld_amd import VecGridWorld
    from oracle import oracle as O
    n, T = 500, 300
    kw = dict(size_reward=False)
    tg = _rt20_targets(n, 11)
    env = VecGridWorld(n, autoreset=True, lanes_per_env=gs, **kw)
    env.set_tasks(tg)
    env.reset()
    env.rollout(T, seed=4242)
    ob = O.OracleBatch(n, **kw)
    ob.set_tasks(tg)
    ob.reset()
    steps, changed = ob.rollout_walking(T, 4242, autoreset=True, nthreads=8)
    torch.cuda.synchronize()
    assert np.array_equal(env.grid.cpu().numpy().reshape(n, -1), np.stack([e.obs()['grid'].reshape(-1) for e in ob.envs]).astype(np.int8))
    assert np.array_equal(env.internals().view(np.uint64), ob.internals().view(np.uint64))
    st = env.stats()
    assert st['rollout_steps'] == steps == n * T and st['changed'] == changed
    # and the same thing as T separate launches
    env2 = VecGridWorld(n, autoreset=True, **kw)
    env2.set_tasks(tg)
    env2.reset()
    acts = env2.fill_actions(T, seed=4242)
    for t in range(T):
        env2.step(acts[t])
    torch.cuda.synchronize()
    assert torch.equal(env.grid_buf, env2.grid_buf) and torch.equal(env.agent_buf, env2.agent_buf)
    assert torch.equal(env.occ_buf, env2.occ_buf)
    _check_occ(env)


@pytest.mark.parametrize('gs', [0, 4, 1])
def test_start_grids_incremental_reward_vs_oracle(gs):
    """CDM structures with partial starting grids (negative synthetic ids, blocks to remove): exercises
    histogram decrements, rescans and the reference's stale-cache case (a change that leaves the block
    count unchanged).  Checked against the oracle every few steps and against a fresh recount."""
    from gridworld_amd import VecGridWorld
    from oracle import oracle as O
    fx = GR.load_fixture('s2_walk_cdm')
    reps = 24
    tg = np.tile(fx['targets'], (reps, 1, 1, 1))
    st = np.tile(fx['starts'], (reps, 1, 1, 1))
    n, T = len(tg), 400
    kw = dict(size_reward=False, max_steps=1000)
    env = VecGridWorld(n, autoreset=False, lanes_per_env=gs, **kw)
    env.set_tasks(tg, st)
    env.reset()
    ob = O.OracleBatch(n, **kw)
    ob.set_tasks(tg, st)
    ob.reset()
    rng = np.random.RandomState(77)
    # a policy that looks down and places / breaks a lot, so start blocks get broken and replaced
    acts = rng.choice([1, 2, 3, 4, 5, 7, 9, 12, 13, 14, 15, 16, 16, 16, 17, 17], size=(T, n)).astype(np.int32)
    acts[:6] = 14
    for t in range(T):
        env.step(torch.as_tensor(acts[t]))
        ob.step_walking(acts[t], nthreads=8)
        if t % 20 == 19 or t == T - 1:
            _compare(env, ob, f'step {t}')
    stt = env.stats()
    ts = env.task_state()
    print('changed', stt['changed'], 'rescans', stt['rescans'], 'dirty now', int(ts['dirty'].sum()))
    assert stt['rescans'] > 0   # histogram row updates
    _check_hist(env, tg, st, sample=range(0, n, 5))
    _check_occ(env)


@pytest.mark.parametrize('gs', [4, 8, 64, 1])
def test_autoreset_restores_starting_grids_desynchronised(gs):
    """Auto-reset at max_steps with partial STARTING grids (CDM structures), episodes de-synchronised by masked resets
    in the first steps: a wavefront then has one env whose episode runs out (its starting row is staged during the step),
    several (the first staged, the others restored at the end), or one in a step where many of its envs changed a block
    (no scratch slot left to stage into).  Oracle at every step; histogram and occupancy recounted at the end."""
    from gridworld_amd import VecGridWorld
    from oracle import oracle as O
    fx = GR.load_fixture('s2_walk_cdm')
    reps = 41
    tg = np.tile(fx['targets'], (reps, 1, 1, 1))[:333]
    st = np.tile(fx['starts'], (reps, 1, 1, 1))[:333]
    n, T = len(tg), 70
    kw = dict(size_reward=False, max_steps=9)
    env = VecGridWorld(n, autoreset=True, lanes_per_env=gs, **kw)
    env.set_tasks(tg, st)
    env.reset()
    ob = O.OracleBatch(n, **kw)
    ob.set_tasks(tg, st)
    ob.reset()
    rng = np.random.RandomState(4)
    acts = rng.choice([1, 3, 5, 7, 9, 12, 14, 14, 15, 16, 16, 16, 17, 17, 17], size=(T, n)).astype(np.int32)
    acts[40:44] = np.array([14, 14, 17, 16], np.int32)[:, None]   # every env looks down, places, breaks: all change at once
    idx = np.arange(n)
    for t in range(T):
        if t < 8:   # shifts the episode phase of one env in eight
            m = (idx % 8 == t).astype(np.uint8)
            env.reset(mask=torch.as_tensor(m))
            ob.reset(mask=m)
        env.step(torch.as_tensor(acts[t]))
        ob.step_walking(acts[t], autoreset=True, nthreads=8)
        _compare(env, ob, f'step {t}')
    assert env.stats()['resets'] >= n * (T // 9 - 1)
    _check_hist(env, tg, st, sample=range(0, n, 4))
    _check_occ(env)


@pytest.mark.parametrize('gs', [0, 16, 1])
def test_single_colour_floors_vs_oracle(gs):
    """Targets whose levels are full floors of one colour: one placed block matches up to 121 target cells at once
    (the vote list of the histogram update overflows and is flushed), several envs of a wave change in the
    same step, and translations / rotations are all admissible or all cut.  Oracle + fresh recount."""
    from gridworld_amd import VecGridWorld
    from oracle import oracle as O
    n, T = 384, 260
    tg = np.zeros((n, 9, 11, 11), np.int8)
    st = np.zeros_like(tg)
    rng = np.random.RandomState(12)
    for e in range(n):
        c = 1 + e % 3
        tg[e, 0] = c                                 # a full floor
        if e % 4 == 1:
            tg[e, 1, 2:9, 2:9] = c                   # plus a smaller second storey of the same colour
        if e % 4 == 2:
            tg[e, 0, rng.randint(11), :] = 0         # a missing row: not every translation is admissible
        if e % 5 == 3:
            st[e, 0, 4:7, 4:7] = c                   # part of it already stands
    kw = dict(size_reward=False, max_steps=1000)
    env = VecGridWorld(n, autoreset=False, lanes_per_env=gs, **kw)
    env.set_tasks(tg, st)
    env.reset()
    ob = O.OracleBatch(n, **kw)
    ob.set_tasks(tg, st)
    ob.reset()
    # look down, then place / break a lot with the three colours in play (hotbar 1..3 = actions 6..8)
    acts = rng.choice([1, 2, 3, 4, 6, 6, 7, 7, 8, 8, 12, 13, 14, 15, 16, 16, 17], size=(T, n)).astype(np.int32)
    acts[:8] = 14
    for t in range(T):
        env.step(torch.as_tensor(acts[t]))
        ob.step_walking(acts[t], nthreads=8)
        if t % 20 == 19 or t == T - 1:
            _compare(env, ob, f'step {t}')
    assert env.stats()['changed'] > 20 * n // 10
    _check_hist(env, tg, st, sample=range(0, n, 3))
    _check_occ(env)


@pytest.mark.parametrize('gs', [0, 8, 4, 2, 1])
def test_every_env_of_a_wave_changes_in_the_same_step(gs):
    """All envs get the same scripted action, so every env of a wave places / breaks in the same step: the
    histogram update runs out of scratch rows (first chunk), uses the rows prefetched into the dead occupancy
    rows (second chunk) and then the fetch-now path (later chunks).  Per-env rt20 targets; oracle at every
    step, then a fresh recount of the persistent histogram and the occupancy bitmap."""
    from gridworld_amd import VecGridWorld
    from oracle import oracle as O
    n = 200  # ragged: the last wave is partly filled
    tg = _rt20_targets(n, 21)
    kw = dict(size_reward=False, max_steps=1000)
    env = VecGridWorld(n, autoreset=False, lanes_per_env=gs, **kw)
    env.set_tasks(tg)
    env.reset()
    ob = O.OracleBatch(n, **kw)
    ob.set_tasks(tg)
    ob.reset()
    script = [14] * 8
    for rep in range(12):
        script += [17, 6 + rep % 6, 17, 1 + rep % 4, 17, 16, 12, 17, 16, 16, 5, 17, 3, 17, 13, 16]
    changes = 0
    for t, a in enumerate(script):
        acts = np.full(n, a, np.int32)
        before = env.stats()['rescans']
        env.step(torch.as_tensor(acts))
        ob.step_walking(acts, nthreads=8)
        _compare(env, ob, f'step {t} (action {a})')
        changes += env.stats()['rescans'] - before == n
    assert changes >= 20, 'the script should make every env change its grid in the same step many times'
    _check_hist(env, tg, sample=range(0, n, 2))
    _check_occ(env)


@pytest.mark.parametrize('autoreset', [True, False])
@pytest.mark.parametrize('gs', [0, 4, 1])
def test_rollout_over_recorded_actions_equals_stepping(gs, autoreset):
    """igw_rollout_walking_actions: T fused steps over the caller's actions == T calls of step(), every step's reward
    and done included, with and without auto-reset (without it the envs keep stepping past done, as the
    reference does)."""
    from gridworld_amd import VecGridWorld
    n, T = 600, 330
    tg = _rt20_targets(n, 31)
    kw = dict(size_reward=False, max_steps=120, autoreset=autoreset, lanes_per_env=gs)
    rng = np.random.RandomState(5)
    acts = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 12, 13, 14, 14, 15, 16, 16, 17, 17, 17, 0], size=(T, n)).astype(np.int32)
    a = VecGridWorld(n, **kw)
    a.set_tasks(tg)
    a.reset()
    b = VecGridWorld(n, **kw)
    b.set_tasks(tg)
    b.reset()
    acts_d = torch.as_tensor(acts, device=a.device)
    rw, dn = a.rollout_actions(acts_d, return_rewards=True)
    rw_b, dn_b = [], []
    for t in range(T):
        b.step(acts_d[t])
        rw_b.append(b.reward.clone())
        dn_b.append(b.done.clone())
    torch.cuda.synchronize()
    assert torch.equal(rw.view(torch.int32), torch.stack(rw_b).view(torch.int32))
    assert torch.equal(dn, torch.stack(dn_b))
    for name in ('grid_buf', 'occ_buf', 'hist_buf', 'agent_buf', 'agent_pos', 'inventory', 'compass', 'reward', 'done'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    sa, sb = a.stats(), b.stats()
    assert sa['changed'] == sb['changed'] and sa['resets'] == sb['resets'] and sa['rescans'] == sb['rescans']
    assert (sa['resets'] > 0) == autoreset


def test_product_does_not_import_oracle():
    import sys
    import gridworld_amd  # noqa: F401
    import inspect
    import gridworld_amd.vec_env as v
    import gridworld_amd._lib as l
    for mod in (v, l, gridworld_amd):
        assert 'oracle' not in inspect.getsource(mod).replace('no CPU fallback', '')
