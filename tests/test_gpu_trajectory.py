"""Episode log on the device (igw_set_trajectory_log) + EpisodeLogger npz dumps: what the reference's Logged
wrapper collects (gridworld/wrappers.py:89-121, no video).  A fixture replay is logged and every dumped episode
must equal the fixture's arrays."""
import numpy as np
import pytest
import torch

import golden_replay as GR
from hip_driver import HipDriver

pytestmark = pytest.mark.gpu


def _expected_episodes(fx, T):
    """Splits the fixture's [E, T] arrays into episodes at its reset points; grids rebuilt from the change log."""
    E = fx['done'].shape[0]
    starts = fx['starts'].reshape(E, -1).astype(np.int32)
    eps = []
    for e in range(E):
        cur = None
        for t in range(T):
            if cur is None or fx['reset_before'][e, t]:
                cur = dict(env=e, agentPos=[np.zeros(5, np.float32)], inventory=[fx['reset_inventory'][e]],
                           compass=[np.zeros(1, np.float32)], grid=[starts[e].copy()], reward=[], done=[], actions=[])
                eps.append(cur)
            g = cur['grid'][-1].copy()
            if fx['grid_change_idx'][e, t] >= 0:
                g[fx['grid_change_idx'][e, t]] = fx['grid_change_val'][e, t]
            cur['grid'].append(g)
            cur['agentPos'].append(fx['agentPos'][e, t]); cur['inventory'].append(fx['inventory'][e, t])
            cur['compass'].append(np.array([fx['compass'][e, t]], np.float32))
            cur['reward'].append(fx['reward'][e, t]); cur['done'].append(bool(fx['done'][e, t]))
            cur['actions'].append(int(fx['actions'][e, t]))
    return eps


@pytest.mark.parametrize('name,gs', [('s3_walk_rt20', 0), ('s2_walk_cdm', 1), ('s5_scripted_leak', 64)])
def test_logged_episodes_equal_the_fixture(name, gs, tmp_path):
    from gridworld_amd.wrappers import EpisodeLogger
    fx = GR.load_fixture(name)
    E, T = fx['done'].shape
    T = min(T, 320)
    drv = HipDriver(fx, lanes_per_env=gs)
    log = EpisodeLogger(drv.env, n_envs=E, path=str(tmp_path), desc='t', glob_step=7)
    drv.set_tasks(fx['targets'], fx['starts'], invariant=fx['task_kwargs'].get('invariant', True))
    drv.reset(None)
    got = []
    for t in range(T):
        rb = fx['reset_before'][:, t].astype(bool)
        if rb.any():
            got += log.collect()          # finished episodes stay readable until the episode after next starts
            drv.reset(rb)
        drv.step_walking(fx['actions'][:, t])
    got += log.collect()
    want = [ep for ep in _expected_episodes(fx, T) if ep['done'] and ep['done'][-1]]
    assert len(got) == len(want) and len(got) >= E // 2
    got.sort(key=lambda ep: (ep['env'], ep['episode']))
    for g, w in zip(got, want):
        n = len(w['reward'])
        assert g['env'] == w['env'] and len(g['reward']) == n, (g['env'], g['episode'])
        assert np.array_equal(g['agentPos'].view(np.uint32), np.stack(w['agentPos']).astype(np.float32).view(np.uint32))
        assert np.array_equal(g['inventory'], np.stack(w['inventory'])) and g['inventory'].dtype == np.float32
        assert np.array_equal(g['compass'], np.stack(w['compass'])) and g['compass'].shape == (n + 1, 1)
        assert np.array_equal(g['grid'].reshape(n + 1, -1), np.stack(w['grid'])) and g['grid'].dtype == np.int32
        assert np.array_equal(g['reward'].astype(np.float32), np.array(w['reward'], np.float32))
        assert np.array_equal(g['done'], np.array(w['done'])) and g['done'][-1]
        assert np.array_equal(g['actions'], np.array(w['actions']))
        z = np.load(g['file'])                      # the dump itself: reference key set minus dialog / pov
        assert {'agentPos', 'inventory', 'compass', 'grid', 'reward', 'done'} <= set(z.files)
        assert np.array_equal(z['grid'], g['grid']) and '/step7/ep_t_' in g['file']
        assert [int(x) for x in open(g['file'][:-4] + '.csv').read().split()] == w['actions']


def test_log_under_autoreset_and_flying():
    """Auto-reset inside the step kernel switches the episode slot; the log of a flying env carries its action."""
    from gridworld_amd import VecGridWorld, workloads
    from gridworld_amd.wrappers import EpisodeLogger
    n = 64
    env = VecGridWorld(n, action_space='flying', size_reward=False, max_steps=12, autoreset=True)
    env.set_tasks(workloads.rt20(n, seed=2))
    log = EpisodeLogger(env, n_envs=5, path="/tmp/igw_traj_test")
    env.reset()
    g = torch.Generator(device='cpu'); g.manual_seed(0)
    eps, acts = [], []
    for t in range(40):
        a = dict(movement=torch.rand((n, 3), generator=g) * 2 - 1, camera=torch.rand((n, 2), generator=g) * 10 - 5,
                 inventory=torch.randint(0, 7, (n,), generator=g, dtype=torch.int32),
                 placement=torch.randint(0, 3, (n,), generator=g, dtype=torch.int32))
        acts.append(a)
        env.step(a)
        pos = env.agent_pos[:5].cpu().numpy().copy()
        if t % 12 == 11:   # every env just finished an episode (max_steps = 12) and was reset inside the kernel
            assert bool(env.done[:5].all()) and np.all(pos == 0)
            eps += log.collect(dump=False)
    assert len(eps) == 15 and sorted({e['episode'] for e in eps}) == [1, 2, 3]
    for e in eps:
        k = e['episode'] - 1
        assert len(e['reward']) == 12 and e['done'][-1] and not e['done'][:-1].any()
        for i in range(12):
            a = acts[12 * k + i]
            assert np.array_equal(e['actions']['movement'][i], a['movement'][e['env']].numpy())
            assert e['actions']['inventory'][i] == int(a['inventory'][e['env']])
            assert e['actions']['placement'][i] == int(a['placement'][e['env']])
        assert np.all(e['agentPos'][0] == 0) and np.any(e['agentPos'][-1] != 0)
    env.disable_trajectory_log()


def test_input_validation_and_counters():
    from gridworld_amd import IgwError, VecGridWorld, workloads
    n = 32
    env = VecGridWorld(n, action_space='flying', size_reward=False)
    tg = workloads.rt20(n, seed=1)
    with pytest.raises(ValueError):
        env.set_tasks(tg, init_pose=np.tile([11.0, 0, 0, 0, 0], (n, 1)))      # |x| > 10
    with pytest.raises(ValueError):
        env.set_tasks(tg, init_pose=np.tile([0, np.nan, 0, 0, 0], (n, 1)))
    with pytest.raises(ValueError):
        env.set_tasks(tg, env_task=np.full(n, n, np.int32))                   # index outside the table
    env.set_tasks(tg)
    # the C ABI itself (a caller that skips the Python checks): bad poses are replaced by the default and counted
    import ctypes as C
    from gridworld_amd import _lib as L
    pose = torch.tensor(np.tile([0.0, 0, 0, 0, 0], (n, 1)), device=env.device)
    pose[3, 0] = 50.0
    pose[5, 2] = float('inf')
    rows = torch.zeros((n, L.GRID_STRIDE), dtype=torch.int8, device=env.device)
    L.check(env.lib.igw_prepare_tasks(env.ctx, 0, n, rows.data_ptr(), None, None, None, pose.data_ptr(), env._stream()), 'prep')
    env.reset()
    torch.cuda.synchronize()
    assert env.stats()['bad_poses'] == 2 and np.all(env.internals()[:, :5] == 0)
    with pytest.raises(ValueError):
        env.step(dict(movement=torch.zeros((n - 1, 3)), camera=torch.zeros((n, 2)), inventory=torch.zeros(n, dtype=torch.int32),
                      placement=torch.zeros(n, dtype=torch.int32)))
    mv = torch.zeros((n, 3)); mv[2, 1] = float('nan')
    inv = torch.zeros(n, dtype=torch.int32); inv[7] = 9
    env.step(dict(movement=mv, camera=torch.zeros((n, 2)), inventory=inv, placement=torch.zeros(n, dtype=torch.int32)))
    torch.cuda.synchronize()
    assert env.stats()['bad_actions'] == 2 and np.isfinite(env.internals()).all()
    # A finite but huge camera delta would spin the reference's `while yaw > 360: yaw -= 360` forever (a hung
    # kernel): host data raises, device data (not read back) runs as a no-op component and is counted.
    cam = torch.zeros((n, 2)); cam[4, 0] = 1e30; cam[9, 1] = -3e7; cam[11, 0] = float('inf')
    zero = dict(movement=torch.zeros((n, 3)), inventory=torch.zeros(n, dtype=torch.int32), placement=torch.zeros(n, dtype=torch.int32))
    with pytest.raises(ValueError):
        env.step(dict(camera=cam, **zero))
    before = env.internals().copy()
    env.step(dict(camera=cam.to(env.device), **zero))
    torch.cuda.synchronize()
    assert env.stats()['bad_actions'] == 2 + 3
    after = env.internals()
    assert np.isfinite(after).all() and np.array_equal(after[:, 3:5], before[:, 3:5])   # yaw / pitch untouched
    with pytest.raises(ValueError):
        env.rollout_actions(dict(movement=torch.zeros((2, n, 3)), camera=cam.expand(2, n, 2), inventory=torch.zeros((2, n), dtype=torch.int32),
                                 placement=torch.zeros((2, n), dtype=torch.int32)))
    env.rollout_actions(dict(movement=torch.zeros((2, n, 3)), camera=cam.expand(2, n, 2).to(env.device),
                             inventory=torch.zeros((2, n), dtype=torch.int32), placement=torch.zeros((2, n), dtype=torch.int32)))
    torch.cuda.synchronize()
    assert env.stats()['bad_actions'] == 5 + 6 and np.array_equal(env.internals()[:, 3:5], before[:, 3:5])
    wd = VecGridWorld(n, discretize=False, size_reward=False)
    wd.set_tasks(tg)
    wd.reset()
    wd.step(dict(buttons=torch.zeros((n, 8), dtype=torch.uint8), camera=cam.to(wd.device)))
    torch.cuda.synchronize()
    assert wd.stats()['bad_actions'] == 3 and np.all(wd.internals()[:, 3:5] == 0)
    with pytest.raises(TypeError):
        VecGridWorld(n, max_step=100)          # a misspelt create_env kwarg is an error, as in the reference
    # block ids are 0..7: Python raises, the C ABI counts the row and never matches such a cell
    odd = tg.clone(); odd[3, 0, 5, 5] = 9
    with pytest.raises(ValueError):
        wd.set_tasks(odd)
    rows9 = torch.zeros((n, L.GRID_STRIDE), dtype=torch.int8, device=wd.device)
    rows9[:, :1089] = odd.reshape(n, -1).to(wd.device)
    L.check(wd.lib.igw_prepare_tasks(wd.ctx, 0, n, rows9.data_ptr(), None, None, None, None, wd._stream()), 'prep')
    torch.cuda.synchronize()
    assert wd.stats()['bad_tasks'] == 1
    # ... and the row stays consistent: the cell is read as empty (target size, colour index)
    clean = odd[3].clone(); clean[0, 5, 5] = 0
    assert int(wd.task_meta[3, 40:42].cpu().numpy().view(np.int16)[0]) == int((clean != 0).sum())
    # the fused loops do not write the episode log: refused while it is enabled (C ABI: IGW_ERR_INVALID)
    w = VecGridWorld(n, autoreset=True)
    w.set_tasks(tg)
    w.reset()
    w.enable_trajectory_log(2)
    with pytest.raises(IgwError):
        w.rollout(4, seed=1)
    acts4 = w.fill_actions(4, seed=1)
    assert w.lib.igw_rollout_walking_actions(w.ctx, acts4.data_ptr(), 4, None, None, w._stream()) == -1
    w.disable_trajectory_log()
    w.rollout(4, seed=1)
    with pytest.raises(ValueError):
        w.step(torch.zeros(n + 1, dtype=torch.int32))
    with pytest.raises(IgwError):
        from gridworld_amd._lib import Config
        cfg = Config.from_buffer_copy(w.cfg); cfg.reserved = 1
        ctx = C.c_void_p()
        L.check(w.lib.igw_create(C.byref(cfg), C.byref(ctx)), 'igw_create')     # ablation switches: IGW_DIAG build only
    with pytest.raises(IgwError):
        L.check(w.lib.igw_debug_set_stamps(w.ctx, None), 'stamps')


def test_logged_wrapper_on_the_facade(tmp_path):
    """The reference's calling sequence: Logged(env); turn_on(); set_path(); an npz per finished episode
    (wrappers.py:66-134), here checked against a fixture replayed through the 1-env facade."""
    import gridworld_amd as G
    from gridworld_amd.wrappers import Logged
    fx = GR.load_fixture('s5_scripted_leak')      # max_steps = 12: several short episodes
    e = 1
    env = Logged(G.make('IGLUGridworldVector-v0', **fx['kwargs']))
    env.set_path(str(tmp_path))
    env.set_desc('facade', 3)
    env.set_task(G.Task('', fx['targets'][e].astype(np.int32), starting_grid=G.Tasks.dense_to_sparse(fx['starts'][e].astype(np.int32))))
    env.reset()
    assert env.max_steps == fx['kwargs']['max_steps']          # attribute pass-through to the wrapped env
    n_done = 0
    for t in range(fx['done'].shape[1]):
        if fx['reset_before'][e, t]:
            env.reset()
        if t == 13:
            env.turn_on()                                        # episodes that end from here on are written
        obs, reward, done, _ = env.step(int(fx['actions'][e, t]))
        assert done == bool(fx['done'][e, t])
        n_done += done
    files = sorted(p for p in (tmp_path / 'step3').iterdir() if p.suffix == '.npz')
    assert n_done >= 3 and len(files) == n_done - 1              # the first episode ended before turn_on()
    z = np.load(files[0])
    assert z['agentPos'].shape[1:] == (5,) and z['grid'].shape[1:] == (9, 11, 11) and len(z['reward']) == len(z['done'])
    assert z['done'][-1] and len(z['agentPos']) == len(z['reward']) + 1
