"""RandomTasks.sample_task on the device (igw_set_random_tasks; gridworld/tasks/task_set.py:135-157): no host in the
reset path.  Same distribution as the reference's procedure (checked against the host mirror, which is stream-matched
to the reference in tests/test_generators.py), structural invariants of every sample, determinism, and bit-exact
episodes on the sampled targets against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_samples(n, **kw):
    from gridworld_amd.tasks import RandomTasks
    gen = RandomTasks(**kw)
    return np.stack([np.asarray(gen.sample_task().target_grid, np.int8) for _ in range(n)])


def _device_samples(n_envs, rounds, seed, **kw):
    from gridworld_amd import VecGridWorld
    env = VecGridWorld(n_envs, size_reward=False, autoreset=True, max_steps=1)
    env.set_random_tasks(True, seed=seed, **kw)
    out = []
    env.reset()
    out.append(env.targets().cpu().numpy().copy())
    acts = torch.zeros(n_envs, dtype=torch.int32, device=env.device)
    for _ in range(rounds - 1):   # max_steps = 1: every step ends the episode and auto-resets with a fresh task
        env.step(acts)
        out.append(env.targets().cpu().numpy().copy())
    torch.cuda.synchronize()
    assert env.stats()['resets'] == n_envs * (rounds - 1)
    return np.concatenate(out), env


@pytest.mark.parametrize('kw', [dict(max_blocks=20, height_levels=1, max_dist=2, num_colors=6),
                                dict(max_blocks=6, height_levels=2, max_dist=2, num_colors=3),
                                dict(max_blocks=4, height_levels=1, max_dist=1, num_colors=1)])
def test_distribution_matches_host_mirror(kw):
    n = 120_000
    dev, _ = _device_samples(4096, (n + 4095) // 4096, seed=11, **kw)
    dev = dev[:n]
    np.random.seed(123)
    n_host = 30_000
    host = _host_samples(n_host, **kw)
    L = kw['height_levels']
    assert (dev[:, L:] == 0).all()
    # (1) every sample obeys the procedure: per level min(max_blocks, window) blocks, all inside a (2d+1)^2 window
    # around one of them, colours in range
    occ = dev[:, :L] != 0
    cnt = occ.reshape(n, L, -1).sum(-1)
    assert cnt.max() <= kw['max_blocks'] and dev.max() <= kw['num_colors'] and dev.min() >= 0
    for lvl in range(L):
        xs = np.where(occ[:, lvl].any(2), np.arange(11)[None, :], -1)
        zs = np.where(occ[:, lvl].any(1), np.arange(11)[None, :], -1)
        xspan = xs.max(1) - np.where(xs < 0, 99, xs).min(1)
        zspan = zs.max(1) - np.where(zs < 0, 99, zs).min(1)
        assert xspan.max() <= 2 * kw['max_dist'] and zspan.max() <= 2 * kw['max_dist']
    # (2) the same distribution as the reference's procedure: block-count histogram, per-cell occupancy, colours
    hcnt = (host[:, :L] != 0).reshape(n_host, L, -1).sum(-1)
    for k in range(1, kw['max_blocks'] + 1):
        assert abs((cnt == k).mean() - (hcnt == k).mean()) < 0.01, k
    p_dev, p_host = occ.mean(0), (host[:, :L] != 0).mean(0)
    se = np.sqrt(p_host * (1 - p_host) / n_host + p_dev * (1 - p_dev) / n) + 1e-9
    assert (np.abs(p_dev - p_host) / se).max() < 5.5, 'per-cell occupancy differs from the reference procedure'
    for c in range(1, kw['num_colors'] + 1):
        assert abs((dev[:, :L][occ] == c).mean() - 1 / kw['num_colors']) < 0.01
        assert abs((host[:, :L][host[:, :L] != 0] == c).mean() - 1 / kw['num_colors']) < 0.02


def test_generated_tasks_carry_their_colour_index():
    """The generator also writes the colour index the step kernels vote from (include/igw.h), for reset_kernel and
    for the in-step auto-reset alike."""
    from test_gpu_parity import _check_index
    for rounds in (1, 3):
        _, env = _device_samples(640, rounds, seed=21, max_blocks=7, height_levels=3, max_dist=2, num_colors=5)
        _check_index(env, rows=range(0, 640, 9))


def test_generated_tasks_are_deterministic_and_seeded():
    kw = dict(max_blocks=8, height_levels=1, max_dist=2, num_colors=4)
    a, _ = _device_samples(512, 5, seed=5, **kw)
    b, _ = _device_samples(512, 5, seed=5, **kw)
    c, _ = _device_samples(512, 5, seed=6, **kw)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert len({x.tobytes() for x in a}) > 0.98 * len(a)   # envs and episodes draw different tasks


@pytest.mark.parametrize('gs', [0, 1, 64])
def test_episodes_on_generated_tasks_match_oracle(gs):
    """Auto-reset regenerates the target on the device; the oracle is handed each new target and must agree on every
    output bit, metadata (target size, admissible boxes, inventory) included via reward / done."""
    from gridworld_amd import VecGridWorld
    from oracle import oracle as O
    n, T = 384, 150
    kw = dict(size_reward=False, max_steps=40)
    env = VecGridWorld(n, autoreset=True, lanes_per_env=gs, **kw)
    env.set_random_tasks(True, seed=42, max_blocks=5, height_levels=2, max_dist=2, num_colors=3)
    env.reset()
    torch.cuda.synchronize()
    tg = env.targets().cpu().numpy()
    envs = [O.OracleEnv(**kw) for _ in range(n)]
    for e, o in enumerate(envs):
        o.set_task(tg[e], None)
        o.reset()
    rng = np.random.RandomState(3)
    acts = rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 13, 14, 14, 15, 16, 17, 17], size=(T, n)).astype(np.int32)
    acts[:5] = 14   # look down first, so blocks get placed and rewards flow
    n_resets = 0
    for t in range(T):
        env.step(torch.as_tensor(acts[t]))
        torch.cuda.synchronize()
        done, rew = env.done.cpu().numpy().astype(bool), env.reward.cpu().numpy()
        new = env.targets().cpu().numpy() if done.any() else None
        for e, o in enumerate(envs):
            _, r, d, _ = o.step(int(acts[t, e]))
            assert d == done[e] and np.float32(r) == rew[e], (t, e)
            if d:
                assert not np.array_equal(new[e], tg[e]) or new[e].any()
                tg[e] = new[e]
                o.set_task(new[e], None)
                o.reset()
                n_resets += 1
    assert n_resets >= n * (T // 40)
    grid = env.grid.cpu().numpy().reshape(n, -1)
    internals = env.internals()
    for e, o in enumerate(envs):
        assert np.array_equal(grid[e], o.obs()['grid'].reshape(-1).astype(np.int8)), e
        assert np.array_equal(internals[e].view(np.uint64), o.internal().view(np.uint64)), e
    # the fused rollout regenerates tasks the same way (same keys => same tasks as the per-step path)
    a = VecGridWorld(256, autoreset=True, **kw)
    b = VecGridWorld(256, autoreset=True, **kw)
    for v in (a, b):
        v.set_random_tasks(True, seed=9, max_blocks=5, max_dist=2, num_colors=2)
        v.reset()
    a.rollout(130, seed=1)
    acts2 = b.fill_actions(130, seed=1)
    for t in range(130):
        b.step_walking_ptr(acts2[t])
    torch.cuda.synchronize()
    assert torch.equal(a.task_target, b.task_target) and torch.equal(a.grid_buf, b.grid_buf)
    assert torch.equal(a.agent_buf, b.agent_buf) and torch.equal(a.episode, b.episode)


def test_random_tasks_argument_checks():
    from gridworld_amd import IgwError, VecGridWorld
    env = VecGridWorld(8, num_tasks=4)
    with pytest.raises(IgwError):
        env.set_random_tasks(True)             # needs one task row per env
    env = VecGridWorld(8)
    with pytest.raises(IgwError):
        env.set_random_tasks(True, num_colors=7)
    env.set_random_tasks(True, max_blocks=3)
    with pytest.raises(IgwError):
        env.set_task_sampling(True, n_tasks=8)  # mutually exclusive
    env.set_random_tasks(False)
