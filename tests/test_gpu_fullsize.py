"""BASELINE full size (65,536 parallel envs, configs[2]/[3]) through size-independent properties:
  * invariants that tie the outputs together (bitmap == grid != 0, inventory == 20 - colour counts,
    prev_size == block count, persistent histogram == recount, max_int bounds);
  * shard independence: every env computed inside the 65,536 batch equals the same env computed in a
    4,096-env shard (so sharding over GPUs cannot change results);
  * a random sample of the batch replayed through the CPU oracle (bit-exact);
  * determinism (two runs -> identical bytes) and fused rollout == step-by-step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N = 65536


def _run(n, T, seed, env_offset=0, task_seed=0, task_slice=None, mode='walking', rollout=False):
    from gridworld_amd import VecGridWorld, workloads
    env = VecGridWorld(n, size_reward=False, autoreset=True, max_steps=100, action_space=mode)
    tg = workloads.rt20(N, seed=task_seed)
    if task_slice is not None:
        tg = tg[task_slice]
    env.set_tasks(tg.to(env.device))
    env.reset()
    if rollout:
        env.rollout(T, seed=seed, env_offset=env_offset)
    else:
        acts = env.fill_actions(T, seed=seed, env_offset=env_offset)
        for t in range(T):
            env.step_walking_ptr(acts[t])
    torch.cuda.synchronize()
    return env, tg


def test_invariants_at_full_size():
    from test_gpu_parity import _check_occ, _check_hist
    T = 230  # two auto-resets (max_steps=100) plus a partial episode
    env, tg = _run(N, T, seed=321)
    st = env.stats()
    assert st['resets'] >= 2 * N
    grid = env.grid.cpu().numpy().reshape(N, -1)
    inv = env.inventory.cpu().numpy()
    ts = env.task_state()
    _check_occ(env)
    # inventory conservation: no starting grid -> inventory[c] == 20 - #cells of colour c+1
    counts = np.stack([(grid == c + 1).sum(1) for c in range(6)], axis=1)
    assert np.array_equal(inv, (20 - counts).astype(np.float32))
    assert np.array_equal(ts['prev_size'], (grid != 0).sum(1))
    tsize = (tg.numpy().reshape(N, -1) != 0).sum(1)
    assert (ts['max_int'] <= np.minimum(tsize, ts['prev_size'])).all() and (ts['max_int'] >= 0).all()
    assert (ts['step_no'] == T - 200).all() or st['resets'] > 2 * N   # completed targets reset earlier
    assert ts['dirty'].sum() == 0                                      # impossible without a starting grid
    _check_hist(env, tg.numpy(), sample=range(0, N, 997))
    # checksum of checksums, stable across runs of the same seed (determinism)
    env2, _ = _run(N, T, seed=321)
    for a, b in ((env.grid_buf, env2.grid_buf), (env.agent_buf, env2.agent_buf), (env.hist_buf, env2.hist_buf),
                 (env.occ_buf, env2.occ_buf), (env.reward, env2.reward), (env.agent_pos, env2.agent_pos)):
        assert torch.equal(a, b)


def test_shard_independence_and_oracle_sample():
    from oracle import oracle as O
    T = 150
    full, tg = _run(N, T, seed=77)
    fg, fa = full.grid_buf.cpu(), full.agent_buf.cpu()
    # (a) 4,096-env shards with rank-style offsets reproduce their slice of the big batch
    for lo in (0, 20480, 61440):
        part, _ = _run(4096, T, seed=77, env_offset=lo, task_slice=slice(lo, lo + 4096))
        assert torch.equal(part.grid_buf.cpu(), fg[lo:lo + 4096])
        assert torch.equal(part.agent_buf.cpu(), fa[lo:lo + 4096])
    # (b) 256 envs sampled across the batch, replayed on the CPU oracle with the same counter-RNG actions
    idx = np.random.RandomState(3).choice(N, 256, replace=False)
    acts = full.fill_actions(T, seed=77).cpu().numpy()[:, idx]
    ob = O.OracleBatch(len(idx), size_reward=False, max_steps=100)
    ob.set_tasks(tg.numpy()[idx])
    ob.reset()
    for t in range(T):
        ob.step_walking(acts[t], autoreset=True, nthreads=8)
    assert np.array_equal(full.grid.cpu().numpy().reshape(N, -1)[idx], ob.grid)
    assert np.array_equal(full.internals()[idx].view(np.uint64), ob.internals().view(np.uint64))
    assert np.array_equal(full.reward.cpu().numpy()[idx], ob.reward)
    assert np.array_equal(full.agent_pos.cpu().numpy()[idx].view(np.uint32), ob.agentPos.view(np.uint32))


def test_whole_batch_equals_oracle_at_full_size():
    """BASELINE configs[2] as the bench runs it -- 65,536 envs, rt20 targets, uniform random walking actions, default
    max_steps = 250 so every env auto-resets once inside the run -- with EVERY env replayed through the CPU oracle
    (counter-RNG actions on both sides, 16 host threads): grid, float64 internals, inventory and the last step's
    observations / reward / done of all 65,536 envs bit for bit.  17 M env-steps on either side."""
    from gridworld_amd import VecGridWorld, workloads
    from oracle import oracle as O
    T, seed = 260, 20260
    tg = workloads.rt20(N, seed=5)
    env = VecGridWorld(N, size_reward=False, autoreset=True)
    env.set_tasks(tg.to(env.device))
    env.reset()
    acts = env.fill_actions(T, seed=seed)
    for t in range(T):
        env.step(acts[t])
    torch.cuda.synchronize()
    ob = O.OracleBatch(N, size_reward=False)
    ob.set_tasks(tg.numpy())
    ob.reset()
    steps, changed = ob.rollout_walking(T, seed, autoreset=True, nthreads=16)
    assert steps == N * T and env.stats()['changed'] == changed and env.stats()['resets'] >= N
    assert np.array_equal(env.internals().view(np.uint64), ob.internals().view(np.uint64))
    grid = env.grid.cpu().numpy().reshape(N, -1)
    inv = env.inventory.cpu().numpy()
    pos = env.agent_pos.cpu().numpy().view(np.uint32)
    for lo in range(0, N, 4096):   # (the oracle's observations env by env, a slice at a time)
        obs = [e.obs() for e in ob.envs[lo:lo + 4096]]
        assert np.array_equal(grid[lo:lo + 4096], np.stack([o['grid'].reshape(-1) for o in obs]).astype(np.int8)), lo
        assert np.array_equal(inv[lo:lo + 4096], np.stack([o['inventory'] for o in obs]).astype(np.float32)), lo
        assert np.array_equal(pos[lo:lo + 4096], np.stack([o['agentPos'] for o in obs]).astype(np.float32).view(np.uint32)), lo


def test_fused_rollout_equals_stepwise_at_full_size():
    T = 120
    a, _ = _run(N, T, seed=9)
    b, _ = _run(N, T, seed=9, rollout=True)
    assert torch.equal(a.grid_buf, b.grid_buf) and torch.equal(a.agent_buf, b.agent_buf)
    assert torch.equal(a.hist_buf, b.hist_buf) and torch.equal(a.occ_buf, b.occ_buf)
    assert a.stats()['changed'] == b.stats()['changed'] and a.stats()['resets'] == b.stats()['resets']


def test_flying_full_size_invariants():
    """configs[3]: 65,536 flying envs -- invariants, and EVERY env replayed through the oracle in device-trig mode
    (grid and float64 internals of the whole batch bit for bit)."""
    from gridworld_amd import VecGridWorld, workloads
    from oracle import oracle as O
    from test_gpu_parity import _check_occ
    T = 60
    env = VecGridWorld(N, size_reward=False, autoreset=True, max_steps=50, action_space='flying')
    tg = workloads.rt20(N, seed=4)
    env.set_tasks(tg.to(env.device))
    env.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(5)
    mv = torch.rand((T, N, 3), generator=g, device=env.device) * 2 - 1
    cam = torch.rand((T, N, 2), generator=g, device=env.device) * 10 - 5
    inv = torch.randint(0, 7, (T, N), generator=g, device=env.device, dtype=torch.int32)
    plc = torch.randint(0, 3, (T, N), generator=g, device=env.device, dtype=torch.int32)
    idx = np.arange(N)
    O.use_device_trig(True)
    try:
        ob = O.OracleBatch(len(idx), size_reward=False, max_steps=50, action_space='flying')
        ob.set_tasks(tg.numpy()[idx])
        ob.reset()
        for t in range(T):
            env.step(dict(movement=mv[t], camera=cam[t], inventory=inv[t], placement=plc[t]))
            ob.step_flying(mv[t].cpu().numpy()[idx], cam[t].cpu().numpy()[idx], inv[t].cpu().numpy()[idx],
                           plc[t].cpu().numpy()[idx], autoreset=True, nthreads=16)
        torch.cuda.synchronize()
        assert np.array_equal(env.grid.cpu().numpy().reshape(N, -1)[idx], ob.grid)
        assert np.array_equal(env.internals()[idx].view(np.uint64), ob.internals().view(np.uint64))
    finally:
        O.use_device_trig(False)
    _check_occ(env)
    grid = env.grid.cpu().numpy().reshape(N, -1)
    counts = np.stack([(grid == c + 1).sum(1) for c in range(6)], axis=1)
    assert np.array_equal(env.inventory.cpu().numpy(), (20 - counts).astype(np.float32))
    assert np.array_equal(env.task_state()['prev_size'], (grid != 0).sum(1))


def test_config4_workload_on_one_gpu_equals_eight_rank_shards():
    """BASELINE configs[4]: 524,288 parallel envs, walking, random targets, sharded over 8 GPUs -- run here on ONE GPU
    (it fits: 2.6 GB) and compared with the eight 65,536-env shards a node would run, each in its own context with
    env_index_base = rank * 65,536 (igw_config.env_index_base keys the on-device RandomTasks generator) and
    rank-offset action streams, exactly as bench.py sets its ranks up.  Sharding must not change a single byte."""
    from gridworld_amd import VecGridWorld
    R, n = 8, N
    T = 70
    kw = dict(size_reward=False, autoreset=True, max_steps=30)   # two auto-resets: the generator runs inside the step
    rt = dict(seed=777, max_blocks=20, height_levels=1, max_dist=2, num_colors=6)

    def run(num, base):
        env = VecGridWorld(num, env_index_base=base, **kw)
        env.set_random_tasks(**rt)
        env.reset()
        acts = env.fill_actions(T, seed=4321, env_offset=base)
        for t in range(T):
            env.step_walking_ptr(acts[t])
        torch.cuda.synchronize()
        return env

    big = run(R * n, 0)
    st = big.stats()
    assert st['resets'] >= 2 * R * n and st['changed'] > 0
    ts = (big.task_target[:, :1089] != 0).sum(1)
    assert int(ts.min()) >= 1 and int(ts.max()) == 20 and big.task_target.unique().numel() == 7   # rt20-shaped targets
    for r in range(R):
        part = run(n, r * n)
        sl = slice(r * n, (r + 1) * n)
        for name in ('grid_buf', 'agent_buf', 'hist_buf', 'occ_buf', 'task_target', 'task_meta', 'episode',
                     'agent_pos', 'inventory', 'reward', 'done'):
            assert torch.equal(getattr(part, name), getattr(big, name)[sl]), (r, name)
        del part


def test_cdm_workload_full_size_properties():
    """65,536 envs tiled from the real IGLU targets (tests/golden/cdm_goals.npz) with random partial starting grids:
    has_start rows, task_start reads on every change, negative synthetic ids, inventories below 20 and the stale-cache
    (`dirty`) path at scale -- invariants over the whole batch, a sample replayed through the CPU oracle."""
    import os
    from gridworld_amd import VecGridWorld, workloads
    from oracle import oracle as O
    from test_gpu_parity import _check_occ, _check_hist
    goals = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cdm_goals.npz'))['dense']
    T = 140
    kw = dict(size_reward=False, max_steps=60)
    tg, st = workloads.cdm(N, 5, goals)
    env = VecGridWorld(N, autoreset=True, **kw)
    env.set_tasks(tg.to(env.device), st.to(env.device))
    env.reset()
    acts = env.fill_actions(T, seed=606)
    seen_dirty = 0
    for t in range(T):
        env.step_walking_ptr(acts[t])
        if t % 35 == 34:
            seen_dirty += int(env.task_state()['dirty'].sum())
    torch.cuda.synchronize()
    tgn, stn = tg.numpy(), st.numpy()
    grid = env.grid.cpu().numpy().reshape(N, -1)
    stats = env.stats()
    assert stats['resets'] >= 2 * N and stats['changed'] > 0
    assert stats['rescans'] > stats['changed']        # cell changes that left the block count alone (recolour in place)
    assert seen_dirty > 0                             # ... which is the stale-cache path of tasks/task.py:112
    _check_occ(env)
    # inventory conservation with a starting grid: 20 - blocks of that colour in the world (env.py:243-246 + callbacks)
    counts = np.stack([(grid == c + 1).sum(1) for c in range(6)], axis=1)
    assert np.array_equal(env.inventory.cpu().numpy(), (20 - counts).astype(np.float32))
    syn = grid.astype(np.int32) - stn.reshape(N, -1)
    ts = env.task_state()
    assert np.array_equal(ts['prev_size'], (syn != 0).sum(1))
    _check_hist(env, tgn, starts=stn, sample=range(0, N, 1499))
    idx = np.random.RandomState(12).choice(N, 192, replace=False)
    a_np = acts.cpu().numpy()[:, idx]
    ob = O.OracleBatch(len(idx), **kw)
    ob.set_tasks(tgn[idx], stn[idx])
    ob.reset()
    for t in range(T):
        ob.step_walking(a_np[t], autoreset=True, nthreads=8)
    assert np.array_equal(grid[idx], ob.grid)
    assert np.array_equal(env.internals()[idx].view(np.uint64), ob.internals().view(np.uint64))
    assert np.array_equal(env.reward.cpu().numpy()[idx], ob.reward)
    assert np.array_equal(env.inventory.cpu().numpy()[idx], ob.inventory)


def _rng_task_host(seed, env, episode, n):
    """Host restatement (numpy uint64) of the device's task sampler -- csrc/igw_device.h rng_task(): the row an env
    draws at the reset that ends episode `episode` -- so the oracle replay below does not ask the device which task
    it chose (CustomTasks.reset, gridworld/tasks/task_set.py:53-56: uniform over the table)."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)

    def splitmix(z):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
        return z ^ (z >> np.uint64(31))
    with np.errstate(over='ignore'):
        env = np.asarray(env, np.uint64)
        episode = np.asarray(episode, np.uint64)
        h = splitmix(np.uint64(seed) ^ splitmix(env * np.uint64(0x9E3779B1) + episode * np.uint64(0x100000001B3) + np.uint64(0x7461736B)))
        return (((h >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def test_extra_kernel_variant_whole_batch_equals_oracle_at_full_size():
    """The EXTRA instantiation of the step kernel (step_kernel<4, 0, true>: episode log + device-side samplers compiled
    in) at BASELINE's full batch: 65,536 envs, the on-device task sampler drawing every episode's task from a 4,096-row
    table (a third of the rows with a starting grid), the episode log on for the first 64 envs, auto-reset inside the
    kernel, 260 steps with max_steps = 100 -- and EVERY env replayed through the CPU oracle step by step: done and
    reward of the whole batch every step, the task rows the device drew against a host restatement of its sampler, the
    decoded log of every finished episode of the logged envs against the oracle's own record (gridworld/wrappers.py:89-121),
    and at the end grid, float64 internals, inventory and observations of all 65,536 envs bit for bit."""
    from fuzz_parity import LogChecker, compare
    from gridworld_amd import VecGridWorld, workloads
    from oracle import oracle as O
    T, ntasks, samp_seed, n_logged = 260, 4096, 424242, 64
    kw = dict(size_reward=False, max_steps=100)
    tg = workloads.rt20(ntasks, seed=61).numpy()
    st = np.zeros_like(tg)
    st[::3, 0, 5, 5] = 2                      # a third of the rows start with a block in place
    st[1::3, 1, 3, 7] = 5                     # ... another third with one that is (usually) not part of the target
    env = VecGridWorld(N, num_tasks=ntasks, autoreset=True, **kw)
    env.set_tasks(tg, st, env_task=np.zeros(N, np.int32))
    env.set_task_sampling(True, seed=samp_seed)
    ob = O.OracleBatch(N, **kw)
    checker = LogChecker(env, ob, n_logged, kw['max_steps'], 'EXTRA full size')
    env.reset()
    torch.cuda.synchronize()
    assert env.cfg.lanes_per_env in (0, 4)
    episode = np.zeros(N, np.int64)           # episodes started so far = the sampler's key of the NEXT draw
    row = _rng_task_host(samp_seed, np.arange(N), episode, ntasks)
    episode += 1
    assert np.array_equal(env.env_task.cpu().numpy(), row)
    for e in range(N):
        ob.envs[e].set_task(tg[row[e]], st[row[e]])
    ob.reset()
    checker.begin(None)
    acts = env.fill_actions(T, seed=8086)
    acts_h = acts.cpu().numpy()
    n_resets = 0
    try:
        for t in range(T):
            env.step(acts[t])
            ob.step_walking(acts_h[t], autoreset=False, nthreads=16)
            checker.step()
            torch.cuda.synchronize()
            done = ob.done.astype(bool)
            assert np.array_equal(env.done.cpu().numpy().astype(bool), done), t
            assert np.array_equal(env.reward.cpu().numpy().view(np.uint32), ob.reward.view(np.uint32)), t
            if done.any():
                idx = np.nonzero(done)[0]
                row[idx] = _rng_task_host(samp_seed, idx, episode[idx], ntasks)
                episode[idx] += 1
                n_resets += len(idx)
                rew, dn = ob.reward.copy(), ob.done.copy()
                for e in idx:
                    ob.envs[e].set_task(tg[row[e]], st[row[e]])
                ob.reset(done)
                ob.reward[:], ob.done[:] = rew, dn      # (reward / done stay the step's, as on the device)
                checker.begin(done)
            checker.check()
            if t % 20 == 19 or t == T - 1:
                assert np.array_equal(env.env_task.cpu().numpy(), row), t
                compare(env, ob, f'EXTRA full size, step {t}')
        assert np.array_equal(env.internals().view(np.uint64), ob.internals().view(np.uint64))
        s = env.stats()
        assert s['resets'] == n_resets and n_resets >= 2 * N and s['steps'] == N * T
        assert checker.checked >= 2 * n_logged
        picks = np.bincount(row, minlength=ntasks)
        assert picks.max() <= 60 and (picks > 0).mean() > 0.99      # 16 draws per row on average: roughly uniform
    finally:
        env.disable_trajectory_log()
