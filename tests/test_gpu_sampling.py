"""Device-side task sampling at reset (CustomTasks.reset semantics) and state snapshots."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_task_sampling_on_autoreset_matches_oracle():
    from gridworld_amd import VecGridWorld, workloads
    from oracle import oracle as O
    n, T, ntasks = 1500, 130, 37
    kw = dict(size_reward=False, max_steps=30)
    tg = workloads.rt20(ntasks, seed=31).numpy()
    st = np.zeros_like(tg)
    st[::3, 0, 5, 5] = 2   # a third of the tasks come with a starting block
    env = VecGridWorld(n, num_tasks=ntasks, autoreset=True, **kw)
    env.set_tasks(tg, st, env_task=np.zeros(n, np.int32))
    env.set_task_sampling(True, seed=99)
    env.reset()
    torch.cuda.synchronize()
    cur = env.env_task.cpu().numpy().copy()
    assert cur.min() >= 0 and cur.max() < ntasks and len(np.unique(cur)) == ntasks
    envs = [O.OracleEnv(**kw) for _ in range(n)]
    for e, o in enumerate(envs):
        o.set_task(tg[cur[e]], st[cur[e]])
        o.reset()
    acts = env.fill_actions(T, seed=5).cpu().numpy()
    seen = [cur.copy()]
    for t in range(T):
        env.step(torch.as_tensor(acts[t]))
        torch.cuda.synchronize()
        done = env.done.cpu().numpy().astype(bool)
        new = env.env_task.cpu().numpy()
        rew = env.reward.cpu().numpy()
        for e, o in enumerate(envs):
            _, r, d, _ = o.step(int(acts[t, e]))
            assert d == done[e] and np.float32(r) == rew[e], (t, e)
            if d:   # the device picked the next task; tell the oracle which one and reset it
                o.set_task(tg[new[e]], st[new[e]])
                o.reset()
            else:
                assert new[e] == cur[e]
        cur = new.copy()
        if done.any():
            seen.append(new[done])
    grid = env.grid.cpu().numpy().reshape(n, -1)
    internals = env.internals()
    for e, o in enumerate(envs):
        assert np.array_equal(grid[e], o.obs()['grid'].reshape(-1).astype(np.int8)), e
        assert np.array_equal(internals[e].view(np.uint64), o.internal().view(np.uint64)), e
    picks = np.concatenate(seen)
    counts = np.bincount(picks, minlength=ntasks)
    assert counts.min() > 0.5 * counts.mean() and counts.max() < 1.5 * counts.mean()   # roughly uniform


def test_sampling_is_deterministic_and_rollout_consistent():
    from gridworld_amd import VecGridWorld, workloads
    n, T, ntasks = 2048, 90, 11
    tg = workloads.rt20(ntasks, seed=1)

    def run(rollout):
        env = VecGridWorld(n, num_tasks=ntasks, autoreset=True, size_reward=False, max_steps=25)
        env.set_tasks(tg.to(env.device), env_task=np.zeros(n, np.int32))
        env.set_task_sampling(True, seed=7)
        env.reset()
        if rollout:
            env.rollout(T, seed=3)
        else:
            a = env.fill_actions(T, seed=3)
            for t in range(T):
                env.step_walking_ptr(a[t])
        torch.cuda.synchronize()
        return env
    a, b, c = run(False), run(False), run(True)
    for x, y in ((a, b), (a, c)):
        assert torch.equal(x.env_task, y.env_task) and torch.equal(x.grid_buf, y.grid_buf)
        assert torch.equal(x.agent_buf, y.agent_buf) and torch.equal(x.hist_buf, y.hist_buf)


def test_state_dict_roundtrip():
    from gridworld_amd import VecGridWorld, workloads
    n = 512
    env = VecGridWorld(n, autoreset=True, size_reward=False, max_steps=40)
    env.set_tasks(workloads.rt20(n, seed=2).to(env.device))
    env.reset()
    acts = env.fill_actions(120, seed=8)
    for t in range(60):
        env.step_walking_ptr(acts[t])
    snap = env.state_dict()
    for t in range(60, 120):
        env.step_walking_ptr(acts[t])
    torch.cuda.synchronize()
    final = env.state_dict()
    env2 = VecGridWorld(n, autoreset=True, size_reward=False, max_steps=40)
    env2.load_state_dict(snap)
    for t in range(60, 120):
        env2.step_walking_ptr(acts[t])
    torch.cuda.synchronize()
    for k, v in final.items():
        if k not in ('stats_buf', 'abi_version'):
            assert torch.equal(v, getattr(env2, k)), k
    # a snapshot that is not complete (e.g. one of an older ABI, without the colour index) must not load
    partial = {k: v for k, v in snap.items() if k != 'task_index'}
    with pytest.raises(ValueError):
        env2.load_state_dict(partial)
    with pytest.raises(ValueError):
        env2.load_state_dict({**snap, 'abi_version': 3})


def test_sub_batches_on_streams_equal_whole_batch():
    """VecGridWorld.split: free-running sub-batches on their own streams give the same bytes as one batch."""
    from gridworld_amd import VecGridWorld, workloads
    n, T = 8192, 150
    tg = workloads.rt20(n, seed=12)

    def make():
        env = VecGridWorld(n, autoreset=True, size_reward=False, max_steps=60)
        env.set_tasks(tg.to(env.device))
        env.reset()
        return env
    whole, parts_env = make(), make()
    acts = whole.fill_actions(T, seed=4)
    for t in range(T):
        whole.step_walking_ptr(acts[t])
    torch.cuda.synchronize()
    subs = parts_env.split(4)
    m = n // 4
    chunks = [acts[:, k * m:(k + 1) * m].contiguous() for k in range(4)]
    torch.cuda.synchronize()
    for t in range(T):
        for k, sb in enumerate(subs):
            sb.step_walking_ptr(chunks[k][t])
    for sb in subs:
        sb.synchronize()
    for a, b in ((whole.grid_buf, parts_env.grid_buf), (whole.agent_buf, parts_env.agent_buf),
                 (whole.hist_buf, parts_env.hist_buf), (whole.reward, parts_env.reward),
                 (whole.agent_pos, parts_env.agent_pos)):
        assert torch.equal(a, b)


def test_step_loop_is_graph_capturable():
    """The C ABI never synchronises or allocates, so a K-step loop can be captured into a HIP graph and
    replayed; the replay produces the same bytes as eager launches."""
    from gridworld_amd import VecGridWorld, workloads
    n, K = 4096, 40
    tg = workloads.rt20(n, seed=21)

    def make():
        env = VecGridWorld(n, autoreset=True, size_reward=False, max_steps=30)
        env.set_tasks(tg.to(env.device))
        env.reset()
        return env
    eager, graphed = make(), make()
    acts = eager.fill_actions(K, seed=9)
    for t in range(K):
        eager.step_walking_ptr(acts[t])
    acts_g = graphed.fill_actions(K, seed=9)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    snap = graphed.state_dict()
    with torch.cuda.graph(g):
        for t in range(K):
            graphed.step_walking_ptr(acts_g[t])
    graphed.load_state_dict(snap)   # capture does not execute; start the replay from the same state
    graphed.stats_buf.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(eager.grid_buf, graphed.grid_buf) and torch.equal(eager.agent_buf, graphed.agent_buf)
    assert torch.equal(eager.hist_buf, graphed.hist_buf) and torch.equal(eager.reward, graphed.reward)


def test_shared_and_partially_updated_task_table():
    """Many envs share a small task table through env_task; rows can be replaced in place (set_tasks first=)."""
    from gridworld_amd import VecGridWorld, workloads
    from oracle import oracle as O
    n, ntasks, T = 600, 5, 80
    tg = workloads.rt20(ntasks + 2, seed=41).numpy()
    mapping = (np.arange(n) * 7 % ntasks).astype(np.int32)
    env = VecGridWorld(n, num_tasks=ntasks, size_reward=False, max_steps=500)
    env.set_tasks(tg[:ntasks], env_task=mapping)
    env.set_tasks(tg[ntasks:ntasks + 2], first=1)          # replace rows 1 and 2
    table = tg[:ntasks].copy()
    table[1:3] = tg[ntasks:ntasks + 2]
    env.reset()
    ob = O.OracleBatch(n, size_reward=False, max_steps=500)
    ob.set_tasks(table[mapping])
    ob.reset()
    acts = env.fill_actions(T, seed=2).cpu().numpy()
    for t in range(T):
        env.step(torch.as_tensor(acts[t]))
        ob.step_walking(acts[t], nthreads=8)
    torch.cuda.synchronize()
    assert np.array_equal(env.grid.cpu().numpy().reshape(n, -1), ob.grid)
    assert np.array_equal(env.reward.cpu().numpy(), ob.reward)
    assert np.array_equal(env.internals().view(np.uint64), ob.internals().view(np.uint64))


def test_sub_batches_inherit_sampling_and_order_after_parent():
    """split() right after set_tasks / reset with no host synchronisation in between: the sub-batch streams wait
    for the parent's queued work; with task sampling on, every sub-batch draws exactly what the whole batch
    draws (global env index in the RNG key), and the parent's stats() include the sub-batch counters."""
    from gridworld_amd import VecGridWorld, workloads
    n, T, ntasks = 4096, 90, 13
    tg = workloads.rt20(ntasks, seed=5)

    def make():
        env = VecGridWorld(n, num_tasks=ntasks, autoreset=True, size_reward=False, max_steps=20)
        env.set_tasks(tg.to(env.device), env_task=np.zeros(n, np.int32))
        env.set_task_sampling(True, seed=77)
        env.reset()
        return env
    whole = make()
    acts = whole.fill_actions(T, seed=6)
    for t in range(T):
        whole.step_walking_ptr(acts[t])
    parts_env = make()
    subs = parts_env.split(4)           # no synchronize: the new streams must order themselves after reset()
    m = n // 4
    for t in range(T):
        for k, sb in enumerate(subs):
            sb.step_walking_ptr(acts[t, k * m:(k + 1) * m])    # a view: recorded on the sub-batch stream
    for sb in subs:
        sb.join()
    torch.cuda.synchronize()
    assert torch.equal(whole.env_task, parts_env.env_task) and torch.equal(whole.episode, parts_env.episode)
    assert torch.equal(whole.grid_buf, parts_env.grid_buf) and torch.equal(whole.agent_buf, parts_env.agent_buf)
    assert len(torch.unique(parts_env.env_task)) == ntasks
    sw, sp = whole.stats(), parts_env.stats()
    assert sw['resets'] == sp['resets'] > 0 and sw['changed'] == sp['changed']


def test_sampling_advances_inside_a_replayed_graph():
    """The sampler's key is the env's episode counter in device memory, not a launch argument: replaying a
    captured step loop keeps drawing new tasks and stays equal to eager stepping."""
    from gridworld_amd import VecGridWorld, workloads
    n, K, ntasks = 2048, 25, 9
    tg = workloads.rt20(ntasks, seed=3)

    def make():
        env = VecGridWorld(n, num_tasks=ntasks, autoreset=True, size_reward=False, max_steps=10)
        env.set_tasks(tg.to(env.device), env_task=np.zeros(n, np.int32))
        env.set_task_sampling(True, seed=1)
        env.reset()
        return env
    eager, graphed = make(), make()
    acts = eager.fill_actions(K, seed=2)
    for rep in range(3):
        for t in range(K):
            eager.step_walking_ptr(acts[t])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    snap = graphed.state_dict()
    with torch.cuda.graph(g):
        for t in range(K):
            graphed.step_walking_ptr(acts[t])
    graphed.load_state_dict(snap)
    seen = []
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append(graphed.env_task.clone())
    assert torch.equal(eager.env_task, graphed.env_task) and torch.equal(eager.grid_buf, graphed.grid_buf)
    assert torch.equal(eager.episode, graphed.episode) and int(graphed.episode.min()) >= 1 + 3 * (K // 10)
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
