"""The public stepping API (gridworld_amd.VecGridWorld): the zero-conversion fast path of step(), the captured step loop
(capture_steps / StepGraph.replay) and the packed records behind the observation views (include/igw.h, ABI 4)."""
import numpy as np
import pytest
import torch

import golden_replay as GR

pytestmark = pytest.mark.gpu

STATE = ('grid_buf', 'occ_buf', 'hist_buf', 'agent_buf', 'aux_buf', 'out_buf')


def _env(n, mode='walking', **kw):
    from gridworld_amd import VecGridWorld, workloads
    env = VecGridWorld(n, action_space=mode, autoreset=True, size_reward=False, max_steps=40, **kw)
    env.set_tasks(workloads.rt20(n, seed=5).to(env.device))
    env.reset()
    return env


@pytest.mark.parametrize('n', [4096, 1000])
def test_graph_replayed_api_stepping_is_byte_identical_to_eager(n):
    """`for t: env.step(actions[t])` eagerly == ONE replay of env.capture_steps(actions), auto-resets inside, twice
    over (the second replay reads refilled buffers): every state and output byte, and the per-step records."""
    T = 90
    eager, graphed = _env(n), _env(n)
    acts = eager.fill_actions(2 * T, seed=9)
    buf = acts[:T].clone()
    g = graphed.capture_steps(buf, record=True)
    for half in range(2):
        buf.copy_(acts[half * T:(half + 1) * T])
        rewards, dones = [], []
        for t in range(T):
            obs, r, d, info = eager.step(acts[half * T + t])
            assert info == {} and obs['grid'].shape == (n, 9, 11, 11) and obs['compass'].shape == (n, 1)
            rewards.append(r.clone()); dones.append(d.clone())
        obs_g, r_g, d_g, _ = g.replay()
        torch.cuda.synchronize()
        for k in STATE:
            assert torch.equal(getattr(eager, k), getattr(graphed, k)), (half, k)
        assert torch.equal(torch.stack(rewards), g.rewards) and torch.equal(torch.stack(dones), g.dones)
        assert torch.equal(obs_g['agentPos'], obs['agentPos']) and torch.equal(r_g, r) and torch.equal(d_g, d)
    assert eager.stats()['resets'] == graphed.stats()['resets'] > 0


def test_graph_replay_flying_and_dict_action_spaces():
    from gridworld_amd import VecGridWorld, workloads
    n, T = 512, 60
    gen = torch.Generator(device='cuda'); gen.manual_seed(3)
    for mode, kw in (('flying', {}), ('walking', dict(discretize=False))):
        a, b = _env(n, mode, **kw), _env(n, mode, **kw)
        if mode == 'flying':
            acts = dict(movement=torch.empty((T, n, 3), device='cuda').uniform_(-1, 1, generator=gen),
                        camera=torch.empty((T, n, 2), device='cuda').uniform_(-5, 5, generator=gen),
                        inventory=torch.randint(0, 7, (T, n), device='cuda', dtype=torch.int32, generator=gen),
                        placement=torch.randint(0, 3, (T, n), device='cuda', dtype=torch.int32, generator=gen))
        else:
            btn = (torch.rand((T, n, 8), device='cuda', generator=gen) < 0.2).to(torch.uint8)
            btn[:, :, 7] = torch.randint(0, 7, (T, n), device='cuda', generator=gen).to(torch.uint8)
            acts = dict(buttons=btn, camera=torch.empty((T, n, 2), device='cuda').uniform_(-5, 5, generator=gen))
        g = b.capture_steps(acts)
        for t in range(T):
            a.step({k: v[t] for k, v in acts.items()})
        g.replay()
        torch.cuda.synchronize()
        for k in STATE:
            assert torch.equal(getattr(a, k), getattr(b, k)), (mode, k)
    with pytest.raises(ValueError):
        a.capture_steps(dict(buttons=btn.cpu(), camera=acts['camera']))     # host buffers cannot be captured


def test_step_fast_path_and_converted_inputs_agree():
    """int32 device tensors go straight to the C ABI; lists, numpy arrays, int64 / host tensors are converted -- same result."""
    n = 300
    envs = [_env(n) for _ in range(4)]
    acts = envs[0].fill_actions(30, seed=4)
    for t in range(30):
        a = acts[t]
        envs[0].step(a)
        envs[1].step(a.cpu().numpy())
        envs[2].step(a.long())
        envs[3].step(a.cpu().tolist())
    torch.cuda.synchronize()
    for e in envs[1:]:
        for k in STATE:
            assert torch.equal(getattr(envs[0], k), getattr(e, k)), k
    obs, r, d, _ = envs[0].step(acts[0])
    # the returned tensors are views of the kernels' output record: nothing is copied per step
    assert obs['agentPos'].data_ptr() == envs[0].out_buf.data_ptr() and r.data_ptr() == envs[0].out_buf.data_ptr() + 48
    assert d.data_ptr() == envs[0].out_buf.data_ptr() + 52 and obs['inventory'].data_ptr() == envs[0].out_buf.data_ptr() + 20
    assert obs['grid'].data_ptr() == envs[0].grid_buf.data_ptr()


@pytest.mark.parametrize('gs', [0, 4, 1])
def test_wide_inventory_state_and_rejected_placements(gs):
    """Starting grids with hundreds of blocks of one colour (fixture s12, recorded from the reference): the int16
    inventory of the agent record and the task's inv_init hold the reference's values (-222 ... -1069), a placement of
    such a colour never succeeds (core/world.py:317 needs inventory > 0) and every break refunds one."""
    from hip_driver import HipDriver
    fx = GR.load_fixture('s12_wide_inventory')
    drv = HipDriver(fx, lanes_per_env=gs)
    drv.set_tasks(fx['targets'], fx['starts'], invariant=True)
    drv.set_initial_pose(fx['init_pose'])
    drv.reset(None)
    env = drv.env
    torch.cuda.synchronize()
    want = fx['reset_inventory'].astype(np.int64)
    assert want.min() == -1069 and (want < -128).any(1).sum() >= 7
    assert np.array_equal(env.task_state()['inventory'], want)
    meta = env.task_meta.cpu().numpy()
    assert np.array_equal(meta[:, 64:76].copy().view(np.int16).astype(np.int64), want)
    assert np.array_equal(env.inventory.cpu().numpy(), fx['reset_inventory'])
    E, T = fx['done'].shape
    for t in range(150):    # the first episode
        before = env.task_state()['inventory']
        grid_before = env.grid.cpu().numpy().reshape(E, -1).copy()
        drv.step_walking(fx['actions'][:, t])
        torch.cuda.synchronize()
        if fx['done'][:, t].any():
            break
        after, grid_after = env.task_state()['inventory'], env.grid.cpu().numpy().reshape(E, -1)
        assert np.array_equal(after.astype(np.float32), fx['inventory'][:, t])
        placed = (grid_after != 0) & (grid_before == 0)
        for e, c in zip(*np.nonzero(placed)):
            col = int(grid_after[e, c])
            assert before[e, col - 1] > 0 and after[e, col - 1] == before[e, col - 1] - 1


@pytest.mark.parametrize('mode,chains', [('walking', 2), ('walking', 4), ('flying', 2), ('walking_dict', 4)])
def test_step_graph_as_independent_chains_equals_the_eager_loop(mode, chains):
    """capture_steps(chains=P): the T steps as P parallel chains of sub-batch launches -- same bytes as T eager
    env.step() calls (state, records, per-step outputs with record=True, device counters), auto-resets and the
    on-device task sampler included."""
    from gridworld_amd import IgwError, VecGridWorld, workloads
    n, T = 4096, 70
    space = dict(action_space='flying') if mode == 'flying' else dict(discretize=False) if mode == 'walking_dict' else {}
    kw = dict(size_reward=False, max_steps=25, autoreset=True, num_tasks=37, **space)
    tg = workloads.rt20(37, seed=5)
    a, b = VecGridWorld(n, **kw), VecGridWorld(n, **kw)
    for env in (a, b):
        env.set_tasks(tg.to(env.device), env_task=np.zeros(n, np.int32))
        env.set_task_sampling(True, seed=3)
        env.reset()
    g = torch.Generator(device=a.device)
    g.manual_seed(1)
    dev = a.device
    if mode == 'walking':
        acts = torch.randint(0, 18, (T, n), generator=g, device=dev, dtype=torch.int32)
        step_t = lambda t: acts[t]  # noqa: E731
    elif mode == 'flying':
        acts = dict(movement=torch.rand((T, n, 3), generator=g, device=dev) * 2 - 1, camera=torch.rand((T, n, 2), generator=g, device=dev) * 10 - 5,
                    inventory=torch.randint(0, 7, (T, n), generator=g, device=dev, dtype=torch.int32),
                    placement=torch.randint(0, 3, (T, n), generator=g, device=dev, dtype=torch.int32))
        step_t = lambda t: {k: v[t] for k, v in acts.items()}  # noqa: E731
    else:
        acts = dict(buttons=(torch.rand((T, n, 8), generator=g, device=dev) < 0.3).to(torch.uint8),
                    camera=torch.rand((T, n, 2), generator=g, device=dev) * 10 - 5)
        acts['buttons'][:, :, 7] = torch.randint(0, 7, (T, n), generator=g, device=dev).to(torch.uint8)
        step_t = lambda t: {k: v[t] for k, v in acts.items()}  # noqa: E731
    graph = a.capture_steps(acts, record=True, chains=chains)
    assert graph.chains == chains and len(graph.subs) == chains
    graph.replay()
    outs = []
    for t in range(T):
        b.step(step_t(t))
        outs.append(b.out_buf.clone())
    torch.cuda.synchronize()
    assert torch.equal(graph.outs, torch.stack(outs))
    for name in ('grid_buf', 'occ_buf', 'hist_buf', 'agent_buf', 'aux_buf', 'out_buf'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    sa, sb = a.stats(), b.stats()
    assert sa == sb and sa['steps'] == n * T and sa['resets'] > 0
    a.set_task_sampling(True, seed=4)         # the settings are kernel parameters of the captured launches
    with pytest.raises(IgwError):
        graph.replay()
    with pytest.raises(ValueError):
        a.capture_steps(acts, chains=3)


def test_step_graph_refuses_to_replay_a_stale_configuration():
    """A captured launch carries the kernel parameters by value: after the episode log is switched on or off (or a
    sampler is reconfigured) replay() raises instead of running the old settings -- or writing into the log buffers
    that were freed (the graph also keeps the buffers of its capture alive)."""
    from gridworld_amd import IgwError, VecGridWorld, workloads
    n, T = 256, 8
    env = VecGridWorld(n, size_reward=False, autoreset=True, max_steps=5)
    env.set_tasks(workloads.rt20(n, seed=1).to(env.device))
    env.reset()
    acts = env.fill_actions(T, seed=2)
    rec, heads = env.enable_trajectory_log(4)
    g1 = env.capture_steps(acts)
    g1.replay()
    torch.cuda.synchronize()
    assert int(heads[:, :, 1].max()) > 0 and g1._held[0] is rec
    env.disable_trajectory_log()
    del rec, heads
    with pytest.raises(IgwError):
        g1.replay()
    g2 = env.capture_steps(acts)
    g2.replay()
    env.set_random_tasks(True, seed=1)
    with pytest.raises(IgwError):
        g2.replay()
    d = env.dense()
    assert all(v.is_contiguous() for v in d.values()) and d['done'].shape == (n,) and d['reward'].stride() == (1,)
    assert env.reward.stride() == (16,) and env.done.stride() == (64,)   # the documented strides of the views


def test_host_resident_records_equal_device_resident():
    """VecGridWorld(host_records=True): output / agent / aux records and the grid in pinned host memory that the kernels
    read and write across PCIe (what the 1-env facade runs on) -- the same bytes as the device-resident batch, step by
    step, auto-resets included; the observation tensors are CPU tensors that alias that memory."""
    from gridworld_amd import VecGridWorld, workloads
    n, T = 200, 90
    kw = dict(size_reward=True, max_steps=30, autoreset=True)
    tg = workloads.rt20(n, seed=9)
    a, b = VecGridWorld(n, host_records=True, **kw), VecGridWorld(n, **kw)
    assert not a.out_buf.is_cuda and a.host_view.is_pinned() and b.out_buf.is_cuda
    for env in (a, b):
        env.set_tasks(tg.to(env.device))
        env.reset()
    acts = b.fill_actions(T, seed=4)
    for t in range(T):
        oa, ra, da, _ = a.step(acts[t])
        ob, rb, db, _ = b.step(acts[t])
        torch.cuda.synchronize()
        assert torch.equal(a.out_buf, b.out_buf.cpu()) and torch.equal(a.agent_buf, b.agent_buf.cpu()), t
        assert torch.equal(oa['grid'], ob['grid'].cpu()) and torch.equal(ra, rb.cpu()) and torch.equal(da, db.cpu())
    assert torch.equal(a.aux_buf, b.aux_buf.cpu()) and torch.equal(a.hist_buf, b.hist_buf) and torch.equal(a.occ_buf, b.occ_buf)
    sa, sb = a.stats(), b.stats()
    assert sa == sb and sa['steps'] == n * T and sa['resets'] > 0
    assert np.array_equal(a.internals().view(np.uint64), b.internals().view(np.uint64))
