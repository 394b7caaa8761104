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
