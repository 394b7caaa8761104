"""The gym-shaped 1-env facade (gridworld_amd.make) replays reference trajectories with the
reference's own calling sequence: make -> set_task -> reset -> step, numpy dict observations."""
import numpy as np
import pytest

import golden_replay as GR

pytestmark = pytest.mark.gpu


def _sparse(dense):
    from gridworld_amd import Tasks
    return Tasks.dense_to_sparse(np.asarray(dense, np.int32))


@pytest.mark.parametrize('name,envs', [('s5_scripted', range(9)), ('s2_walk_cdm', [0, 7]),
                                       ('s1_walk_dummy_sizereward', [1]), ('s4_fly_rt20', [0]),
                                       # inventories of -222, -1069 and -129 at reset (env.py:243-246)
                                       ('s12_wide_inventory', [0, 2, 6]), ('s12_wide_inventory_sizereward', [1])])
def test_facade_matches_reference(name, envs):
    import gridworld_amd as G
    fx = GR.load_fixture(name)
    for e in envs:
        env = G.make('IGLUGridworld-v0', vector_state=True, render=False, **fx['kwargs'])
        task = G.Task('chat', fx['targets'][e].astype(np.int32), starting_grid=_sparse(fx['starts'][e]),
                      **fx['task_kwargs'])
        env.set_task(task)
        if 'init_pose' in fx:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                env.initialize_world(_sparse(fx['starts'][e]), [float(v) for v in fx['init_pose'][e]])
        obs = env.reset()
        assert set(obs) == {'inventory', 'compass', 'dialog', 'grid', 'agentPos'}
        assert obs['grid'].dtype == np.int32 and obs['grid'].shape == (9, 11, 11)
        assert obs['agentPos'].dtype == np.float32 and obs['inventory'].dtype == np.float32
        assert obs['compass'].shape == (1,) and obs['dialog'] == 'chat'
        assert np.array_equal(obs['inventory'], fx['reset_inventory'][e])
        T = min(fx['done'].shape[1], 150)
        for t in range(T):
            if fx['reset_before'][e, t]:
                env.reset()
            if fx['flying']:
                a = {'movement': fx['act_movement'][e, t], 'camera': fx['act_camera'][e, t],
                     'inventory': int(fx['act_inventory'][e, t]), 'placement': int(fx['act_placement'][e, t])}
            else:
                a = int(fx['actions'][e, t])
            obs, reward, done, info = env.step(a)
            assert done == bool(fx['done'][e, t]), (name, e, t)
            # the reference's double, not a float32 rounding of it (-0.1, not -0.10000000149011612)
            assert reward == fx['reward'][e, t] and isinstance(reward, (int, float)), (name, e, t, reward)
            assert np.array_equal(obs['inventory'], fx['inventory'][e, t])
            # float32 observations bit-exact, flying included (A-fly contract)
            assert np.array_equal(obs['agentPos'].view(np.uint32), fx['agentPos'][e, t].view(np.uint32)), (name, e, t)
            assert obs['compass'][0] == fx['compass'][e, t]


@pytest.mark.parametrize('name', GR.SUBTASK_FIXTURES)
def test_facade_subtasks_generator_matches_reference(name):
    """The reference's own calling sequence with a task generator: env.set_task_generator(Subtasks(dialog,
    structure_seq)); env.reset() draws the turn from np.random exactly as the reference does
    (tasks/task.py:225-243), so with the recorded seed the port's Subtasks walks through the same episodes:
    turns, GridWorld.max_int at every reset, observations, rewards and dones."""
    import json
    import gridworld_amd as G
    fx = GR.load_subtasks_fixture(name)
    specs = json.loads(str(fx['specs']))
    for e in (0, 3, 5, 10, 12):
        np.random.seed(int(fx['np_seed'][e]))
        env = G.make('IGLUGridworld-v0', vector_state=True, render=False, **fx['kwargs'])
        seq = [[tuple(b) for b in turn] for turn in specs[e]['seq']]
        st = G.Subtasks(specs[e]['dialog'], seq)
        env.set_task_generator(st)
        ep = -1
        done = True
        for t in range(fx['done'].shape[1]):
            if done:
                obs = env.reset()
                ep += 1
                assert (st.task_start, st.task_goal) == tuple(fx['ep_turn'][e, ep]), (e, ep)
                assert env.unwrapped.max_int == fx['ep_env_max_int'][e, ep], (e, ep)
                assert np.array_equal(obs['grid'], fx['ep_starts'][e, ep])
                assert np.array_equal(obs['inventory'], fx['ep_reset_inventory'][e, ep])
                assert np.array_equal(env.task.target_grid, fx['ep_targets'][e, ep])
            assert bool(fx['reset_before'][e, t]) == (done and t > 0)
            obs, reward, done, _ = env.step(int(fx['actions'][e, t]))
            assert done == bool(fx['done'][e, t]) and reward == fx['reward'][e, t], (name, e, t, reward)
            assert np.array_equal(obs['agentPos'].view(np.uint32), fx['agentPos'][e, t].view(np.uint32)), (name, e, t)
            assert np.array_equal(obs['inventory'], fx['inventory'][e, t])
        assert ep + 1 == fx['n_episodes'][e]


def test_facade_errors_and_spaces():
    import gridworld_amd as G
    env = G.make('IGLUGridworldVector-v0')
    assert env.action_space.n == 18 and 0 <= env.action_space.sample() < 18
    with pytest.raises(ValueError):
        env.reset()
    with pytest.raises(ValueError):
        env.step(0)
    with pytest.raises(NotImplementedError):
        G.make('IGLUGridworld-v0')  # render=True default: renderer is out of scope
    env.set_task(G.dummy_task())     # DUMMY_TASK works here (starting_grid None == [])
    obs = env.reset()
    obs, r, d, _ = env.step(env.action_space.sample())
    assert obs['grid'].sum() >= 0 and isinstance(r, float) and isinstance(d, bool)
    fly = G.make('IGLUGridworldVector-v0', action_space='flying')
    fly.set_task(G.dummy_task())
    with pytest.raises(ValueError):
        fly.step({'movement': [0, 0, 0], 'camera': [0, 0], 'inventory': 9, 'placement': 0})


def test_task_generator_resamples_on_reset():
    import gridworld_amd as G
    np.random.seed(0)
    gen = G.RandomTasks(max_blocks=5, num_colors=3, max_dist=2)
    env = G.make('IGLUGridworldVector-v0', size_reward=False)
    env.set_task_generator(gen)
    seen = set()
    for _ in range(4):
        env.reset()
        seen.add(env.task.target_grid.tobytes())
        assert env.task.target_size == 5
    assert len(seen) > 1


def test_target_in_obs():
    """obs['target_grid'] is the task's target as int32 at reset and at every step (env.py:255-256, 297-298)."""
    import gridworld_amd as G
    fx = GR.load_fixture('s2_walk_cdm')
    env = G.make('IGLUGridworldVector-v0', target_in_obs=True, **fx['kwargs'])
    assert env.observation_space['target_grid'].shape == (9, 11, 11)
    tgt = fx['targets'][3].astype(np.int32)
    env.set_task(G.Task('', tgt, starting_grid=_sparse(fx['starts'][3])))
    obs = env.reset()
    assert obs['target_grid'].dtype == np.int32 and np.array_equal(obs['target_grid'], tgt)
    for t in range(5):
        obs, *_ = env.step(int(fx['actions'][3, t]))
        assert np.array_equal(obs['target_grid'], tgt) and obs['target_grid'] is not tgt
    plain = G.make('IGLUGridworldVector-v0', **fx['kwargs'])
    plain.set_task(G.Task('', tgt, starting_grid=[]))
    assert 'target_grid' not in plain.reset()


def test_initialize_then_deinitialize_world():
    """initialize_world overrides the starting grid and pose (fixture s5_init_pose was recorded that way);
    deinitialize_world restores the task's own start and the default pose (env.py:177-193): after it the env
    replays the plain fixture bit for bit."""
    import gridworld_amd as G
    fp, f1 = GR.load_fixture('s5_init_pose'), GR.load_fixture('s3_walk_rt20')
    env = G.make('IGLUGridworldVector-v0', **fp['kwargs'])
    e = 1
    env.set_task(G.Task('', fp['targets'][e].astype(np.int32), starting_grid=[]))
    with pytest.warns(UserWarning):
        env.initialize_world(_sparse(fp['starts'][e]), [float(v) for v in fp['init_pose'][e]])
    assert env.initial_position == tuple(float(v) for v in fp['init_pose'][e][:3])
    env.reset()
    for t in range(60):
        obs, reward, done, _ = env.step(int(fp['actions'][e, t]))
        assert np.array_equal(obs['agentPos'].view(np.uint32), fp['agentPos'][e, t].view(np.uint32)), t
        assert reward == fp['reward'][e, t] and done == bool(fp['done'][e, t])
    env.deinitialize_world()
    assert env.initial_position == (0, 0, 0) and env.initial_rotation == (0, 0) and env._overwrite_starting_grid is None
    env.set_task(G.Task('', f1['targets'][e].astype(np.int32), starting_grid=_sparse(f1['starts'][e])))
    obs = env.reset()
    assert np.array_equal(obs['grid'], f1['reset_grid'][e].reshape(9, 11, 11))
    for t in range(80):
        if f1['reset_before'][e, t]:
            env.reset()
        obs, reward, done, _ = env.step(int(f1['actions'][e, t]))
        assert np.array_equal(obs['agentPos'].view(np.uint32), f1['agentPos'][e, t].view(np.uint32)), t
        assert reward == f1['reward'][e, t] and done == bool(f1['done'][e, t])


def test_actions_wrapper():
    """Actions (wrappers.py:11-32): Discrete(17), index i -> Discrete(18) action i, `place` (17) dropped."""
    import gridworld_amd as G
    from gridworld_amd.wrappers import Actions
    fx = GR.load_fixture('s1_walk_dummy')
    env = Actions(G.make('IGLUGridworldVector-v0', **fx['kwargs']))
    assert env.action_space.n == 17 and env.action_map == list(range(17))
    env.set_task(G.Task('', fx['targets'][0].astype(np.int32), starting_grid=[], **fx['task_kwargs']))   # pass-through attribute
    env.reset()
    acts = [int(a) for a in fx['actions'][0] if a != 17][:40]
    ref = G.make('IGLUGridworldVector-v0', **fx['kwargs'])
    ref.set_task(G.Task('', fx['targets'][0].astype(np.int32), starting_grid=[], **fx['task_kwargs']))
    ref.reset()
    for a in acts:
        o1, r1, d1, _ = env.step(a)
        o2, r2, d2, _ = ref.step(a)
        assert r1 == r2 and d1 == d2 and np.array_equal(o1['agentPos'], o2['agentPos'])
    assert env.unwrapped is env.env and env.max_steps == ref.max_steps


def test_public_attribute_surface_of_the_reference_env():
    """GridWorld's public attributes (gridworld/env.py:32-38) and the wrapper structure (env.py:306-331): `unwrapped`
    is the GridWorld inside SizeReward; step_no, grid, agent.{position, rotation, dy, time_int_steps, active_block,
    inventory} and world.placed follow the device state step by step (float64 internals of the fixture, bit for
    bit); the reference's own Logged wrapper reads unwrapped.step_no (wrappers.py:99)."""
    import gridworld_amd as G
    fx = GR.load_fixture('s2_walk_cdm_sizereward')
    e = 2
    env = G.make('IGLUGridworld-v0', vector_state=True, render=False, **fx['kwargs'])
    assert isinstance(env, G.SizeReward) and isinstance(env.unwrapped, G.GridWorld) and env.unwrapped is env.env
    assert env.unwrapped.unwrapped is env.unwrapped and env.size == 0
    inner = env.unwrapped
    assert inner.agent.inventory == [20] * 6 and inner.agent.active_block == 1 and inner.agent.time_int_steps == 2
    assert inner.step_no == 0 and inner.grid.shape == (9, 11, 11) and inner.grid.dtype == np.int32 and not inner.grid.any()
    env.set_task(G.Task('', fx['targets'][e].astype(np.int32), starting_grid=_sparse(fx['starts'][e])))
    obs = env.reset()
    assert np.array_equal(inner.grid, obs['grid']) and inner.step_no == 0 and env.max_steps == fx['kwargs']['max_steps']
    assert inner.agent.position == (0.0, 0.0, 0.0) and inner.agent.rotation == (0.0, 0.0)
    assert inner.world.placed == {(int(x) - 5, int(y) - 1, int(z) - 5) for y, x, z in zip(*np.nonzero(obs['grid']))}
    w = inner.world.world
    assert w[(0, -2, 0)] == -1 and w[(18, -2, -18)] == 0 and len(w) == 37 * 37 + len(inner.world.placed)
    steps = 0
    for t in range(120):
        if fx['reset_before'][e, t]:
            env.reset()
            steps = 0
            assert env.size == 0
        obs, reward, done, _ = env.step(int(fx['actions'][e, t]))
        steps += 1
        it = fx['internal'][e, t]
        a = inner.agent
        got = np.array([*a.position, *a.rotation, a.dy, a.time_int_steps, a.active_block], np.float64)
        assert np.array_equal(got.view(np.uint64), it.view(np.uint64)), t
        assert inner.step_no == steps and np.array_equal(inner.grid, obs['grid'])
        assert a.inventory == [int(v) for v in obs['inventory']] and reward == fx['reward'][e, t]
        assert env.size == max(env.size, inner.max_int)
    # set_task goes to GridWorld.reset through the wrapper's attribute pass-through: SizeReward.size survives it
    # (env.py:155-166 vs 321-323), reset() through the wrapper clears it
    env.size = 7
    env.set_task(G.Task('', fx['targets'][e].astype(np.int32), starting_grid=[]))
    assert env.size == 7
    env.reset()
    assert env.size == 0
    # GridWorld's own constructor defaults differ from create_env's (env.py:27-31 vs 333-338)
    raw = G.GridWorld(render=False)
    assert raw.select_and_place is False and raw.discretize is False and raw.vector_state is True
    assert set(raw.action_space.keys()) == {'forward', 'back', 'left', 'right', 'jump', 'attack', 'use', 'camera', 'hotbar'}
