"""The gym-shaped 1-env facade (gridworld_amd.make) replays reference trajectories with the
reference's own calling sequence: make -> set_task -> reset -> step, numpy dict observations."""
import numpy as np
import pytest

import golden_replay as GR

pytestmark = pytest.mark.gpu


def _sparse(dense):
    from gridworld_amd import Tasks
    return Tasks.to_sparse(np.asarray(dense, np.int32))


@pytest.mark.parametrize('name,envs', [('s5_scripted', range(9)), ('s2_walk_cdm', [0, 7]),
                                       ('s1_walk_dummy_sizereward', [1]), ('s4_fly_rt20', [0])])
def test_facade_matches_reference(name, envs):
    import gridworld_amd as G
    fx = GR.load_fixture(name)
    for e in envs:
        env = G.make('IGLUGridworld-v0', vector_state=True, render=False, **fx['kwargs'])
        task = G.Task('chat', fx['targets'][e].astype(np.int32), starting_grid=_sparse(fx['starts'][e]),
                      **fx['task_kwargs'])
        env.set_task(task)
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
            assert np.float32(reward) == np.float32(fx['reward'][e, t]), (name, e, t)
            assert np.array_equal(obs['inventory'], fx['inventory'][e, t])
            if not fx['flying']:
                assert np.array_equal(obs['agentPos'].view(np.uint32), fx['agentPos'][e, t].view(np.uint32))
                assert obs['compass'][0] == fx['compass'][e, t]


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
