"""Host-side task generators (gridworld_amd.tasks) against the reference's own sampling under the same
np.random seed (tests/golden/gen_generators.py): RandomTasks / CustomTasks / Subtasks consume the global
NumPy stream exactly like gridworld/tasks/task_set.py and gridworld/tasks/task.py:208-308."""
import json

import numpy as np
import pytest

import golden_replay as GR


def _fx():
    return np.load(GR.GOLDEN_DIR + '/s7_generators.npz')


def _sparse(lst):
    return [tuple(int(v) for v in b) for b in lst]


def make_generator(G, z, which):
    spec = json.loads(str(z[which + '_spec']))
    if which == 'subtasks':
        return G.Subtasks(spec['dialog'], [_sparse(s) for s in spec['seq']])
    return G.CustomTasks([(c, _sparse(b)) for c, b in spec['goals']], task_kwargs={'starting_grid': []})


@pytest.mark.parametrize('tag', ['random_a', 'random_b'])
def test_random_tasks_sampling_stream(tag):
    import gridworld_amd.tasks as T
    z = _fx()
    kw = json.loads(str(z[tag + '_spec']))
    np.random.set_state(('MT19937', z[tag + '_state0'], int(z[tag + '_pos0']), 0, 0.0))
    gen = T.RandomTasks(**kw)
    got = np.stack([np.asarray(gen.reset().target_grid, np.int8) for _ in range(40)])
    assert np.array_equal(got, z[tag + '_targets'])


@pytest.mark.parametrize('which', ['subtasks', 'custom'])
def test_generator_task_sequence_without_device(which):
    """Only the sampling: the sequence of (target, start, full grid, chat) the generator hands to reset()."""
    import gridworld_amd.tasks as T

    class G:  # the tasks module is enough here
        Subtasks, CustomTasks = T.Subtasks, T.CustomTasks
    z = _fx()
    np.random.seed(int(z[which + '_seed']))
    gen = make_generator(G, z, which)
    n = len(z[which + '_task_targets'])
    gen.reset()  # set_task_generator() calls reset() once before the user's first reset()
    for i in range(n):
        t = gen.reset()
        t.reset()
        assert np.array_equal(np.asarray(t.target_grid, np.int8), z[which + '_task_targets'][i]), i
        assert np.array_equal(T.Tasks.to_dense(t.starting_grid).astype(np.int8), z[which + '_task_starts'][i]), i
        if t.full_grid is not None:
            assert np.array_equal(np.asarray(t.full_grid, np.int8), z[which + '_task_fulls'][i]), i
        assert t.chat == str(z[which + '_task_chats'][i])


@pytest.mark.gpu
@pytest.mark.parametrize('which', ['subtasks', 'custom'])
def test_generator_driven_episodes_on_device(which):
    """set_task_generator -> reset -> step ... with resets on done, through the facade on the GPU."""
    import gridworld_amd as G
    z = _fx()
    np.random.seed(int(z[which + '_seed']))
    gen = make_generator(G, z, which)
    env = G.make('IGLUGridworldVector-v0', size_reward=False, max_steps=100)
    env.set_task_generator(gen)
    env.reset()
    acts = z[which + '_actions']
    k = 0
    assert np.array_equal(np.asarray(env.task.target_grid, np.int8), z[which + '_task_targets'][k])
    for t in range(len(acts)):
        if z[which + '_reset_before'][t]:
            env.reset()
            k += 1
            assert np.array_equal(np.asarray(env.task.target_grid, np.int8), z[which + '_task_targets'][k])
        obs, r, d, _ = env.step(int(acts[t]))
        assert d == bool(z[which + '_done'][t]), t
        assert np.float32(r) == np.float32(z[which + '_reward'][t]), t
        assert np.array_equal(obs['inventory'], z[which + '_inventory'][t]), t
        assert np.array_equal(obs['agentPos'].view(np.uint32), z[which + '_agentPos'][t].view(np.uint32)), t
    assert np.array_equal(obs['grid'].astype(np.int8), z[which + '_grid_final'])
