"""Pins the CPU oracle (oracle/igw_oracle.c) to the golden vectors recorded from the
Python reference (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

import golden_replay as GR
from oracle import oracle as O


@pytest.mark.parametrize('name', GR.WALK_FIXTURES + GR.FLY_FIXTURES + GR.DICT_FIXTURES)
def test_oracle_replays_fixture(name):
    fx = GR.load_fixture(name)
    n = GR.replay(fx, GR.OracleDriver(fx))
    assert n == fx['done'].size


@pytest.mark.parametrize('name', GR.SUBTASK_FIXTURES)
def test_oracle_replays_subtasks_episodes(name):
    """env.set_task_generator(Subtasks(...)) episodes recorded from the reference: start = turn k - 1, target =
    turn k, full_grid = final structure; GridWorld.max_int at every reset and the SizeReward it feeds
    (tasks/task.py:63-72, 260-286; env.py:227-241, 325-331)."""
    fx = GR.load_subtasks_fixture(name)
    # OracleBatch.reset copies the reset observation; the batch driver steps without auto-reset
    n = GR.replay_subtasks(fx, GR.OracleDriver(dict(targets=fx['full_grids'], kwargs=fx['kwargs'])))
    assert n == fx['done'].size


def test_oracle_env_max_int_and_syn_max_int():
    """GridWorld.max_int at reset (env.py:241, user task) and the synthetic task's max_int."""
    for name in ('s2_walk_cdm', 's3_walk_rt20'):
        fx = GR.load_fixture(name)
        E = len(fx['targets'])
        for e in range(0, E, 5):
            env = O.OracleEnv(**fx['kwargs'])
            env.set_task(fx['targets'][e], fx['starts'][e])
            env.reset()
            assert env.task_state()['env_max_int'] == fx['env_max_int'][e]
            for t in range(120):
                if fx['reset_before'][e, t]:
                    env.reset()
                env.step(int(fx['actions'][e, t]))
                assert env.task_state()['syn_max_int'] == fx['syn_max_int'][e, t]


def test_oracle_task_vectors():
    """Rotations, admissible sets, maximal/argmax intersection (tasks/task.py:47-72, 121-161)."""
    z = np.load(GR.GOLDEN_DIR + '/s6_task_vectors.npz')
    targets, grids, fulls = z['targets'], z['grids'], z['full_grids']
    for p in range(len(targets)):
        for g in range(len(grids)):
            r = O.task_eval(targets[p], grids[g])
            assert r['max_int'] == z['max_int'][p, g]
            assert tuple(r['argmax']) == tuple(z['argmax'][p, g])
            r2 = O.task_eval(targets[p], grids[g], invariant=False)
            assert r2['max_int'] == z['ni_max_int'][p, g]
            assert tuple(r2['argmax']) == tuple(z['ni_argmax'][p, g])
            r3 = O.task_eval(targets[p], grids[g], full_grid=fulls[p])
            assert r3['max_int'] == z['fg_max_int'][p, g]
            assert tuple(r3['argmax']) == tuple(z['fg_argmax'][p, g])
        assert r['target_size'] == z['target_size'][p]
        assert np.array_equal(r['adm_count'], z['adm_count'][p])
        assert np.array_equal(r['adm_mask'], z['adm_mask'][p])
        assert np.array_equal(r['rot'], z['rot'][p])
        assert np.array_equal(r2['adm_count'], z['ni_adm_count'][p])
        assert np.array_equal(r3['adm_count'], z['fg_adm_count'][p])
        assert np.array_equal(r3['adm_mask'], z['fg_adm_mask'][p])


def test_known_answers():
    """SURVEY.md Appendix B known answers (captured from the reference)."""
    two = np.zeros((9, 11, 11), np.int8)
    two[0, 5, 3] = two[0, 5, 4] = 1
    env = O.OracleEnv(size_reward=False)
    env.set_task(two, [])
    o = env.reset()
    assert o['agentPos'].tolist() == [0] * 5 and o['compass'][0] == 0 and o['inventory'].tolist() == [20] * 6
    ys = []
    for _ in range(4):
        env.step(0)
        ys.append((env.internal()[1], env.internal()[5]))
    assert ys == [(-0.037500000000000006, -1.0), (-0.125, -2.0), (-0.25, 0.0), (-0.25, 0.0)]
    env.step(1)
    assert tuple(env.internal()[:3]) == (1.5308084989341915e-17, -0.25, -0.25)
    env.step(4)
    assert tuple(env.internal()[:3]) == (0.25, -0.25, -0.25)
    env.step(5)
    assert env.internal()[1] == 0.058910161513775455 and env.internal()[5] == 5.928203230275509
    for _ in range(13):
        env.step(0)
    assert env.internal()[1] == -0.25 and env.internal()[5] == 0 and env.internal()[6] == 2
    for _ in range(9):
        env.step(14)
    rewards = [env.step(a)[1] for a in (17, 17, 16, 16, 8)]
    assert rewards == [1, 0, -1, 0, -0.1]
    o = env.obs()
    assert o['grid'][0, 5, 4] == 3 and o['inventory'].tolist() == [20, 20, 19, 20, 20, 20]
    assert o['agentPos'].tolist() == [0.25, -0.25, -0.25, -45, 0] and o['compass'][0] == -180
