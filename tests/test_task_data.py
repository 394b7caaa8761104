"""Host Task attributes that are pure data layout (no GPU): rotations, admissible translation lists, sparse /
dense conversion -- against the reference-recorded vectors of s6_task_vectors.npz (tests/golden/gen_golden.py)."""
import os

import numpy as np

import golden_replay as GR


def _z():
    return np.load(GR.GOLDEN_DIR + '/s6_task_vectors.npz')


def _mask(adm):
    m = np.zeros((21, 21), np.uint8)
    for dx, dz in adm:
        m[dx + 10, dz + 10] = 1
    return m


def test_rotations_and_admissible_lists_match_reference():
    from gridworld_amd.tasks import Task
    z = _z()
    for p in range(len(z['targets'])):
        t = Task('', z['targets'][p].astype(np.int32), starting_grid=[])
        assert t.target_size == z['target_size'][p]
        for r in range(4):
            assert np.array_equal(np.asarray(t.target_grids[r], np.int8), z['rot'][p, r]), (p, r)
            assert len(t.admissible[r]) == z['adm_count'][p, r]
            assert np.array_equal(_mask(t.admissible[r]), z['adm_mask'][p, r])
            assert t.admissible[r] == sorted(t.admissible[r])  # the reference appends in (dx, dz) order
        f = Task('', z['targets'][p].astype(np.int32), starting_grid=[], full_grid=z['full_grids'][p].astype(np.int32))
        for r in range(4):
            assert len(f.admissible[r]) == z['fg_adm_count'][p, r]
            assert np.array_equal(_mask(f.admissible[r]), z['fg_adm_mask'][p, r])
        ni = Task('', z['targets'][p].astype(np.int32), starting_grid=[], invariant=False)
        assert ni.admissible == [[(0, 0)]] and z['ni_adm_count'][p, 0] == 1


def test_sparse_dense_round_trip_and_reset_without_device():
    from gridworld_amd.tasks import Task, Tasks
    z = _z()
    d = z['targets'][5].astype(np.int32)
    sp = Tasks.dense_to_sparse(d)
    assert np.array_equal(Tasks.to_dense(sp), d)
    # the reference's own to_sparse(ndarray) (task.py:178-187), bit for bit incl. its index mix-up
    z10 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 's10_task_protocol.npz'))
    got = Tasks.to_sparse(z10['to_sparse_in'].astype(np.int32))
    assert [tuple(int(v) for v in b) for b in got] == [tuple(r) for r in z10['to_sparse_out'].tolist()]
    assert Tasks.to_sparse(sp) is sp and Tasks.to_dense(d) is d  # pass-through (task.py:169-170, 179)
    assert np.array_equal(Tasks.to_dense(None), np.zeros((9, 11, 11)))
    # building and resetting a task needs no GPU; only reading an intersection does
    t = Task('chat', d, starting_grid=[], last_instruction='do it')
    assert t.reset() is t and t.prev_grid_size == 0 and t.max_int == 0
    assert len(t) == 1 and list(t) == [t] and 'do it' in repr(t)
    t2 = Task('', d, starting_grid=sp[:3]).reset()
    assert t2.prev_grid_size == 3


def test_random_tasks_dump_load(tmp_path):
    from gridworld_amd.tasks import RandomTasks
    np.random.seed(5)
    a = RandomTasks(max_blocks=5, num_colors=3, max_cache=4)
    a.dump(tmp_path / 'tasks.pkl')
    b = RandomTasks(max_blocks=5, num_colors=3, max_cache=0)
    b.load(tmp_path / 'tasks.pkl')
    assert list(a.tasks) == list(b.tasks)
    for uid in a.tasks:
        assert np.array_equal(a.tasks[uid].target_grid, b.tasks[uid].target_grid)
    assert b.set_task(next(iter(b.tasks))) is b.current
