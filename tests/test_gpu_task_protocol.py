"""The host Task protocol on the device evaluator (igw_task_eval) against reference-recorded vectors
(tests/golden/gen_task_protocol.py -> s10_task_protocol.npz; s6_task_vectors.npz): maximal / argmax /
get_intersection, Task.reset + step_intersection (stale cache included), Subtasks progressive goals
(gridworld/tasks/task.py:74-161, 288-298)."""
import json

import numpy as np
import pytest

import golden_replay as GR

pytestmark = pytest.mark.gpu


def _z(name):
    return np.load(GR.GOLDEN_DIR + '/' + name)


def test_task_methods_match_reference_vectors():
    from gridworld_amd.tasks import Task
    z = _z('s6_task_vectors.npz')
    for p in range(0, len(z['targets']), 2):
        t = Task('', z['targets'][p].astype(np.int32), starting_grid=[])
        f = Task('', z['targets'][p].astype(np.int32), starting_grid=[], full_grid=z['full_grids'][p].astype(np.int32))
        ni = Task('', z['targets'][p].astype(np.int32), starting_grid=[], invariant=False)
        for g in range(0, len(z['grids']), 3):
            grid = z['grids'][g].astype(np.int32)
            assert t.maximal_intersection(grid) == z['max_int'][p, g]
            assert t.argmax_intersection(grid) == tuple(z['argmax'][p, g])
            assert f.maximal_intersection(grid) == z['fg_max_int'][p, g]
            assert f.argmax_intersection(grid) == tuple(z['fg_argmax'][p, g])
            assert ni.maximal_intersection(grid) == z['ni_max_int'][p, g]
            assert ni.argmax_intersection(grid) == tuple(z['ni_argmax'][p, g])


def test_get_intersection_matches_reference():
    from gridworld_amd.tasks import Task
    z6, z = _z('s6_task_vectors.npz'), _z('s10_task_protocol.npz')
    tasks = {}
    for p, g, q, v in zip(z['gi_target'], z['gi_grid'], z['gi_query'], z['gi_value']):
        t = tasks.setdefault(int(p), Task('', z6['targets'][p].astype(np.int32), starting_grid=[]))
        assert t.get_intersection(z6['grids'][g].astype(np.int32), *[int(x) for x in q]) == v, (p, g, q)


def test_step_intersection_sequences_match_reference():
    from gridworld_amd.tasks import Task, Tasks
    z = _z('s10_task_protocol.npz')
    for k in range(len(z['si_targets'])):
        use_full, inv = z['si_flags'][k]
        t = Task('', z['si_targets'][k].astype(np.int32), starting_grid=Tasks.dense_to_sparse(z['si_starts'][k].astype(np.int32)),
                 full_grid=z['si_fulls'][k].astype(np.int32) if use_full else None, invariant=bool(inv))
        t.reset()
        assert (t.max_int, t.prev_grid_size) == tuple(z['si_reset'][k]), k
        for i, g in enumerate(z['si_grids'][k]):
            r, w, d = t.step_intersection(g.astype(np.int32))
            assert (r, w, int(d), t.max_int, t.prev_grid_size) == tuple(z['si_out'][k, i]), (k, i)
            assert (t.right_placement, t.wrong_placement) == (r, w)


@pytest.mark.parametrize('tag,progressive', [('prog', True), ('noprog', False)])
def test_subtasks_progressive_goals_match_reference(tag, progressive):
    from gridworld_amd.tasks import Subtasks
    z = _z('s10_task_protocol.npz')
    spec = json.loads(str(z['sub_spec']))
    seq = [[tuple(b) for b in s] for s in spec['seq']]
    np.random.seed(77)
    st = Subtasks(spec['dialog'], seq, progressive=progressive)
    st.next = 0
    st.reset()
    assert (st.task_start, st.task_goal) == (0, 1)
    for i, g in enumerate(z['sub_%s_grids' % tag]):
        r, w, d = st.step_intersection(g.astype(np.int32))
        assert (r, w, int(d), st.task_goal, st.current.target_size, st.current.max_int) == tuple(z['sub_%s_out' % tag][i]), i
    assert st.current.chat == str(z['sub_%s_chat' % tag]) and st.current.last_instruction == str(z['sub_%s_last' % tag])
    assert st.target_size == st.current.target_size  # attribute fall-through to the current task
    other = st.create_task(-1, 0)
    assert st.set_task_obj(other) is other and st.current is other
    assert st.set_task(2).target_size == len(seq[2]) and (st.task_start, st.task_goal) == (1, 2)
