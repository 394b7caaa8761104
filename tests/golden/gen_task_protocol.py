#!/usr/bin/env python3
"""Golden vectors for the host Task protocol: the reference's
Task.get_intersection, Task.reset + step_intersection sequences (incl. the stale cached max_int), and
Subtasks.step_intersection with progressive goal switching (gridworld/tasks/task.py:74-119, 138-145,
288-298).  Build container only (needs /root/reference); writes s10_task_protocol.npz."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402


def grid_walk(rng, target, start, n_steps):
    """A sequence of grids: start, then one cell placed / broken / recoloured per step, biased to the target."""
    g = start.copy()
    seq = []
    tgt_cells = np.argwhere(target != 0)
    for _ in range(n_steps):
        r = rng.rand()
        if r < 0.55 and len(tgt_cells):   # place a (mostly correct) target block, possibly shifted
            y, x, z = tgt_cells[rng.randint(len(tgt_cells))]
            dx, dz = (rng.randint(-1, 2), rng.randint(-1, 2)) if rng.rand() < 0.4 else (0, 0)
            xx, zz = int(np.clip(x + dx, 0, 10)), int(np.clip(z + dz, 0, 10))
            g[y, xx, zz] = target[y, x, z] if rng.rand() < 0.85 else rng.randint(1, 7)
        elif r < 0.8 and (g != 0).any():  # break a block
            cells = np.argwhere(g != 0)
            y, x, z = cells[rng.randint(len(cells))]
            g[y, x, z] = 0
        elif r < 0.9 and (g != 0).any():  # recolour in place (block count unchanged -> stale max_int)
            cells = np.argwhere(g != 0)
            y, x, z = cells[rng.randint(len(cells))]
            g[y, x, z] = rng.randint(1, 7)
        # else: nothing changes
        seq.append(g.copy())
    return np.stack(seq)


def main():
    gym, Task, Tasks = H.load_reference()
    from gridworld.tasks.task import Subtasks
    z6 = np.load(os.path.join(HERE, 's6_task_vectors.npz'))
    goals = H.load_cdm_goals()
    rng = np.random.RandomState(1010)
    out = {}

    # ---- get_intersection at explicit (dx, dz, rot)
    picks = [1, 2, 3, 5, 9, 13, 14, 16, 20, 23]
    gi_t, gi_g, gi_q, gi_v = [], [], [], []
    for p in picks:
        task = Task('', z6['targets'][p].astype(np.int32), starting_grid=[])
        for gidx in rng.choice(len(z6['grids']), size=4, replace=False):
            grid = z6['grids'][gidx].astype(np.int32)
            for _ in range(6):
                dx, dz, rot = int(rng.randint(-10, 11)), int(rng.randint(-10, 11)), int(rng.randint(4))
                gi_t.append(p); gi_g.append(gidx); gi_q.append((dx, dz, rot))
                gi_v.append(task.get_intersection(grid, dx, dz, rot))
            am = task.argmax_intersection(grid)   # and at the argmax itself
            gi_t.append(p); gi_g.append(gidx); gi_q.append(tuple(int(v) for v in am))
            gi_v.append(task.get_intersection(grid, *am))
    out.update(gi_target=np.array(gi_t, np.int32), gi_grid=np.array(gi_g, np.int32), gi_query=np.array(gi_q, np.int32),
               gi_value=np.array(gi_v, np.int32))

    # ---- Task.reset + step_intersection sequences (targets with / without starting grid, full grid, invariant)
    names = sorted(goals)
    cases = []
    for k in range(8):
        tgt = goals[names[rng.randint(len(names))]].astype(np.int32)
        cells = np.argwhere(tgt != 0)
        start = np.zeros_like(tgt)
        if k % 2 == 1 and len(cells) > 2:
            sub = cells[rng.permutation(len(cells))[:len(cells) // 3]]
            for y, x, zc in sub:
                start[y, x, zc] = tgt[y, x, zc]
        cases.append((tgt, start, k % 4 == 2, k >= 6))
    si_targets, si_starts, si_fulls, si_inv, si_grids, si_out, si_reset = [], [], [], [], [], [], []
    for tgt, start, use_full, non_inv in cases:
        full = None
        if use_full:
            full = tgt.copy()
            extra = rng.permutation(1089)[:4]
            full.reshape(-1)[extra] = np.where(full.reshape(-1)[extra] == 0, 2, full.reshape(-1)[extra])
        task = Task('', tgt, starting_grid=H.dense_to_sparse(start), full_grid=full, invariant=not non_inv)
        task.reset()
        si_reset.append((task.max_int, task.prev_grid_size))
        seq = grid_walk(rng, tgt, start, 40)
        res = []
        for g in seq:
            r, w, d = task.step_intersection(g)
            res.append((r, w, int(d), task.max_int, task.prev_grid_size))
        si_targets.append(tgt); si_starts.append(start)
        si_fulls.append(np.zeros_like(tgt) if full is None else full)
        si_inv.append((int(use_full), int(not non_inv)))
        si_grids.append(seq); si_out.append(res)
    out.update(si_targets=np.stack(si_targets).astype(np.int8), si_starts=np.stack(si_starts).astype(np.int8),
               si_fulls=np.stack(si_fulls).astype(np.int8), si_flags=np.array(si_inv, np.int32),
               si_grids=np.stack(si_grids).astype(np.int8), si_out=np.array(si_out, np.int32),
               si_reset=np.array(si_reset, np.int32))

    # ---- Subtasks.step_intersection, progressive and not
    blocks = H.dense_to_sparse(goals['C3'])
    cuts = [len(blocks) * k // 4 for k in range(1, 5)]
    seq_blocks = [blocks[:c] for c in cuts]
    dialog = [['<Architect> step %d' % k, '<Builder> ok'] for k in range(4)]
    out['sub_spec'] = json.dumps(dict(dialog=dialog, seq=[[list(map(int, b)) for b in s] for s in seq_blocks]))
    for tag, progressive in (('prog', True), ('noprog', False)):
        np.random.seed(77)
        st = Subtasks(dialog, seq_blocks, progressive=progressive)
        st.next = 0                      # start from the first structure, goal = the second
        st.reset()
        # build the whole structure block by block on top of the starting grid, with two mistakes on the way
        grid = np.asarray(Tasks.to_dense(list(seq_blocks[0]))).astype(np.int32).copy()
        grids, res = [], []
        todo = [b for b in blocks if grid[b[1] + 1, b[0] + 5, b[2] + 5] == 0]
        for k, (x, y, zc, c) in enumerate(todo):
            if k in (2, 5):              # a wrong block, then removed again
                wy, wx, wz = 8, 0, k
                grid[wy, wx, wz] = 3
                grids.append(grid.copy()); r = st.step_intersection(grid)
                res.append((r[0], r[1], int(r[2]), st.task_goal, st.current.target_size, st.current.max_int))
                grid[wy, wx, wz] = 0
                grids.append(grid.copy()); r = st.step_intersection(grid)
                res.append((r[0], r[1], int(r[2]), st.task_goal, st.current.target_size, st.current.max_int))
            grid[y + 1, x + 5, zc + 5] = c
            grids.append(grid.copy()); r = st.step_intersection(grid)
            res.append((r[0], r[1], int(r[2]), st.task_goal, st.current.target_size, st.current.max_int))
        out['sub_%s_grids' % tag] = np.stack(grids).astype(np.int8)
        out['sub_%s_out' % tag] = np.array(res, np.int32)
        out['sub_%s_chat' % tag] = st.current.chat
        out['sub_%s_last' % tag] = st.current.last_instruction
    # ---- Tasks.to_sparse on a dense array, exactly as the reference returns it (tasks/task.py:178-187: the
    # nonzero() indices (y, x, z) are unpacked as (x, y, z), i.e. (y_idx - 5, x_idx - 1, z_idx - 5, id))
    dense = goals['C17'].astype(np.int32)
    out['to_sparse_in'] = dense.astype(np.int8)
    out['to_sparse_out'] = np.array([[int(v) for v in b] for b in Tasks.to_sparse(dense)], np.int32)
    path = os.path.join(HERE, 's10_task_protocol.npz')
    np.savez_compressed(path, **out)
    print('s10_task_protocol: %d get_intersection queries, %d step sequences, subtasks goals %s / %s -> %.0f KiB' % (
        len(gi_v), len(cases), out['sub_prog_out'][:, 3].tolist()[-3:], out['sub_noprog_out'][:, 3].tolist()[-3:],
        os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
