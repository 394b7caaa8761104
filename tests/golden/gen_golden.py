#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by running the Python reference.

Run in the build container only:   python tests/golden/gen_golden.py
(needs /root/reference; the GPU box never runs this -- it only reads the .npz files).

Every file stores the INPUTS (kwargs, targets, starting grids, actions) next to the
reference's OUTPUTS (obs, reward, done, grid, float64 agent internals), so the CPU
oracle and the HIP path can both be replayed against it.  Scenario list follows
SURVEY.md Appendix B (S1..S6).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402


def starts_to_dense(starts):
    out = np.zeros((len(starts), 9, 11, 11), np.int8)
    for e, blocks in enumerate(starts):
        for x, y, z, c in blocks:
            out[e, y + 1, x + 5, z + 5] = c
    return out


def save(name, kwargs, targets, starts, actions, ref, task_kwargs=None, init_pose=None):
    walkdict = isinstance(actions, dict) and 'buttons' in actions
    flying = isinstance(actions, dict) and not walkdict
    d = dict(kwargs=json.dumps(kwargs), task_kwargs=json.dumps(task_kwargs or {}),
             targets=np.asarray(targets, np.int8), starts=starts_to_dense(starts))
    if init_pose is not None:
        d['init_pose'] = np.asarray(init_pose, np.float64)
    if walkdict:
        d.update(act_buttons=actions['buttons'].astype(np.uint8), act_camera=actions['camera'].astype(np.float32))
    elif flying:
        d.update(act_movement=actions['movement'].astype(np.float32),
                 act_camera=actions['camera'].astype(np.float32),
                 act_inventory=actions['inventory'].astype(np.int32),
                 act_placement=actions['placement'].astype(np.int32))
    else:
        d['actions'] = np.asarray(actions, np.int32)
    # the grid is stored as a per-step change log (at most one cell changes per step,
    # SURVEY F16) plus the grid after the last step; resets restore `starts`.
    E, T = ref['done'].shape
    g = ref['grid'].reshape(E, T, -1)
    prev = np.concatenate([d['starts'].reshape(E, 1, -1), g[:, :-1]], axis=1)
    # a reset before step t restores the starting grid
    rb = ref['reset_before'].astype(bool)
    prev = np.where(rb[:, :, None], d['starts'].reshape(E, 1, -1), prev)
    diff = g != prev
    assert diff.sum(-1).max() <= 1
    idx = np.where(diff.any(-1), diff.argmax(-1), -1).astype(np.int16)
    val = np.take_along_axis(g, np.maximum(idx, 0)[:, :, None].astype(np.int64), axis=2)[:, :, 0]
    d.update(grid_change_idx=idx, grid_change_val=np.where(idx >= 0, val, 0).astype(np.int8),
             grid_final=g[:, -1].reshape(E, 9, 11, 11))
    for k, v in ref.items():
        if k != 'grid':
            d[k] = v
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **d)
    print(f'{name}: E={E} T={T} resets={int(rb.sum())} changed={int((idx >= 0).sum())} '
          f'reward_nz={int((ref["reward"] != 0).sum())} done={int(ref["done"].sum())} '
          f'-> {os.path.getsize(path) / 1024:.0f} KiB')


def rt20(rng):
    """RandomTasks(max_blocks=20, height_levels=1, max_dist=2, num_colors=6)-shaped target
    (tasks/task_set.py:135-157): 20 blocks on level 0 inside the 5x5 window around a first
    block drawn from [2,8]^2 (avoids the reference's infinite loop near corners)."""
    g = np.zeros((9, 11, 11), np.int8)
    bx, bz = rng.randint(2, 9, size=2)
    cells = [(bx + dx, bz + dz) for dx in range(-2, 3) for dz in range(-2, 3)]
    for i in rng.permutation(25)[:20]:
        g[0, cells[i][0], cells[i][1]] = rng.randint(1, 7)
    return g


def uniform20(rng):
    g = np.zeros(1089, np.int8)
    g[rng.permutation(1089)[:20]] = rng.randint(1, 7, size=20)
    return g.reshape(9, 11, 11)


def cdm_with_start(rng, goals, names):
    g = goals[names[rng.randint(len(names))]]
    sp = H.dense_to_sparse(g)
    k = rng.randint(0, len(sp) + 1)
    sub = [sp[i] for i in rng.permutation(len(sp))[:k]]
    for _ in range(rng.randint(0, 4)):  # blocks that are NOT in the target -> negative synthetic ids
        x, y, z = int(rng.randint(-5, 6)), int(rng.randint(-1, 3)), int(rng.randint(-5, 6))
        if g[y + 1, x + 5, z + 5] == 0 and not any(b[:3] == (x, y, z) for b in sub):
            sub.append((x, y, z, int(rng.randint(1, 7))))
    return g, sub


def flying_actions(rng, E, T):
    return dict(movement=rng.uniform(-1, 1, size=(E, T, 3)).astype(np.float32),
                camera=rng.uniform(-5, 5, size=(E, T, 2)).astype(np.float32),
                inventory=rng.randint(7, size=(E, T)), placement=rng.randint(3, size=(E, T)))


def pad_actions(seqs, fill=0):
    T = max(len(s) for s in seqs)
    return np.array([list(s) + [fill] * (T - len(s)) for s in seqs], np.int32)


def main():
    goals = H.load_cdm_goals()
    names = sorted(goals)

    # S1 -- walking, DUMMY-equivalent task (tasks/task_set.py:160: one blue block at dense [8,10,10])
    rng = np.random.RandomState(101)
    E, T = 16, 520
    tg = np.zeros((E, 9, 11, 11), np.int8)
    tg[:, 8, 10, 10] = 1
    acts = np.stack([np.random.RandomState(e).randint(18, size=T) for e in range(E)])
    kw = dict(size_reward=False)
    save('s1_walk_dummy', kw, tg, [[]] * E, acts,
         H.run_batch(kw, tg, [[]] * E, acts, task_kwargs=dict(invariant=False)),
         task_kwargs=dict(invariant=False))
    kw = dict(size_reward=True)  # the gym.make default (SizeReward wrapper, env.py:316-331)
    save('s1_walk_dummy_sizereward', kw, tg[:4], [[]] * 4, acts[:4],
         H.run_batch(kw, tg[:4], [[]] * 4, acts[:4], task_kwargs=dict(invariant=False)),
         task_kwargs=dict(invariant=False))

    # S3 -- walking, rt20 + uniform20 targets, empty starting grid
    rng = np.random.RandomState(303)
    E, T = 24, 500
    tg = np.stack([rt20(rng) for _ in range(16)] + [uniform20(rng) for _ in range(8)])
    acts = rng.randint(18, size=(E, T))
    kw = dict(size_reward=False)
    save('s3_walk_rt20', kw, tg, [[]] * E, acts, H.run_batch(kw, tg, [[]] * E, acts))

    # S2 -- walking, CDM structures (skills/goals.pkl) with random starting subsets
    rng = np.random.RandomState(202)
    E, T = 24, 500
    pairs = [cdm_with_start(rng, goals, names) for _ in range(E)]
    tg = np.stack([p[0] for p in pairs])
    st = [p[1] for p in pairs]
    acts = rng.randint(18, size=(E, T))
    kw = dict(size_reward=False)
    save('s2_walk_cdm', kw, tg, st, acts, H.run_batch(kw, tg, st, acts))
    kw = dict(size_reward=True, max_steps=100)
    save('s2_walk_cdm_sizereward', kw, tg[:8], st[:8], acts[:8, :300],
         H.run_batch(kw, tg[:8], st[:8], acts[:8, :300]))

    # S4 -- flying (Python-float actions widened from float32), rt20 targets, native libm
    rng = np.random.RandomState(404)
    E, T = 16, 500
    tg = np.stack([rt20(rng) for _ in range(E)])
    fa = flying_actions(rng, E, T)
    kw = dict(size_reward=False, action_space='flying')
    save('s4_fly_rt20', kw, tg, [[]] * E, fa, H.run_batch(kw, tg, [[]] * E, fa))
    pairs = [cdm_with_start(rng, goals, names) for _ in range(8)]
    tg = np.stack([p[0] for p in pairs])
    st = [p[1] for p in pairs]
    fa = flying_actions(rng, 8, 400)
    save('s4_fly_cdm', kw, tg, st, fa, H.run_batch(kw, tg, st, fa))

    # S5 -- scripted edge cases (walking)
    two = np.zeros((9, 11, 11), np.int8)
    two[0, 5, 3] = 1
    two[0, 5, 4] = 1
    dummy = np.zeros((9, 11, 11), np.int8)
    dummy[8, 10, 10] = 1
    one = np.zeros((9, 11, 11), np.int8)
    one[0, 5, 4] = 1
    tower = [14] * 18
    for _ in range(7):
        tower += [5, 0, 0, 0, 0, 17, 0, 0, 0, 0, 0, 0]
    tower += [1] * 4 + [0] * 30 + [2] * 6 + [5] + [0] * 14
    # 19 blue blocks in the starting grid -> inventory[0] == 1 at reset
    nineteen = [(x, -1, z, 1) for x in range(-5, 5) for z in (-5, -4)][:19]
    seqs = [
        # SURVEY Appendix B known-answer trace: fall, move, jump arc, look down 45, place / rejected
        # place (agent overlap) / break / break ground (rejected) / hotbar-3 select_and_place
        ([0, 0, 0, 0, 1, 4, 5] + [0] * 13 + [14] * 9 + [17, 17, 16, 16, 8] + [0] * 4, two, []),
        # walk into the pad-2 boundary (|z| -> 7, then |x| -> 7), jump there, come back
        ([1] * 40 + [4] * 40 + [5] + [0] * 14 + [2] * 12 + [3] * 12 + [13] * 30 + [1] * 30, dummy, []),
        # build a 7-block tower under the agent, walk off, fall (time_int_steps 4 / 8 / 12)
        (tower, dummy, []),
        # inventory exhaustion: one blue block left, place it, next place is refused, break gives it back
        ([14] * 9 + [17, 17, 12, 17, 13, 13, 17, 16, 17] + [0] * 3, dummy, nineteen),
        # camera: yaw wrap in both directions (0 and 360 both occur), pitch clamp at +-90
        ([12] * 3 + [13] * 80 + [12] * 80 + [15] * 20 + [14] * 40 + [16, 17], dummy, []),
        # all hotbar ids with select_and_place while turning
        ([14] * 6 + sum([[6 + k, 13, 13, 13] for k in range(6)] * 3, []), two, []),
        # scripted completion: one-block target placed -> done, reward = right_placement_scale
        ([0, 0, 0, 0, 1, 4] + [14] * 9 + [17] + [0] * 5, one, []),
        # target == starting grid -> synthetic target empty -> done on the first step
        ([0] * 6, one, [(0, -1, -1, 1)]),
        # spawn inside a block column: hit_test's first sample is already in the world (previous None)
        ([17, 16, 0, 5, 0, 0, 16, 17] + [14] * 18 + [16, 17, 0, 0], dummy,
         [(0, 0, 0, 2), (0, -1, 0, 3), (1, 0, 0, 4)]),
    ]
    acts = pad_actions([s[0] for s in seqs])
    tg = np.stack([s[1] for s in seqs])
    st = [s[2] for s in seqs]
    kw = dict(size_reward=False)
    save('s5_scripted', kw, tg, st, acts, H.run_batch(kw, tg, st, acts))
    kw = dict(size_reward=False, right_placement_scale=2., wrong_placement_scale=1.)  # README.md:137-181
    save('s5_scripted_scales', kw, tg, st, acts, H.run_batch(kw, tg, st, acts))
    # episode boundary every 12 steps: dy / active_block / time_int_steps leak through reset
    kw = dict(size_reward=False, max_steps=12)
    leak = pad_actions([[0] * 10 + [5, 8] + [0] * 12 + [5, 0, 0, 0, 0, 17] * 6,
                        tower[:100]])
    save('s5_scripted_leak', kw, np.stack([two, dummy]), [[], []], leak,
         H.run_batch(kw, np.stack([two, dummy]), [[], []], leak))
    # non-default initial pose through initialize_world (env.py:177-187): x,y,z,yaw,pitch
    rng = np.random.RandomState(505)
    E, T = 6, 200
    poses = np.array([[2.5, 3.0, -1.25, 90., -30.], [-4.0, 0.0, 4.0, 355., 10.], [0.3, 7.5, 0.3, 180., -90.],
                      [6.5, -0.25, 6.5, 45., 0.], [-7.0, 2.0, 0.0, 270., 45.], [0., 0., 0., 360., 90.]])
    tg = np.stack([rt20(rng) for _ in range(E)])
    acts = rng.randint(18, size=(E, T))
    kw = dict(size_reward=False)
    save('s5_init_pose', kw, tg, [[]] * E, acts,
         H.run_batch(kw, tg, [[]] * E, acts, init_pose=poses), init_pose=poses)

    # S8 -- walking with discretize=False (Dict of buttons + continuous camera, env.py:60-70)
    rng = np.random.RandomState(808)
    E, T = 12, 400
    tg = np.stack([rt20(rng) for _ in range(8)] + [cdm_with_start(rng, goals, names)[0] for _ in range(4)])
    b = (rng.rand(E, T, 8) < 0.25).astype(np.uint8)
    b[:, :, 7] = rng.randint(0, 7, size=(E, T)) * (rng.rand(E, T) < 0.3)
    cam = rng.uniform(-5, 5, size=(E, T, 2)).astype(np.float32)
    cam[:, ::7] = 0.0  # steps without camera motion
    kw = dict(size_reward=False, discretize=False)
    wd = dict(buttons=b, camera=cam)
    save('s8_walk_dict', kw, tg, [[]] * E, wd, H.run_batch(kw, tg, [[]] * E, wd))

    # S4c -- flying with the reference's trig replaced by correctly rounded sin/cos/atan2 ("CR-libm oracle")
    rng = np.random.RandomState(414)
    E, T = 8, 300
    pairs = [cdm_with_start(rng, goals, names) for _ in range(4)]
    tg = np.stack([rt20(rng) for _ in range(4)] + [p[0] for p in pairs])
    st = [[]] * 4 + [p[1] for p in pairs]
    fa = flying_actions(rng, E, T)
    fa['movement'][:, ::9, rng.randint(3)] = 0.0   # exact zeros / axis-aligned strafes
    kw = dict(size_reward=False, action_space='flying')
    with H.cr_libm():
        save('s4_fly_crlibm', kw, tg, st, fa, H.run_batch(kw, tg, st, fa))

    rng = np.random.RandomState(818)   # the same for walking with discretize=False
    E, T = 6, 300
    tg = np.stack([rt20(rng) for _ in range(4)] + [cdm_with_start(rng, goals, names)[0] for _ in range(2)])
    b = (rng.rand(E, T, 8) < 0.3).astype(np.uint8)
    b[:, :, 7] = rng.randint(0, 7, size=(E, T)) * (rng.rand(E, T) < 0.3)
    cam = rng.uniform(-5, 5, size=(E, T, 2)).astype(np.float32)
    kw = dict(size_reward=False, discretize=False)
    wd = dict(buttons=b, camera=cam)
    with H.cr_libm():
        save('s8_walk_dict_crlibm', kw, tg, [[]] * E, wd, H.run_batch(kw, tg, [[]] * E, wd))

    # S9 -- select_and_place=False (GridWorld's own default; hotbar only selects, core/world.py:444-446)
    rng = np.random.RandomState(909)
    E, T = 8, 300
    pairs = [cdm_with_start(rng, goals, names) for _ in range(4)]
    tg = np.stack([rt20(rng) for _ in range(4)] + [p[0] for p in pairs])
    st = [[]] * 4 + [p[1] for p in pairs]
    acts = rng.choice(18, size=(E, T), p=np.array([1] * 14 + [3] + [1] + [2, 3]) / 23.0)
    kw = dict(size_reward=False, select_and_place=False)
    save('s9_walk_no_select_and_place', kw, tg, st, acts, H.run_batch(kw, tg, st, acts))
    fa = flying_actions(rng, 6, 300)
    kw = dict(size_reward=False, select_and_place=False, action_space='flying')
    save('s9_fly_no_select_and_place', kw, tg[:6], st[:6], fa, H.run_batch(kw, tg[:6], st[:6], fa))

    # S6 -- pure Task vectors: admissible sets, rotations, maximal / argmax intersection
    rng = np.random.RandomState(606)
    targets = [np.zeros((9, 11, 11), np.int8), dummy, two]
    targets += [rt20(rng) for _ in range(6)] + [uniform20(rng) for _ in range(4)]
    targets += [goals[n] for n in ('C1', 'C3', 'C12', 'C17', 'C32', 'C100', 'C157')]
    for _ in range(4):  # synthetic targets with negative ids (target - starting grid)
        g, sub = cdm_with_start(rng, goals, names)
        targets.append((g.astype(np.int16) - starts_to_dense([sub])[0]).astype(np.int8))
    targets = np.stack(targets)
    grids = [np.zeros((9, 11, 11), np.int8)]
    for p in rng.randint(len(targets), size=10):  # shifted / rotated / perturbed copies of targets
        g = np.rot90(targets[p], k=rng.randint(4), axes=(1, 2)).copy()
        g = np.roll(g, (rng.randint(-2, 3), rng.randint(-2, 3)), axis=(1, 2))
        m = rng.rand(*g.shape) < 0.3
        g[m] = 0
        grids.append(g)
    for _ in range(6):
        g = np.zeros(1089, np.int8)
        n = rng.randint(1, 60)
        g[rng.permutation(1089)[:n]] = rng.randint(-3, 7, size=n)
        grids.append(g.reshape(9, 11, 11))
    grids = np.stack(grids)
    tv = H.task_vectors(targets, grids)
    tv_ni = H.task_vectors(targets, grids, invariant=False)
    # full_grid variant (tasks/task.py:63-72): the full structure is a superset of the target
    fulls = targets.copy()
    for f in fulls:
        extra = rng.permutation(1089)[:5]
        f.reshape(-1)[extra] = np.where(f.reshape(-1)[extra] == 0, 1, f.reshape(-1)[extra])
    tv_fg = H.task_vectors(targets, grids, full_grids=fulls)
    path = os.path.join(HERE, 's6_task_vectors.npz')
    np.savez_compressed(path, targets=targets, grids=grids, full_grids=fulls,
                        **{k: v for k, v in tv.items()},
                        **{'ni_' + k: v for k, v in tv_ni.items() if k in ('adm_count', 'max_int', 'argmax')},
                        **{'fg_' + k: v for k, v in tv_fg.items() if k in ('adm_count', 'adm_mask', 'max_int', 'argmax')})
    print(f's6_task_vectors: P={len(targets)} G={len(grids)} -> {os.path.getsize(path) / 1024:.0f} KiB',
          'adm/rot', tv['adm_count'][:, 0].tolist())


if __name__ == '__main__':
    main()
