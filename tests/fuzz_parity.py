#!/usr/bin/env python3
"""GPU box: randomised parity fuzz (test infrastructure: it drives the CPU oracle, so it lives under tests/) -- random batch sizes (ragged), lanes per env, reward mode, max_steps, start
grids, scales and action mixes, HIP path vs the CPU oracle at every step (walking and, with the oracle in
device-trig mode, flying).  Not part of the test suite (minutes); prints the first mismatch.

    python tests/fuzz_parity.py [n_cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gridworld_amd import VecGridWorld  # noqa: E402
from oracle import oracle as O  # noqa: E402


def targets(rng, n, with_start, dense=False):
    tg = np.zeros((n, 9, 11, 11), np.int8)
    st = np.zeros_like(tg)
    if dense:   # whole floors of one colour in the starting grid: inventories far below zero (env.py:243-246)
        for e in range(n):
            if rng.rand() < 0.5:
                for y in range(rng.randint(1, 5)):
                    st[e, y] = rng.randint(1, 7)
                tg[e] = st[e]
                tg[e].reshape(-1)[rng.randint(0, 1089, 8)] = rng.randint(0, 7, 8)
        return tg, st
    for e in range(n):
        k = rng.randint(1, 40)
        lv = rng.randint(1, 4)
        for _ in range(k):
            tg[e, rng.randint(lv), rng.randint(11), rng.randint(11)] = rng.randint(1, 7)
        if with_start and rng.rand() < 0.5:
            for _ in range(rng.randint(1, 12)):
                y, x, z = rng.randint(2), rng.randint(11), rng.randint(11)
                st[e, y, x, z] = tg[e, y, x, z] if (tg[e, y, x, z] and rng.rand() < 0.6) else rng.randint(1, 7)
    return tg, st


def full_grids(rng, tg):
    """Subtasks-style full structures: a superset of each target (tasks/task.py:63-66: the user task's admissible
    translations come from it), which changes GridWorld.max_int at reset and so SizeReward's first reward."""
    fg = tg.copy()
    for e in range(len(fg)):
        for _ in range(rng.randint(0, 8)):
            y, x, z = rng.randint(3), rng.randint(11), rng.randint(11)
            if fg[e, y, x, z] == 0:
                fg[e, y, x, z] = rng.randint(1, 7)
    return fg


def compare(env, ob, where):
    torch.cuda.synchronize()
    for name, a, b in (('done', env.done.cpu().numpy(), ob.done),
                       ('reward', env.reward.cpu().numpy().view(np.uint32), ob.reward.view(np.uint32)),
                       ('grid', env.grid.cpu().numpy().reshape(env.num_envs, -1), ob.grid),
                       ('inventory', env.inventory.cpu().numpy(), ob.inventory),
                       ('agentPos', env.agent_pos.cpu().numpy().view(np.uint32), ob.agentPos.view(np.uint32)),
                       ('compass', env.compass.cpu().numpy().view(np.uint32), ob.compass.view(np.uint32))):
        if not np.array_equal(a, b):
            bad = np.argwhere(np.atleast_2d(a != b).reshape(len(a), -1).any(1))[:5, 0]
            raise AssertionError(f'{where}: {name} differs in envs {bad.tolist()}')


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    for c in range(cases):
        n = int(rng.choice([1, 3, 15, 16, 17, 63, 64, 65, 200, 777, 1500, 4097]))
        gs = int(rng.choice([0, 64, 32, 16, 8, 4, 2, 1]))
        mode = 'flying' if rng.rand() < 0.3 else 'walking'
        kw = dict(size_reward=bool(rng.rand() < 0.3), max_steps=int(rng.choice([1, 7, 40, 250])),
                  right_placement_scale=float(rng.choice([1.0, 2.5])), wrong_placement_scale=float(rng.choice([0.1, 0.25])),
                  select_and_place=bool(rng.rand() < 0.7))
        autoreset = bool(rng.rand() < 0.6)
        with_start = rng.rand() < 0.5
        T = int(rng.choice([30, 90]))
        dense = rng.rand() < 0.12
        tg, st = targets(rng, n, with_start, dense)
        fg = full_grids(rng, tg) if rng.rand() < 0.4 else None
        # a third of the cases: arbitrary initial poses -- off the 5-degree lattice (general trig path, oracle in
        # device-trig mode), up to the edge of the validated range (clamped occupancy keys, agents outside the zone)
        poses = None
        if dense:   # on top of the floors
            poses = np.zeros((n, 5))
            poses[:, 1] = np.where(st[:, :, 5, 5].any(1), (st[:, :, 5, 5] != 0).sum(1) - 2 + 0.5 + 1.25, 0.0)
        elif rng.rand() < 0.33:
            poses = np.stack([rng.uniform(-9.5, 9.5, n), np.where(rng.rand(n) < 0.8, rng.uniform(-0.25, 9.0, n), rng.uniform(-6.0, 30.0, n)),
                              rng.uniform(-9.5, 9.5, n), rng.uniform(-400.0, 400.0, n), rng.uniform(-90.0, 90.0, n)], axis=1)
            if rng.rand() < 0.5:   # half of them on the lattice, at the border
                poses[:, 3:] = np.round(poses[:, 3:] / 5.0) * 5.0
                poses[:, [0, 2]] = np.round(poses[:, [0, 2]] * 4.0) / 4.0
        desc = f'case {c}: n={n} gs={gs} {mode} autoreset={autoreset} start={with_start} dense={dense} full_grid={fg is not None} poses={poses is not None} {kw}'
        env = VecGridWorld(n, action_space=mode, autoreset=autoreset, lanes_per_env=gs, **kw)
        env.set_tasks(tg, st, full_grids=fg, init_pose=poses)
        env.reset()
        ob = O.OracleBatch(n, action_space=mode, **kw)
        ob.set_tasks(tg, st, full_grids=fg)
        if poses is not None:
            ob.set_initial_pose(poses)
        ob.reset()
        try:
            O.use_device_trig(mode == 'flying' or poses is not None)
            fused = mode == 'walking' and c % 3 == 0   # every third walking case: chunks through the fused replay
            t = 0
            while fused and t < T:
                w = np.array([1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4, 4], float)
                chunk = rng.choice(18, size=(int(rng.choice([1, 7, 30])), n), p=w / w.sum()).astype(np.int32)
                rw, dn = env.rollout_actions(torch.as_tensor(chunk), return_rewards=True)
                torch.cuda.synchronize()
                rw, dn = rw.cpu().numpy(), dn.cpu().numpy()
                for k in range(len(chunk)):
                    ob.step_walking(chunk[k], autoreset=autoreset, nthreads=8)
                    if not (np.array_equal(rw[k].view(np.uint32), ob.reward.view(np.uint32)) and np.array_equal(dn[k], ob.done)):
                        raise AssertionError(f'{desc} fused chunk at step {t + k}: per-step reward / done differ')
                t += len(chunk)
                compare(env, ob, f'{desc} after fused chunk ending at step {t}')
            for t in range(0 if not fused else T, T):
                if mode == 'walking':
                    w = np.array([1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4, 4], float)
                    a = rng.choice(18, size=n, p=w / w.sum()).astype(np.int32)
                    env.step(torch.as_tensor(a))
                    ob.step_walking(a, autoreset=autoreset, nthreads=8)
                else:
                    mv = np.float32(rng.uniform(-1, 1, (n, 3))) * (rng.rand(n, 1) < 0.8)
                    cam = np.float32(rng.uniform(-15, 15, (n, 2)))
                    inv = rng.randint(0, 7, n).astype(np.int32)
                    plc = rng.randint(0, 3, n).astype(np.int32)
                    env.step(dict(movement=torch.as_tensor(mv.astype(np.float32)), camera=torch.as_tensor(cam), inventory=torch.as_tensor(inv),
                                  placement=torch.as_tensor(plc)))
                    ob.step_flying(mv.astype(np.float32), cam, inv, plc, autoreset=autoreset, nthreads=8)
                compare(env, ob, f'{desc} step {t}')
        finally:
            O.use_device_trig(False)
        print('ok', desc + (' [fused replay]' if mode == 'walking' and c % 3 == 0 else ''), flush=True)
    print('fuzz: all', cases, 'cases bit-exact')


if __name__ == '__main__':
    main()
