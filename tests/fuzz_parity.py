#!/usr/bin/env python3
"""GPU box: randomised parity fuzz (test infrastructure: it drives the CPU oracle, so it lives under tests/).

Every case draws a batch size (ragged), lanes per env, an action space -- walking Discrete(18), walking Dict
(discretize=False) or flying --, reward mode, max_steps, starting grids, full grids, scales, initial poses and an
action mix, and compares the HIP path with the CPU oracle after EVERY step (continuous action spaces and off-lattice
poses with the oracle in device-trig mode).  On top of that a case may draw

  * the fused replay (rollout_actions) for chunks of walking steps;
  * the EXTRA kernel variant: the episode log on (the decoded log of every finished episode must equal what the
    oracle produced step by step), the on-device task sampler (set_task_sampling: the oracle is handed the row the
    device drew) or the on-device RandomTasks generator (the oracle is handed the generated target);
  * VecGridWorld.split(2): the two halves stepped on their own streams;
  * a mid-run state_dict() -> fresh VecGridWorld -> load_state_dict() round trip;
  * host-side masked resets of finished episodes instead of the in-kernel auto-reset.

Not part of the CPU suite (minutes); tests/test_gpu_fuzz.py runs 3 x 100 cases under the driver.

    python tests/fuzz_parity.py [n_cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gridworld_amd import VecGridWorld  # noqa: E402
from oracle import oracle as O  # noqa: E402

WALK_W = np.array([1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4, 4], float)


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


def draw_actions(rng, mode, n):
    """(what VecGridWorld.step takes, what the OracleBatch driver takes)"""
    if mode == 'walking':
        a = rng.choice(18, size=n, p=WALK_W / WALK_W.sum()).astype(np.int32)
        return torch.as_tensor(a), (a,)
    cam = np.float32(rng.uniform(-15, 15, (n, 2)))
    if rng.rand() < 0.2:
        cam[rng.rand(n) < 0.5] = 0.0
    if mode == 'walking_dict':
        b = (rng.rand(n, 8) < 0.3).astype(np.uint8)
        b[:, 7] = rng.randint(0, 7, size=n) * (rng.rand(n) < 0.3)
        return dict(buttons=torch.as_tensor(b), camera=torch.as_tensor(cam)), (b, cam)
    mv = (np.float32(rng.uniform(-1, 1, (n, 3))) * (rng.rand(n, 1) < 0.8)).astype(np.float32)
    inv = rng.randint(0, 7, n).astype(np.int32)
    plc = rng.randint(0, 3, n).astype(np.int32)
    return (dict(movement=torch.as_tensor(mv), camera=torch.as_tensor(cam), inventory=torch.as_tensor(inv),
                 placement=torch.as_tensor(plc)), (mv, cam, inv, plc))


def oracle_step(ob, mode, oa, autoreset):
    if mode == 'walking':
        ob.step_walking(*oa, autoreset=autoreset, nthreads=8)
    elif mode == 'walking_dict':
        ob.step_walking_dict(*oa, autoreset=autoreset, nthreads=8)
    else:
        ob.step_flying(*oa, autoreset=autoreset, nthreads=8)


class LogChecker:
    """Oracle-side history of the logged envs, episode by episode, against EpisodeLogger.collect()."""

    def __init__(self, env, ob, n_logged, cap, where):
        from gridworld_amd.wrappers import EpisodeLogger
        self.log = EpisodeLogger(env, n_envs=n_logged, capacity=cap)
        self.ob, self.n, self.cap, self.where = ob, n_logged, cap, where
        self.cur = [None] * n_logged      # the running episode of every logged env
        self.closed = [[] for _ in range(n_logged)]   # finished (or abandoned) episodes in order
        self.checked = 0

    def _snap(self, e):
        ob = self.ob
        return (ob.agentPos[e].copy(), ob.inventory[e].copy(), np.float32(ob.compass[e]), ob.grid[e].copy())

    def begin(self, mask):
        """after a reset of the masked envs (oracle arrays hold the reset observation)"""
        for e in range(self.n):
            if mask is None or mask[e]:
                if self.cur[e] is not None:
                    self.closed[e].append(self.cur[e])
                self.cur[e] = dict(obs=[self._snap(e)], reward=[], done=[])

    def step(self):
        """after an oracle step WITHOUT auto-reset (arrays hold the step's own observation)"""
        for e in range(self.n):
            c = self.cur[e]
            c['obs'].append(self._snap(e))
            c['reward'].append(np.float32(self.ob.reward[e]))
            c['done'].append(bool(self.ob.done[e]))

    def check(self):
        """every episode the device reports as finished equals the oracle's record of it (first `cap` steps)"""
        for ep in self.log.collect(dump=False):
            e = ep['env']
            hist = self.closed[e] + [self.cur[e]]
            k = ep['episode'] - 1            # the first reset starts episode 1
            assert 0 <= k < len(hist), f'{self.where}: log reports episode {ep["episode"]} of env {e}, the oracle saw {len(hist)}'
            w = hist[k]
            n = len(ep['reward'])
            assert n == min(len(w['reward']), self.cap) and n > 0, f'{self.where}: env {e} episode {k}: {n} logged steps vs {len(w["reward"])}'
            pos = np.stack([o[0] for o in w['obs'][:n + 1]])
            pos[0] = 0                        # the reset observation's agentPos is zeros (env.py:254)
            comp = np.array([o[2] for o in w['obs'][:n + 1]], np.float32)
            comp[0] = 0
            ok = (np.array_equal(ep['agentPos'].view(np.uint32), pos.view(np.uint32)) and
                  np.array_equal(ep['inventory'], np.stack([o[1] for o in w['obs'][:n + 1]])) and
                  np.array_equal(ep['compass'][:, 0].view(np.uint32), comp.view(np.uint32)) and
                  np.array_equal(ep['grid'].reshape(n + 1, -1), np.stack([o[3] for o in w['obs'][:n + 1]]).astype(np.int32)) and
                  np.array_equal(ep['reward'].astype(np.float32).view(np.uint32), np.array(w['reward'][:n], np.float32).view(np.uint32)) and
                  np.array_equal(ep['done'], np.array(w['done'][:n])))
            assert ok, f'{self.where}: logged episode {k} of env {e} differs from the oracle'
            self.checked += 1


def run_case(c, rng):
    n = int(rng.choice([1, 3, 15, 16, 17, 63, 64, 65, 200, 777, 1500, 4097]))
    gs = int(rng.choice([0, 64, 32, 16, 8, 4, 2, 1]))
    mode = str(rng.choice(['walking', 'flying', 'walking_dict'], p=[0.5, 0.3, 0.2]))
    kw = dict(size_reward=bool(rng.rand() < 0.3), max_steps=int(rng.choice([1, 7, 40, 250])),
              right_placement_scale=float(rng.choice([1.0, 2.5])), wrong_placement_scale=float(rng.choice([0.1, 0.25])),
              select_and_place=bool(rng.rand() < 0.7))
    autoreset = bool(rng.rand() < 0.6)
    with_start = rng.rand() < 0.5
    T = int(rng.choice([30, 90]))
    dense = rng.rand() < 0.12
    extra = str(rng.choice(['none', 'log', 'sampling', 'log+sampling', 'random_tasks'], p=[0.5, 0.15, 0.15, 0.1, 0.1]))
    if extra == 'random_tasks':
        dense = with_start = False
    sampling, logging = 'sampling' in extra, 'log' in extra
    ntasks = n if not sampling else int(rng.choice([1, 2, 5, 37]))
    tg, st = targets(rng, ntasks, with_start, dense)
    fg = full_grids(rng, tg) if (rng.rand() < 0.4 and extra != 'random_tasks') else None
    # a third of the cases: arbitrary initial poses -- off the 5-degree lattice (general trig path, oracle in
    # device-trig mode), up to the edge of the validated range (clamped occupancy keys, agents outside the zone)
    poses = None
    if dense:   # on top of the floors
        poses = np.zeros((ntasks, 5))
        poses[:, 1] = np.where(st[:, :, 5, 5].any(1), (st[:, :, 5, 5] != 0).sum(1) - 2 + 0.5 + 1.25, 0.0)
    elif rng.rand() < 0.33:
        m = ntasks
        poses = np.stack([rng.uniform(-9.5, 9.5, m), np.where(rng.rand(m) < 0.8, rng.uniform(-0.25, 9.0, m), rng.uniform(-6.0, 30.0, m)),
                          rng.uniform(-9.5, 9.5, m), rng.uniform(-400.0, 400.0, m), rng.uniform(-90.0, 90.0, m)], axis=1)
        if rng.rand() < 0.5:   # half of them on the lattice, at the border
            poses[:, 3:] = np.round(poses[:, 3:] / 5.0) * 5.0
            poses[:, [0, 2]] = np.round(poses[:, [0, 2]] * 4.0) / 4.0
    manual = extra != 'none'                 # the oracle is reset by hand (it must be told the device's task choice / logged)
    host_resets = manual and not autoreset and rng.rand() < 0.6   # finished episodes: env.reset(mask) from the host
    fused = mode == 'walking' and extra == 'none' and c % 3 == 0   # chunks through the fused replay
    split = mode == 'walking' and not fused and not logging and extra != 'random_tasks' and n % 2 == 0 and rng.rand() < 0.3
    snapshot = not logging and not split and rng.rand() < 0.25
    desc = (f'case {c}: n={n} gs={gs} {mode} autoreset={autoreset} start={with_start} dense={dense} full_grid={fg is not None} '
            f'poses={poses is not None} extra={extra} tasks={ntasks} host_resets={host_resets} split={split} snapshot={snapshot} {kw}')
    space = dict(action_space='flying') if mode == 'flying' else dict(discretize=False) if mode == 'walking_dict' else {}
    mk = lambda: VecGridWorld(n, autoreset=autoreset, lanes_per_env=gs, num_tasks=ntasks, **space, **kw)  # noqa: E731
    env = mk()
    ob = O.OracleBatch(n, **space, **kw)
    row = np.zeros(n, np.int64) if sampling else np.arange(n)

    def give_oracle_tasks(mask):
        """the oracle envs of `mask` get the task the device holds for them"""
        idx = np.nonzero(mask)[0]
        if extra == 'random_tasks':
            gen = env.targets().cpu().numpy()          # the generated targets (empty start: synthetic == user target)
            for e in idx:
                ob.envs[e].set_task(gen[e], None)
        else:
            et = env.env_task.cpu().numpy()
            for e in idx:
                ob.envs[e].set_task(tg[et[e]], st[et[e]], None if fg is None else fg[et[e]])
                if poses is not None:
                    ob.envs[e].set_initial_pose(poses[et[e]])

    samp_seed = int(rng.randint(1 << 30))
    if extra == 'random_tasks':
        rk = dict(max_blocks=int(rng.randint(1, 9)), height_levels=int(rng.randint(1, 4)), max_dist=int(rng.randint(1, 4)),
                  num_colors=int(rng.randint(1, 7)))
        env.set_random_tasks(True, seed=samp_seed, **rk)
        desc += f' {rk}'
    else:
        env.set_tasks(tg, st, full_grids=fg, init_pose=poses, env_task=np.zeros(n, np.int32) if sampling else None)
        if sampling:
            env.set_task_sampling(True, seed=samp_seed)
    checker = None
    if logging:
        n_logged = int(min(n, rng.choice([1, 3, 8])))
        cap = int(rng.choice([kw['max_steps'], max(1, kw['max_steps'] // 2), 5]))
        checker = LogChecker(env, ob, n_logged, cap, desc)
    env.reset()
    torch.cuda.synchronize()
    if manual:
        give_oracle_tasks(np.ones(n, bool))
    else:
        ob.set_tasks(tg, st, full_grids=fg)
        if poses is not None:
            ob.set_initial_pose(poses)
    ob.reset()
    if checker:
        checker.begin(None)
    subs = env.split(2) if split else None
    try:
        O.use_device_trig(mode != 'walking' or poses is not None)
        t = 0
        while fused and t < T:
            chunk = rng.choice(18, size=(int(rng.choice([1, 7, 30])), n), p=WALK_W / WALK_W.sum()).astype(np.int32)
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
            if snapshot and t == T // 2:   # the complete state through a snapshot into a fresh env
                torch.cuda.synchronize()
                snap = env.state_dict()
                env = mk()
                env.load_state_dict(snap)
                if sampling:      # (generator settings are kernel parameters of the context, not state)
                    env.set_task_sampling(True, seed=samp_seed, n_tasks=ntasks)
                elif extra == 'random_tasks':
                    env.set_random_tasks(True, seed=samp_seed, **rk)
            da, oa = draw_actions(rng, mode, n)
            if subs:
                a = da.to(env.device)
                h = n // 2
                for k, sb in enumerate(subs):
                    sb.stream.wait_stream(torch.cuda.current_stream(env.device))   # the action upload, host-side resets
                    sb.step_walking_ptr(a[k * h:(k + 1) * h].contiguous())
                for sb in subs:
                    sb.join()
            else:
                env.step(da)
            oracle_step(ob, mode, oa, autoreset and not manual)
            if manual:
                if checker:
                    checker.step()
                torch.cuda.synchronize()
                done = ob.done.astype(bool)
                if not np.array_equal(env.done.cpu().numpy().astype(bool), done):
                    raise AssertionError(f'{desc} step {t}: done differs')
                if done.any() and (autoreset or host_resets):
                    if not autoreset:
                        env.reset(done)
                        torch.cuda.synchronize()
                    rew, dn = ob.reward.copy(), ob.done.copy()
                    give_oracle_tasks(done)
                    ob.reset(done)                 # (the batch arrays now hold the reset observation of those envs,
                    ob.reward[:], ob.done[:] = rew, dn   # reward / done stay the step's, as on the device)
                    if not autoreset:              # a host-side reset writes reward 0 / done 0 (igw_reset)
                        ob.reward[done], ob.done[done] = 0.0, 0
                    if checker:
                        checker.begin(done)
                if checker:
                    checker.check()
            compare(env, ob, f'{desc} step {t}')
    finally:
        O.use_device_trig(False)
        if checker:
            env.disable_trajectory_log()
    tags = ((' [fused replay]' if fused else '') + (f' [{checker.checked} logged episodes checked]' if checker else ''))
    print('ok', desc + tags, flush=True)
    return dict(mode=mode, extra=extra, split=split, snapshot=snapshot, fused=fused, host_resets=host_resets,
                logged=checker.checked if checker else 0)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    tally = {}
    for c in range(cases):
        r = run_case(c, rng)
        for k in ('mode:' + r['mode'], 'extra:' + r['extra'], 'split' if r['split'] else None, 'snapshot' if r['snapshot'] else None,
                  'fused' if r['fused'] else None, 'host_resets' if r['host_resets'] else None):
            if k:
                tally[k] = tally.get(k, 0) + 1
        tally['logged_episodes'] = tally.get('logged_episodes', 0) + r['logged']
    print('fuzz: all', cases, 'cases bit-exact;', ' '.join(f'{k}={v}' for k, v in sorted(tally.items())))
    return tally


if __name__ == '__main__':
    main()
