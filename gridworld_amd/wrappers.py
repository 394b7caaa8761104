"""Wrappers of the reference that sit on the step path (gridworld/wrappers.py)."""
from . import spaces
from .env import Wrapper  # noqa: F401  (gridworld/env.py:306-314; re-exported: the reference imports gym.Wrapper here)


class Actions(Wrapper):
    """Discrete(17) action set without `place` (gridworld/wrappers.py:11-32): new index -> Discrete(18) index.
    With select_and_place the hotbar actions place blocks, so index 17 (use) is redundant."""

    def __init__(self, env):
        super().__init__(env)
        self.action_map = list(range(17))  # 0 noop, 1-4 move, 5 jump, 6-11 hotbar, 12-15 camera, 16 break
        self.action_space = spaces.Discrete(len(self.action_map))

    def step(self, action):
        return self.env.step(self.action_map[action])


class EpisodeLogger:
    """Per-episode npz dumps of what the reference's `Logged` wrapper collects (gridworld/wrappers.py:66-134:
    every observation key stacked over reset + steps, `reward`, `done`, the actions as csv; no video), taken
    from the trajectory log the step kernels keep on the device (VecGridWorld.enable_trajectory_log,
    include/igw.h: igw_set_trajectory_log) -- the env loop itself never copies observations to the host.

        log = EpisodeLogger(vec_env, n_envs=4, path='episodes')
        ... step / reset as usual ...; files = log.collect()      # any time; dumps the episodes that finished

    Arrays per file (T = steps of the episode): agentPos f32[T+1,5], inventory f32[T+1,6], compass f32[T+1,1],
    grid int32[T+1,9,11,11] (rebuilt from the starting grid and the logged one-cell changes), reward f64[T],
    done bool[T], plus `task` (row of the task table), `env`, `episode`.  Entry 0 is the reset observation.

    `actions` are the actions AS EXECUTED, not the raw inputs the reference's Logged appends (wrappers.py:98): the
    device record packs a flying action's inventory into 3 bits and its placement into 2 (include/igw.h), so an
    inventory id the kernel rejected (outside 0..6: run as 0 and counted in stats()['bad_actions']) is logged as 0 and
    a placement other than 1 / 2 as 0.  Replaying a log therefore reproduces the trajectory, not the bad-action
    counts; Discrete(18) ids, movement / camera floats and Dict buttons are logged raw."""

    def __init__(self, vec, n_envs=1, path='episodes', desc='', glob_step=0, capacity=None):
        import numpy as np
        self.np = np
        self.vec, self.path, self.desc, self.glob_step = vec, path, desc, glob_step
        self.records, self.heads = vec.enable_trajectory_log(n_envs, capacity)
        self.n_envs = int(n_envs)
        self._dumped = {}  # env -> last episode number written
        self.episodes = []  # the dicts that were dumped (also returned by collect)

    def set_path(self, path):
        self.path = path

    def set_desc(self, desc, glob_step):
        self.desc, self.glob_step = desc, glob_step

    def _decode(self, env, slot, task, length):
        np, v = self.np, self.vec
        from . import _lib as L
        raw = self.records[env, slot, :length].cpu().numpy()   # record layout: include/igw.h (igw_set_trajectory_log)
        f32 = raw[:, :28].copy().view(np.float32).reshape(length, 7)
        inv = raw[:, 28:40].copy().view(np.int16).reshape(length, 6)
        change = raw[:, 40:42].copy().view(np.uint16)[:, 0].astype(np.int64)
        tag = raw[:, 43].astype(np.int64)
        meta = v.task_meta[task].cpu().numpy()
        inv0 = meta[64:76].view(np.int16).astype(np.float32)
        grid0 = v.task_start[task, :L.CELLS].cpu().numpy().astype(np.int32).reshape(9, 11, 11)
        grids = np.empty((length + 1, 9, 11, 11), np.int32)
        grids[0] = grid0
        for t in range(length):
            grids[t + 1] = grids[t]
            if change[t] != 0xffff:
                grids[t + 1].reshape(-1)[change[t] & 0x7ff] = (change[t] >> 11) & 7
        space = int(tag[0] & 3) if length else 0
        if space == 0:
            actions = raw[:, 44:48].copy().view(np.int32)[:, 0]
        elif space == 1:
            actions = {'movement': raw[:, 44:56].copy().view(np.float32).reshape(length, 3),
                       'camera': raw[:, 56:64].copy().view(np.float32).reshape(length, 2),
                       'inventory': ((tag >> 2) & 7).astype(np.int32), 'placement': ((tag >> 5) & 3).astype(np.int32)}
        else:
            actions = {'buttons': raw[:, 44:52].copy(), 'camera': raw[:, 52:60].copy().view(np.float32).reshape(length, 2)}
        zeros = np.zeros((1, 5), np.float32)
        return {'agentPos': np.concatenate([zeros, f32[:, :5]]),
                'inventory': np.concatenate([inv0[None], inv.astype(np.float32)]),
                'compass': np.concatenate([np.zeros((1, 1), np.float32), f32[:, 6:7]]),
                'grid': grids, 'reward': f32[:, 5].astype(np.float64), 'done': raw[:, 42].astype(bool),
                'actions': actions, 'task': int(task)}

    def collect(self, dump=True):
        """Decodes (and with dump=True writes) every logged episode that finished since the last call."""
        import os
        import uuid
        np = self.np
        import torch
        torch.cuda.synchronize(self.vec.device)
        heads = self.heads.cpu().numpy()
        out = []
        for env in range(self.n_envs):
            for slot in (0, 1):
                task, length, episode, finished = (int(x) for x in heads[env, slot])
                if not finished or length == 0 or self._dumped.get(env, -1) >= episode:
                    continue
                ep = self._decode(env, slot, task, length)
                ep.update(env=env, episode=episode)
                self._dumped[env] = max(self._dumped.get(env, -1), episode)
                if dump:
                    d = f'{self.path}/step{self.glob_step}'
                    os.makedirs(d, exist_ok=True)
                    fname = f'{d}/ep_{self.desc}_{uuid.uuid4().hex[:6]}'
                    arrays = {k: v for k, v in ep.items() if k != 'actions'}
                    if isinstance(ep['actions'], dict):
                        arrays.update({'action_' + k: v for k, v in ep['actions'].items()})
                    np.savez_compressed(fname + '.npz', **arrays)
                    if not isinstance(ep['actions'], dict):
                        with open(fname + '.csv', 'w') as f:
                            for a in ep['actions']:
                                f.write(f'{int(a)}\n')
                    ep['file'] = fname + '.npz'
                out.append(ep)
        self.episodes.extend(out)
        return out


class Logged(Wrapper):
    """The reference's Logged wrapper for the 1-env facade (gridworld/wrappers.py:66-134), without the renderer:
    turn_on() / set_path() / set_desc(); an episode is written when it ends while logging is on."""

    def __init__(self, env):
        super().__init__(env)
        self.logging = False
        self.turned_off = True
        self._log = EpisodeLogger(env.unwrapped._vec, 1, path='episodes')

    def turn_on(self):
        self.turned_off = False
        self.logging = True

    def set_path(self, path):
        self._log.set_path(path)

    def set_desc(self, desc, glob_step):
        self._log.set_desc(desc, glob_step)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        if done:
            self._log.collect(dump=self.logging)
        return obs, reward, done, info
