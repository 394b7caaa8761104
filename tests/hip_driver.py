"""gridworld_amd.VecGridWorld behind the golden-replay driver interface (GPU)."""
import numpy as np
import torch

from gridworld_amd import VecGridWorld


class HipDriver:
    def __init__(self, fx, lanes_per_env=0, **extra):
        kw = dict(fx['kwargs'])
        kw.update(extra)
        self.env = VecGridWorld(len(fx['targets']), lanes_per_env=lanes_per_env, **kw)
        self._pose = None
        self._tasks = None

    def set_tasks(self, targets, starts, invariant=True):
        self._tasks = (targets, starts, invariant)
        self.env.set_tasks(targets, starts, invariant=invariant, init_pose=self._pose)

    def set_initial_pose(self, poses):
        self._pose = np.asarray(poses, np.float64)
        t, s, inv = self._tasks
        self.env.set_tasks(t, s, invariant=inv, init_pose=self._pose)

    def reset(self, mask):
        self.env.reset(mask)

    def step_walking(self, actions):
        self.env.step(torch.as_tensor(np.asarray(actions, np.int32)))

    def step_flying(self, mv, cam, inv, place):
        self.env.step(dict(movement=np.asarray(mv, np.float32), camera=np.asarray(cam, np.float32),
                           inventory=np.asarray(inv, np.int32), placement=np.asarray(place, np.int32)))

    def step_walking_dict(self, buttons, cam):
        self.env.step(dict(buttons=np.asarray(buttons, np.uint8), camera=np.asarray(cam, np.float32)))

    def set_task_table(self, targets, starts, full_grids):
        """One row per (env, episode) task of a Subtasks fixture; needs num_tasks >= len(targets)."""
        self.env.set_tasks(targets, starts, full_grids=full_grids, env_task=np.zeros(self.env.num_envs, np.int32))
        self._with_table = True

    def assign_tasks(self, mask, rows):
        et = self.env.env_task.cpu().numpy()
        et[np.asarray(mask, bool)] = np.asarray(rows, np.int32)[np.asarray(mask, bool)]
        self.env.env_task.copy_(torch.as_tensor(et))

    def outputs(self):
        e = self.env
        torch.cuda.synchronize()
        extra = {}
        if getattr(self, '_with_table', False):
            meta = e.task_meta.cpu().numpy()[e.env_task.cpu().numpy().astype(np.int64)]
            extra = dict(env_max_int=meta[:, 42:44].copy().view(np.int16)[:, 0].astype(np.int64),
                         syn_max_int=e.task_state()['max_int'])
        return dict(**extra, agentPos=e.agent_pos.cpu().numpy(), inventory=e.inventory.cpu().numpy(),
                    compass=e.compass.cpu().numpy(), reward=e.reward.cpu().numpy(),
                    done=e.done.cpu().numpy(), grid=e.grid.cpu().numpy().reshape(e.num_envs, -1),
                    internal=e.internals())
