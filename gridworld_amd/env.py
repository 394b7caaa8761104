"""1-env, gym-shaped facade over the HIP path: same construction kwargs, methods, observation dict,
dtypes and errors as the reference's `GridWorld` + `SizeReward` + `create_env`
(gridworld/env.py:26-362) for render=False.  It drives a VecGridWorld with N = 1 and copies the
1.2 KB observation to the host each step, so it is for API parity, not for speed."""
import warnings

import numpy as np
import torch

from . import spaces
from .tasks import Task, Tasks
from .vec_env import VecGridWorld

_NO_TASK = ('Task is not initialized! Initialize task before working with the environment using .set_task '
            'method OR set tasks distribution using .set_task_generator method')


class GridWorld:
    """create_env(...) of the reference (defaults of gridworld/env.py:333-338)."""

    def __init__(self, render=True, discretize=True, size_reward=True, select_and_place=True,
                 right_placement_scale=1, render_size=(64, 64), target_in_obs=False, vector_state=False,
                 max_steps=250, action_space='walking', wrong_placement_scale=0.1, name='', fake=False,
                 device='cuda:0'):
        if render and not fake:
            raise NotImplementedError('the renderer is out of scope of the MI355X step path; pass render=False '
                                      "(or use 'IGLUGridworldVector-v0')")
        self.vector_state, self.target_in_obs, self.fake, self.do_render = vector_state, target_in_obs, fake, render
        self.render_size, self.name = render_size, name
        self.max_steps, self.select_and_place, self.discretize = max_steps, select_and_place, discretize
        self.action_space_type = action_space
        # kept with the caller's types: the reference's reward is `int * scale` (env.py:293-296), a Python int
        # for the default right_placement_scale=1 and a float for wrong_placement_scale=0.1
        self.right_placement_scale, self.wrong_placement_scale = right_placement_scale, wrong_placement_scale
        self.size_reward = bool(size_reward)
        self.right_placement = self.wrong_placement = 0
        self._vec = VecGridWorld(1, device=device, action_space=action_space, select_and_place=select_and_place,
                                 size_reward=size_reward, max_steps=max_steps,
                                 right_placement_scale=right_placement_scale, discretize=discretize,
                                 wrong_placement_scale=wrong_placement_scale, num_tasks=1)
        self._task = None
        self._task_generator = None
        self._overwrite_starting_grid = None
        self.initial_position = (0, 0, 0)
        self.initial_rotation = (0, 0)
        self.starting_grid = None
        self.max_int = 0
        self.prev_grid_size = 0
        if action_space == 'walking' and discretize:
            self.action_space = spaces.Discrete(18)
        elif action_space == 'walking':  # env.py:60-70
            self.action_space = spaces.Dict({
                'forward': spaces.Discrete(2), 'back': spaces.Discrete(2), 'left': spaces.Discrete(2),
                'right': spaces.Discrete(2), 'jump': spaces.Discrete(2), 'attack': spaces.Discrete(2),
                'use': spaces.Discrete(2), 'camera': spaces.Box(low=-5, high=5, shape=(2,)),
                'hotbar': spaces.Discrete(7)})
        elif action_space == 'flying':
            self.action_space = spaces.Dict({
                'movement': spaces.Box(low=-1, high=1, shape=(3,), dtype=np.float32),
                'camera': spaces.Box(low=-5, high=5, shape=(2,), dtype=np.float32),
                'inventory': spaces.Discrete(7), 'placement': spaces.Discrete(3)})
        else:
            raise ValueError(f'unknown action_space {action_space!r}')
        obs = {'inventory': spaces.Box(low=0, high=20, shape=(6,), dtype=np.float32),
               'compass': spaces.Box(low=-180, high=180, shape=(1,), dtype=np.float32), 'dialog': spaces.String()}
        if vector_state:
            obs['agentPos'] = spaces.Box(low=np.array([-8, -2, -8, -90, 0], dtype=np.float32),
                                         high=np.array([8, 12, 8, 90, 360], dtype=np.float32), shape=(5,))
            obs['grid'] = spaces.Box(low=-1, high=7, shape=(9, 11, 11), dtype=np.int32)
        if target_in_obs:
            obs['target_grid'] = spaces.Box(low=-1, high=7, shape=(9, 11, 11), dtype=np.int32)
        if render:
            obs['pov'] = spaces.Box(low=0, high=255, shape=(*render_size, 3), dtype=np.uint8)
        self.observation_space = spaces.Dict(obs)

    @property
    def unwrapped(self):
        return self

    # -- tasks (env.py:155-204) --
    def set_task(self, task):
        if self._task_generator is not None:
            warnings.warn('The .set_task method has no effect with an initialized tasks generator. '
                          'Drop it using .set_tasks_generator(None) after calling .set_task')
        self._task = task
        self._reset(keep_size=True)  # GridWorld.reset, not SizeReward.reset

    def set_task_generator(self, task_generator):
        self._task_generator = task_generator
        self._reset(keep_size=True)

    def initialize_world(self, starting_grid, initial_poisition):
        self._overwrite_starting_grid = starting_grid
        warnings.warn('Default task starting grid is overwritten using .initialize_world method. '
                      'Use .deinitialize_world to restore the original state.')
        self.initial_position = tuple(initial_poisition[:3])
        self.initial_rotation = tuple(initial_poisition[3:])
        self._reset(keep_size=True)

    def deinitialize_world(self):
        self._overwrite_starting_grid = None
        self.initial_position = (0, 0, 0)
        self.initial_rotation = (0, 0)
        self._reset(keep_size=True)

    @property
    def task(self):
        if self._task is None:
            if self._task_generator is None:
                raise ValueError(_NO_TASK)
            self._task = self._task_generator.reset()
            self.starting_grid = self._task.starting_grid
        return self._task

    # -- reset / step (env.py:206-303) --
    def _upload_task(self):
        t = self._task
        start = self._overwrite_starting_grid if self._overwrite_starting_grid is not None else t.starting_grid
        self.starting_grid = start
        pose = [[*self.initial_position, *self.initial_rotation]]
        self._vec.set_tasks(np.asarray(t.target_grid)[None], Tasks.to_dense(start)[None],
                            None if getattr(t, 'full_grid', None) is None else np.asarray(t.full_grid)[None],
                            invariant=getattr(t, 'invariant', True), init_pose=pose)

    def _reset(self, keep_size):
        if self._task is None:
            if self._task_generator is None:
                raise ValueError(_NO_TASK)
            self._task = self._task_generator.reset()
        elif self._task_generator is not None:
            self._task = self._task_generator.reset()
        self._task.reset()
        self._upload_task()
        self._vec.reset(keep_size=keep_size)
        obs = self._obs()
        self._counters = self._read_counters()
        # GridWorld.max_int (env.py:241): the user task -- full_grid admissibility included -- on the starting grid
        self.max_int = int(self._vec.task_meta[0, 42:44].cpu().numpy().view(np.int16)[0])
        self.prev_grid_size = int(np.count_nonzero(obs['grid'])) if self.vector_state else \
            int(np.count_nonzero(self._vec.grid[0].cpu().numpy()))   # env.py:242
        return obs

    def reset(self):
        return self._reset(keep_size=False)

    def _read_counters(self):
        """(max_int, prev_grid_size of the synthetic task, SizeReward.size): the integers the reward is made of."""
        st = self._vec.task_state()
        return int(st['max_int'][0]), int(st['prev_size'][0]), int(st['size'][0])

    def _obs(self):
        v = self._vec
        torch.cuda.synchronize(v.device)
        obs = {'inventory': v.inventory[0].cpu().numpy().astype(np.float32),
               'compass': v.compass.cpu().numpy().astype(np.float32),
               'dialog': self._task.chat}
        if self.vector_state:
            obs['grid'] = v.grid[0].cpu().numpy().astype(np.int32)
            obs['agentPos'] = v.agent_pos[0].cpu().numpy().astype(np.float32)
        if self.target_in_obs:
            obs['target_grid'] = np.asarray(self._task.target_grid).copy().astype(np.int32)
        if self.do_render:
            obs['pov'] = self.observation_space['pov'].sample()
        return obs

    def step(self, action):
        if self._task is None:
            if self._task_generator is None:
                raise ValueError(_NO_TASK)
            raise ValueError('Task is not initialized! Run .reset() first.')
        if self.action_space_type == 'flying':
            inv = int(action['inventory'])
            if inv < 0 or inv > 6:
                raise ValueError(f'Bad inventory id: {inv}')  # core/world.py:354-355
            a = {'movement': np.asarray(action['movement'], np.float32)[None],
                 'camera': np.asarray(action['camera'], np.float32)[None],
                 'inventory': np.array([inv], np.int32), 'placement': np.array([int(action['placement'])], np.int32)}
        elif not self.discretize:
            hot = int(action['hotbar'])
            if hot < 0 or hot > 6:
                raise ValueError(f'Bad inventory id: {hot}')
            b = [int(bool(action[k])) for k in ('forward', 'back', 'left', 'right', 'jump', 'attack', 'use')] + [hot]
            a = {'buttons': np.array([b], np.uint8), 'camera': np.asarray(action['camera'], np.float32)[None]}
        else:
            a = torch.tensor([int(action)], dtype=torch.int32)
        self._vec.step(a)
        obs = self._obs()
        # The reward as the reference's Python value: rebuilt from the device's integer counters with the
        # reference's own expression, so -1 * 0.1 is the double -0.1 (the device's float32 copy is only a
        # convenience for tensor consumers) and right * 1 stays an int.
        mi0, sz0, size0 = self._counters
        self._counters = mi1, sz1, size1 = self._read_counters()
        self.right_placement, self.wrong_placement = right, wrong = mi1 - mi0, sz0 - sz1   # task.py:108-116
        if right == 0:
            reward = wrong * self.wrong_placement_scale                                       # env.py:293-296
        else:
            reward = right * self.right_placement_scale
        if self.size_reward:   # SizeReward.step (env.py:325-331); GridWorld.wrong_placement stays 0 there (F6)
            reward = (size1 - size0) + min(0 * 0.02, 0)
        return obs, reward, bool(self._vec.done[0].item()), {}

    def render(self):
        raise ValueError('create env with render=True')


def create_env(**kwargs):
    return GridWorld(**kwargs)


_REGISTRY = {'IGLUGridworld-v0': {}, 'IGLUGridworldVector-v0': {'vector_state': True, 'render': False}}


def make(id, **kwargs):
    """gym.make for the two ids the reference registers (gridworld/env.py:352-362)."""
    if id not in _REGISTRY:
        raise KeyError(f'unknown env id {id!r}; known: {sorted(_REGISTRY)}')
    kw = dict(_REGISTRY[id])
    kw.update(kwargs)
    return create_env(**kw)


def make_vec(num_envs, device='cuda:0', **kwargs):
    """N envs with tensor observations (the fast path); kwargs as create_env."""
    kwargs.pop('render', None)
    kwargs.pop('vector_state', None)
    return VecGridWorld(num_envs, device=device, **kwargs)


def _register_with_gym():
    for modname in ('gymnasium', 'gym'):
        try:
            mod = __import__(modname)
            for env_id, kw in _REGISTRY.items():
                mod.envs.register(id=env_id, entry_point='gridworld_amd.env:create_env', kwargs=kw)
        except Exception:
            pass


_register_with_gym()
