"""1-env, gym-shaped facade over the HIP path: the reference's `GridWorld`, `Wrapper`, `SizeReward`, `create_env` and
the two registered ids (gridworld/env.py:26-362) for render=False -- same construction kwargs, methods, attributes,
observation dict, dtypes and errors.

    env = gridworld_amd.make('IGLUGridworld-v0', vector_state=True, render=False)   # SizeReward(GridWorld(...))
    env.set_task(task); obs = env.reset(); obs, reward, done, info = env.step(action)
    env.unwrapped            # the GridWorld; .agent / .world / .grid / .step_no / .max_int as in the reference

It drives a VecGridWorld with N = 1 whose records, grid and action live in pinned host memory the kernel reads and
writes itself: a step is one kernel launch and one stream synchronisation, no copies; the observation arrays are
fresh numpy copies of that memory, as the reference's are fresh arrays."""
import ctypes as C
import warnings

import numpy as np
import torch

from . import _lib as L
from . import spaces
from .tasks import Task, Tasks
from .vec_env import VecGridWorld

_NO_TASK = ('Task is not initialized! Initialize task before working with the environment using .set_task '
            'method OR set tasks distribution using .set_task_generator method')
_TIS = (2, 4, 8, 12)


class AgentView:
    """Read-only view of the env's `Agent` (gridworld/core/world.py:8-29) as of the last reset / step: position
    (x, y, z), rotation (yaw, pitch), dy, time_int_steps, inventory (list of 6 ints), active_block, flying, strafe,
    sustain -- float64 / int values straight from the device's agent record (include/igw.h)."""
    PAD = 0.25
    __slots__ = ('_env',)

    def __init__(self, env):
        self._env = env

    def _f64(self):
        return self._env._host['agent'][:48].view(np.float64)

    @property
    def position(self):
        f = self._f64()
        return (float(f[0]), float(f[1]), float(f[2]))

    @property
    def rotation(self):
        f = self._f64()
        return (float(f[3]), float(f[4]))

    @property
    def dy(self):
        return float(self._f64()[5])

    @property
    def inventory(self):
        return [int(v) for v in self._env._host['agent'][48:60].view(np.int16)]

    @property
    def time_int_steps(self):
        return _TIS[int(self._env._host['agent'][62:64].view(np.uint16)[0]) & 3]

    @property
    def active_block(self):
        return (int(self._env._host['agent'][62:64].view(np.uint16)[0]) >> 2) & 7

    @property
    def flying(self):
        return self._env.action_space_type == 'flying'

    strafe = property(lambda self: [0, 0])   # reset at the end of every update (core/world.py:217-218)
    sustain = property(lambda self: False)
    reticle = property(lambda self: None)


class WorldView:
    """Read-only view of the env's `World` (gridworld/core/world.py:31-71): `world` = {(x, y, z): texture id} of the
    ground plane (37 x 37 at y = -2: WHITE -1 inside the build zone's footprint, GREY 0 around it, :60-71) and of the
    blocks in the build zone, `placed` = the set of the latter's positions; rebuilt from the grid on access."""
    __slots__ = ('_env',)
    initialized = True

    def __init__(self, env):
        self._env = env

    @property
    def placed(self):
        g = self._env.grid
        return {(int(x) - 5, int(y) - 1, int(z) - 5) for y, x, z in zip(*np.nonzero(g))}

    @property
    def world(self):
        w = {(x, -2, z): (-1 if abs(x) <= 5 and abs(z) <= 5 else 0) for x in range(-18, 19) for z in range(-18, 19)}
        g = self._env.grid
        for y, x, z in zip(*np.nonzero(g)):
            w[(int(x) - 5, int(y) - 1, int(z) - 5)] = int(g[y, x, z])
        return w

    def build_zone(self, x, y, z, pad=0):   # core/world.py:57-58
        return -5 - pad <= x <= 5 + pad and -5 - pad <= z <= 5 + pad and -1 - pad <= y < 8 + pad


class GridWorld:
    """The reference's GridWorld(Env) (gridworld/env.py:26-303), its own defaults included (select_and_place=False,
    discretize=False, vector_state=True -- create_env's differ, env.py:333-338)."""

    def __init__(self, render=True, max_steps=250, select_and_place=False, discretize=False, right_placement_scale=1.,
                 wrong_placement_scale=0.1, render_size=(64, 64), target_in_obs=False, action_space='walking',
                 vector_state=True, fake=False, name='', device='cuda:0'):
        if render and not fake:
            raise NotImplementedError('the renderer is out of scope of the MI355X step path; pass render=False '
                                      "(or use 'IGLUGridworldVector-v0')")
        self.vector_state, self.target_in_obs, self.fake, self.do_render = vector_state, target_in_obs, fake, render
        self.render_size, self.name = render_size, name
        self.max_steps, self.select_and_place, self.discretize = max_steps, select_and_place, discretize
        self.action_space_type = action_space
        # kept with the caller's types: the reference's reward is `int * scale` (env.py:293-296), a Python int
        # for right_placement_scale=1 and a float for wrong_placement_scale=0.1
        self.right_placement_scale, self.wrong_placement_scale = right_placement_scale, wrong_placement_scale
        self.right_placement = self.wrong_placement = 0
        # the device never computes SizeReward here: it is the reference's Python wrapper around this env (below)
        self._vec = VecGridWorld(1, device=device, action_space=action_space, select_and_place=select_and_place,
                                 size_reward=False, max_steps=max_steps, right_placement_scale=right_placement_scale,
                                 discretize=discretize, wrong_placement_scale=wrong_placement_scale, num_tasks=1,
                                 host_records=True)
        self._task = None
        self._task_generator = None
        self._overwrite_starting_grid = None
        self.initial_position = (0, 0, 0)
        self.initial_rotation = (0, 0)
        self.starting_grid = None
        self.max_int = 0
        self.prev_grid_size = 0
        self.renderer = None
        self.agent = AgentView(self)
        self.world = WorldView(self)
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
        self._init_host_path()

    # -- the host <-> device path of one step ---------------------------------------------------------------------
    def _init_host_path(self):
        """The env's records and grid live in pinned, device-mapped HOST memory (VecGridWorld(host_records=True)) and
        so does the action: the step kernel reads its 84 bytes of per-env input and writes its 144 bytes of records
        across PCIe itself, and a step costs the host one launch and one stream synchronisation -- no copy in either
        direction (a copy engine round trip costs more than the kernel).  The numpy views below alias that memory."""
        v = self._vec
        h = v.host_view.numpy()
        o, a, x = L.OUT_BYTES, L.AGENT_BYTES, L.AUX_BYTES
        self._host = {'out': h[:o], 'agent': h[o:o + a], 'aux': h[o + a:o + a + x],
                      'grid': h[o + a + x:o + a + x + L.CELLS].view(np.int8)}
        self._out_f32 = self._host['out'][:52].view(np.float32)
        self._aux_i16 = self._host['aux'][:8].view(np.int16)
        # the action, 40 bytes of pinned host memory: walking action i32 / Dict buttons u8[8] at 0, camera f32[2] at 8,
        # movement f32[3] at 16, inventory i32 at 28, placement i32 at 32
        self._act_pin = torch.zeros(40, dtype=torch.uint8).pin_memory()
        ah = self._act_pin.numpy()
        self._act = {'walk': ah[0:4].view(np.int32), 'buttons': ah[0:8], 'camera': ah[8:16].view(np.float32),
                     'movement': ah[16:28].view(np.float32), 'inventory': ah[28:32].view(np.int32),
                     'placement': ah[32:36].view(np.int32)}
        self._stream = torch.cuda.Stream(device=v.device)   # the env's own stream: step() never waits for other work
        self._stream_handle = C.c_void_p(self._stream.cuda_stream)
        p = self._act_pin.data_ptr()
        if v.flying:
            self._step_call = lambda: v.lib.igw_step_flying(v.ctx, p + 16, p + 8, p + 28, p + 32, self._stream_handle)
        elif v.walk_dict:
            self._step_call = lambda: v.lib.igw_step_walking_dict(v.ctx, p, p + 8, self._stream_handle)
        else:
            self._step_call = lambda: v.lib.igw_step_walking(v.ctx, p, self._stream_handle)
        self._read_back()   # a fresh Agent (core/world.py:12-29): inventory 20 x 6, BLUE active, time_int_steps 2

    def _device_step(self):
        rc = self._step_call()
        if rc:
            L.check(rc, 'igw_step (facade)')
        self._stream.synchronize()

    def _read_back(self):
        """After work issued through the VecGridWorld on the current stream (task upload, reset): wait for it."""
        torch.cuda.current_stream(self._vec.device).synchronize()

    @property
    def unwrapped(self):
        return self

    # -- the reference's public attributes (env.py:32-38), read-only views of the device state ----------------------
    @property
    def step_no(self):
        return int(self._host['agent'][60:62].view(np.uint16)[0])

    @property
    def grid(self):
        """int32 [9, 11, 11] copy of the world grid as of the last reset / step (GridWorld.grid, env.py:34)."""
        return self._host['grid'].reshape(9, 11, 11).astype(np.int32)

    # -- tasks (env.py:155-204) --
    def set_task(self, task):
        if self._task_generator is not None:
            warnings.warn('The .set_task method has no effect with an initialized tasks generator. '
                          'Drop it using .set_tasks_generator(None) after calling .set_task')
        self._task = task
        self.reset()

    def set_task_generator(self, task_generator):
        self._task_generator = task_generator
        self.reset()

    def initialize_world(self, starting_grid, initial_poisition):
        self._overwrite_starting_grid = starting_grid
        warnings.warn('Default task starting grid is overwritten using .initialize_world method. '
                      'Use .deinitialize_world to restore the original state.')
        self.initial_position = tuple(initial_poisition[:3])
        self.initial_rotation = tuple(initial_poisition[3:])
        self.reset()

    def deinitialize_world(self):
        self._overwrite_starting_grid = None
        self.initial_position = (0, 0, 0)
        self.initial_rotation = (0, 0)
        self.reset()

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

    def reset(self):
        if self._task is None:
            if self._task_generator is None:
                raise ValueError(_NO_TASK)
            self._task = self._task_generator.reset()
        elif self._task_generator is not None:
            self._task = self._task_generator.reset()
        self._task.reset()
        self._upload_task()
        self._vec.reset()
        # GridWorld.max_int (env.py:241): the user task -- full_grid admissibility included -- on the starting grid
        meta = self._vec.task_meta[0, 42:44]
        self._read_back()
        self.max_int = int(meta.cpu().numpy().view(np.int16)[0])
        obs = self._obs()
        self._counters = self._read_counters()
        self.prev_grid_size = int(np.count_nonzero(self._host['grid']))   # env.py:242
        return obs

    def _read_counters(self):
        """(max_int, prev_grid_size) of the synthetic task: the integers the reward is made of."""
        a = self._aux_i16
        return int(a[2]), int(a[1]) & 0x7fff

    def _obs(self):
        f = self._out_f32
        obs = {'inventory': f[5:11].copy(), 'compass': f[11:12].copy(), 'dialog': self._task.chat}
        if self.vector_state:
            obs['grid'] = self._host['grid'].reshape(9, 11, 11).astype(np.int32)
            obs['agentPos'] = f[0:5].copy()
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
        A = self._act
        if self.action_space_type == 'flying':
            inv = int(action['inventory'])
            if inv < 0 or inv > 6:
                raise ValueError(f'Bad inventory id: {inv}')  # core/world.py:354-355
            cam = np.asarray(action['camera'], np.float32)
            VecGridWorld._check_camera(cam)
            A['movement'][:] = np.asarray(action['movement'], np.float32)
            A['camera'][:] = cam
            A['inventory'][0] = inv
            A['placement'][0] = int(action['placement'])
        elif not self.discretize:
            hot = int(action['hotbar'])
            if hot < 0 or hot > 6:
                raise ValueError(f'Bad inventory id: {hot}')
            cam = np.asarray(action['camera'], np.float32)
            VecGridWorld._check_camera(cam)
            A['buttons'][:7] = [int(bool(action[k])) for k in ('forward', 'back', 'left', 'right', 'jump', 'attack', 'use')]
            A['buttons'][7] = hot
            A['camera'][:] = cam
        else:
            A['walk'][0] = int(action)
        self._device_step()
        obs = self._obs()
        # The reward as the reference's Python value: rebuilt from the device's integer counters with the
        # reference's own expression, so -1 * 0.1 is the double -0.1 (the device's float32 copy is only a
        # convenience for tensor consumers) and right * 1 stays an int.
        # (locals, as in the reference: GridWorld.right_placement / wrong_placement themselves are never updated by
        # step -- env.py:291 -- which is why SizeReward's penalty term is always 0)
        mi0, sz0 = self._counters
        self._counters = mi1, sz1 = self._read_counters()
        right, wrong = mi1 - mi0, sz0 - sz1                                                   # task.py:108-116
        if right == 0:
            reward = wrong * self.wrong_placement_scale                                       # env.py:293-296
        else:
            reward = right * self.right_placement_scale
        return obs, reward, bool(self._host['out'][52]), {}

    def render(self):
        raise ValueError('create env with render=True')


class Wrapper:
    """gridworld/env.py:306-314: attribute pass-through to the wrapped env."""

    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space

    def __getattr__(self, name):
        if name == 'env':
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self):
        return self.env.reset()

    def step(self, action):
        return self.env.step(action)

    def render(self, mode='human', **kwargs):
        return self.env.render()


class SizeReward(Wrapper):
    """gridworld/env.py:316-331, literally: the reward is the growth of max(GridWorld.max_int, size) -- and
    GridWorld.max_int is only evaluated at reset (env.py:241), so it is non-zero on the first step of an episode
    that starts with part of the target built -- plus min(GridWorld.wrong_placement * 0.02, 0), which is 0.0."""

    def __init__(self, env):
        super().__init__(env)
        self.size = 0

    def reset(self):
        self.size = 0
        return super().reset()

    def step(self, action):
        obs, reward, done, info = super().step(action)
        intersection = self.unwrapped.max_int
        reward = max(intersection, self.size) - self.size
        self.size = max(intersection, self.size)
        reward += min(self.unwrapped.wrong_placement * 0.02, 0)
        return obs, reward, done, info


def create_env(render=True, discretize=True, size_reward=True, select_and_place=True, right_placement_scale=1,
               render_size=(64, 64), target_in_obs=False, vector_state=False, max_steps=250, action_space='walking',
               wrong_placement_scale=0.1, name='', fake=False, device='cuda:0'):
    """gridworld/env.py:333-350 (same defaults)."""
    env = GridWorld(render=render, select_and_place=select_and_place, discretize=discretize,
                    right_placement_scale=right_placement_scale, wrong_placement_scale=wrong_placement_scale, name=name,
                    render_size=render_size, target_in_obs=target_in_obs, vector_state=vector_state, max_steps=max_steps,
                    action_space=action_space, fake=fake, device=device)
    if size_reward:
        env = SizeReward(env)
    return env


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


def register(gym_module=None):
    """Registers 'IGLUGridworld-v0' and 'IGLUGridworldVector-v0' (gridworld/env.py:352-362) with `gym_module` --
    anything with the classic `envs.register(id=, entry_point=, kwargs=)` -- or, by default, with classic `gym` when it
    is importable (what the reference registers with: its envs speak the old API -- `reset() -> obs`,
    `step() -> (obs, reward, done, info)`).  `gymnasium` is NOT registered with automatically: its `make` wraps every env
    in a passive checker and an order enforcer that expect the 5-tuple / `reset(seed=)` API; pass the module explicitly
    (`register(gymnasium)`) to register the ids there with both wrappers switched off.  Returns the names of the
    modules registered with.  An absent `gym` is the only thing ignored: a registry that rejects the ids raises."""
    mods = []
    if gym_module is not None:
        mods.append(gym_module)
    else:
        import importlib
        try:
            mods.append(importlib.import_module('gym'))
        except ImportError:
            pass
    for mod in mods:
        extra = {}
        if getattr(mod, '__name__', '').split('.')[0] == 'gymnasium':
            extra = dict(disable_env_checker=True, order_enforce=False)
        for env_id, kw in _REGISTRY.items():
            mod.envs.register(id=env_id, entry_point='gridworld_amd.env:create_env', kwargs=dict(kw), **extra)
    return [m.__name__ for m in mods]


try:
    register()
except Exception as _e:  # noqa: BLE001 -- a gym that is present but refuses the ids: say so, importing the package still works
    warnings.warn(f'gridworld_amd: registering the env ids with gym failed: {_e}')
