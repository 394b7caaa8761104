"""Batched IGLU gridworld on one MI355X: N independent envs whose state IS a set of torch
tensors in HBM, stepped by the HIP kernels behind the C ABI (include/igw.h).

Mirrors the reference's env protocol for render=False / vector_state=True
(gridworld/env.py:155-303): set_tasks -> reset -> step, with the create_env kwargs of
gridworld/env.py:333-338.  Observations are tensor views of the state, not copies.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _as_rows(x, device, n=None):
    """[T,9,11,11] / [T,1089] / [9,11,11] int array -> int8 tensor [T, GRID_STRIDE] on device."""
    if x is None:
        return None
    t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    t = t.reshape(t.shape[0], -1)
    if t.shape[1] == L.GRID_STRIDE:
        return t.to(device=device, dtype=torch.int8).contiguous()
    if t.shape[1] != L.CELLS:
        raise ValueError(f'grid rows must have 1089 cells, got {t.shape[1]}')
    out = torch.zeros((t.shape[0], L.GRID_STRIDE), dtype=torch.int8, device=device)
    out[:, :L.CELLS] = t.to(device=device, dtype=torch.int8)
    return out


class VecGridWorld:
    """N envs on one GPU.  kwargs follow create_env (gridworld/env.py:333-338)."""

    def __init__(self, num_envs, device='cuda:0', action_space='walking', select_and_place=True,
                 size_reward=True, max_steps=250, right_placement_scale=1., wrong_placement_scale=0.1,
                 discretize=True, autoreset=False, num_tasks=None, lanes_per_env=0, debug_flags=0, env_index_base=0,
                 host_records=False, render=False, render_size=(64, 64), target_in_obs=False, vector_state=True, name='', fake=False):
        """create_env's keyword arguments (gridworld/env.py:333-338) plus the batch's own: num_envs, device,
        autoreset (reset inside step), num_tasks (rows of the task table, default num_envs), lanes_per_env
        (0 = automatic), env_index_base (global index of env 0: rank / sub-batch offset), debug_flags (IGW_DIAG
        build only), host_records (the output / agent / aux records and the grid live in PINNED HOST memory that the
        kernels read and write across PCIe: a host-side consumer of a FEW envs -- the 1-env gym facade -- then needs no
        copy at all, only a stream synchronisation; the observation tensors are CPU tensors in that case).  Anything else is a TypeError, as in create_env.  render / render_size / fake / name /
        target_in_obs / vector_state are accepted for signature compatibility: this is the render=False,
        vector_state=True path (observations are the state tensors; `targets()` gives the target grids)."""
        if render and not fake:
            raise NotImplementedError('the renderer is out of scope of the MI355X step path; pass render=False')
        if not torch.cuda.is_available():
            raise L.IgwError('VecGridWorld needs a HIP device (no CPU fallback)')
        if action_space not in ('walking', 'flying'):
            raise ValueError(f'unknown action_space {action_space!r}')
        self.lib = L.load()
        self.device = torch.device(device)
        self.num_envs = int(num_envs)
        self.num_tasks = int(num_tasks or num_envs)
        self.flying = action_space == 'flying'
        self.walk_dict = action_space == 'walking' and not discretize
        self.max_steps = int(max_steps)
        self.autoreset = bool(autoreset)
        N, T, dev = self.num_envs, self.num_tasks, self.device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        # What a host-side consumer reads back after a step -- output, agent and aux records, the grid -- comes from ONE
        # allocation (`host_view`), so the 1-env facade moves it with a single device-to-host copy (gridworld_amd/env.py)
        rec = L.OUT_BYTES + L.AGENT_BYTES + L.AUX_BYTES
        self.host_records = bool(host_records)
        if self.host_records:   # pinned, device-mapped host memory (hipHostMalloc): the same address on both sides
            self.host_view = torch.zeros((N * (rec + L.GRID_STRIDE),), dtype=torch.uint8).pin_memory()
        else:
            self.host_view = z((N * (rec + L.GRID_STRIDE),), torch.uint8)
        cut = lambda lo, width: self.host_view[N * lo:N * (lo + width)].view(N, width)  # noqa: E731
        self.out_buf = cut(0, L.OUT_BYTES)         # agentPos, inventory, compass, reward, done of every step
        self.agent_buf = cut(L.OUT_BYTES, L.AGENT_BYTES)   # pose, inventory, step_no, pack (include/igw.h)
        self.aux_buf = cut(L.OUT_BYTES + L.AGENT_BYTES, L.AUX_BYTES)   # size, prev_size, max_int, target_size, task, episode
        self.grid_buf = cut(rec, L.GRID_STRIDE).view(torch.int8)
        self.occ_buf = z((N, L.OCC_WORDS), torch.int32)
        self.hist_buf = z((N, L.HIST_ROW), torch.int16)
        self.task_target = z((T, L.GRID_STRIDE), torch.int8)
        self.task_start = z((T, L.GRID_STRIDE), torch.int8)
        self.task_start_occ = z((T, L.OCC_WORDS), torch.int32)
        self.task_meta = z((T, L.TASK_META_BYTES), torch.uint8)
        self.task_index = z((T, L.TASK_INDEX_BYTES), torch.uint8)   # colour index of the synthetic targets (include/igw.h)
        self.stats_buf = z((L.STAT_STRIPES, 8), torch.int64)
        self._make_views()
        # Agent.__init__ (core/world.py:12-29): time_int_steps = 2, active_block = BLUE, inventory 20
        self.agent_buf.view(torch.int16)[:, 24:30] = 20
        self.agent_buf[:, 62] = 1 << 2  # u16 pack: time_int_steps code 0 (= 2), active_block 1
        self.cfg = L.Config(dev.index or 0, N, T,
                            L.FLYING if self.flying else L.WALKING_DICT if self.walk_dict else L.WALKING_DISCRETE,
                            int(select_and_place), int(size_reward), self.max_steps, int(autoreset),
                            float(right_placement_scale), float(wrong_placement_scale), int(lanes_per_env),
                            int(debug_flags), int(env_index_base))
        self.env_index_base = int(env_index_base)
        self.ctx = C.c_void_p()
        L.check(self.lib.igw_create(C.byref(self.cfg), C.byref(self.ctx)), 'igw_create')
        b = L.Buffers(*[t.data_ptr() for t in (self.grid_buf, self.occ_buf, self.hist_buf, self.agent_buf, self.aux_buf,
                                                self.task_target, self.task_start, self.task_start_occ, self.task_meta,
                                                self.task_index, self.out_buf, self.stats_buf)])
        L.check(self.lib.igw_bind_buffers(self.ctx, C.byref(b)), 'igw_bind_buffers')
        self.user_target = None
        self._have_tasks = False
        self._tasks_filled = 0       # rows of the task table written so far (what task sampling draws from)
        self._sampling = None        # (seed,) / ('random', kwargs) of the device-side generator, for sub-batches
        self._traj = None
        self._children = []
        # bumped by every call that changes what a step LAUNCH is given by value (the kernel parameters: sampler
        # settings, the episode-log buffers): a captured StepGraph carries the values of its capture
        self.config_epoch = 0

    def __del__(self):
        ctx = getattr(self, 'ctx', None)
        if ctx:
            self.lib.igw_destroy(ctx)
            self.ctx = None

    def _make_views(self):
        """The observation / result tensors of the env protocol (env.py:281-303) and the per-env task row / episode
        counter as strided VIEWS of the records the kernels read and write (include/igw.h): nothing is copied.
        They are NOT dense [N] tensors: agent_pos [N,5], inventory [N,6], compass [N], reward [N] are float32 views
        with a row stride of 16 elements (the 64-byte output record), done [N] is uint8 with stride 64, env_task /
        episode int32 with stride 4, grid [N,9,11,11] int8 with row stride 1104.  A consumer that needs packed
        memory (.view(), DLPack, a custom kernel indexing [i]) takes `dense()` or calls .contiguous()."""
        N = self.num_envs
        f = self.out_buf.view(torch.float32)          # [N, 16]
        self.agent_pos = f[:, 0:5]                    # x, y, z, pitch, yaw
        self.inventory = f[:, 5:11]
        self.compass = f[:, 11]
        self.reward = f[:, 12]
        self.done = self.out_buf[:, 52]
        a = self.aux_buf.view(torch.int32)            # [N, 4]
        self.env_task = a[:, 2]                       # row of the task table
        self.episode = a[:, 3]                        # episodes started (keys the device-side task generators)
        self.grid = torch.as_strided(self.grid_buf, (N, 9, 11, 11), (L.GRID_STRIDE, 121, 11, 1))
        self._obs = {'agentPos': self.agent_pos, 'inventory': self.inventory, 'compass': self.compass.unsqueeze(1),
                     'grid': self.grid}
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

    def _stream(self):
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(self._dev_index))

    # ---- tasks (GridWorld.set_task / Task.__init__ / initialize_world) ----
    def set_tasks(self, targets, starts=None, full_grids=None, invariant=None, init_pose=None,
                  env_task=None, first=0):
        """Fills task-table rows [first, first+T).  targets/starts/full_grids: dense [T,9,11,11];
        invariant: bool or [T]; init_pose: [T,5] (x,y,z,yaw,pitch).  Does not reset."""
        dev = self.device
        tgt = _as_rows(targets, dev)
        T = tgt.shape[0]
        st = _as_rows(starts, dev)
        fg = _as_rows(full_grids, dev)
        inv = None
        if invariant is not None:
            inv = torch.as_tensor(np.broadcast_to(np.asarray(invariant, dtype=np.uint8), (T,)).copy(), device=dev)
        for name, g, hi in (('targets', tgt, 7), ('starts', st, 6), ('full_grids', fg, 7)):
            # block ids are 0..7 (the observation space's range, env.py:85); a starting block is taken off the inventory
            # of its colour, ids 1..6 (env.py:243-246: the reference raises IndexError for 7).  The C ABI counts
            # offending rows (IGW_STAT_BAD_TASK) and reads such a cell as empty.
            if g is not None and g.numel() and (int(g.min()) < 0 or int(g.max()) > hi):
                raise ValueError(f'{name}: block ids must be in 0..{hi}')
        if first < 0 or first + T > self.num_tasks:
            raise ValueError(f'task rows [{first}, {first + T}) do not fit the table of {self.num_tasks}')
        for name, g in (('starts', st), ('full_grids', fg)):
            if g is not None and g.shape[0] != T:
                raise ValueError(f'{name} has {g.shape[0]} rows, targets {T}')
        pose = None
        if init_pose is not None:
            pose_np = np.asarray(init_pose, dtype=np.float64).reshape(T, 5)
            # the kernels index the world around the agent without range checks; that holds for |x|, |z| <= 10
            # (the C ABI replaces anything else by the default pose and counts it, IGW_STAT_BAD_POSE)
            ok = np.isfinite(pose_np).all() and (np.abs(pose_np[:, [0, 2]]) <= 10).all() and \
                (np.abs(pose_np[:, 1]) <= 64).all() and (np.abs(pose_np[:, 3:]) <= 1e6).all()
            if not ok:
                raise ValueError('init_pose must be finite with |x|, |z| <= 10, |y| <= 64 and |yaw|, |pitch| <= 1e6')
            pose = torch.as_tensor(pose_np, device=dev).contiguous()
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        L.check(self.lib.igw_prepare_tasks(self.ctx, first, T, ptr(tgt), ptr(st), ptr(fg), ptr(inv), ptr(pose),
                                           self._stream()), 'igw_prepare_tasks')
        if self.user_target is None:
            self.user_target = torch.zeros((self.num_tasks, L.GRID_STRIDE), dtype=torch.int8, device=dev)
        self.user_target[first:first + T] = tgt
        self._tasks_filled = max(self._tasks_filled, first + T)
        if env_task is not None:
            et = np.asarray(env_task.cpu() if torch.is_tensor(env_task) else env_task).reshape(-1)
            if et.shape[0] != self.num_envs or et.min() < 0 or et.max() >= self.num_tasks:
                raise ValueError(f'env_task needs {self.num_envs} indices into the task table [0, {self.num_tasks})')
            self.env_task.copy_(torch.as_tensor(et.astype(np.int32), device=dev))
        elif first == 0:
            if T == self.num_envs:
                self.env_task.copy_(torch.arange(self.num_envs, dtype=torch.int32, device=dev))
            elif T == 1:
                self.env_task.zero_()
        self._keep = (tgt, st, fg, inv, pose)  # keep inputs alive until the async kernel ran
        self._have_tasks = True

    def set_task_sampling(self, enabled=True, seed=0, n_tasks=None):
        """Draw every env's task uniformly from the filled rows of the task table at each reset / auto-reset, on
        the device (CustomTasks.reset semantics; counter RNG keyed by seed, global env index and the env's
        episode counter, so it also advances inside a replayed HIP graph)."""
        n = int(n_tasks if n_tasks is not None else self._tasks_filled)
        if enabled and not 0 < n <= self.num_tasks:
            raise ValueError('set_task_sampling needs filled task rows: call set_tasks first')
        L.check(self.lib.igw_set_task_sampling(self.ctx, int(bool(enabled)), int(seed), n), 'igw_set_task_sampling')
        self._sampling = ('table', int(seed), n) if enabled else None
        self.config_epoch += 1
        for c in self._children:
            c._inherit_sampling()

    def set_random_tasks(self, enabled=True, seed=0, max_blocks=4, height_levels=1, max_dist=2, num_colors=1):
        """RandomTasks(max_blocks, height_levels, max_dist=.., num_colors=..) (gridworld/tasks/task_set.py:59-157)
        with sample_task() on the device: every reset / auto-reset writes a freshly sampled target into the env's
        own task row -- no host in the reset path.  Needs num_tasks == num_envs (the default)."""
        kw = dict(max_blocks=int(max_blocks), height_levels=int(height_levels), max_dist=int(max_dist),
                  num_colors=int(num_colors))
        L.check(self.lib.igw_set_random_tasks(self.ctx, int(bool(enabled)), int(seed), kw['max_blocks'],
                                              kw['height_levels'], kw['max_dist'], kw['num_colors'], self._stream()),
                'igw_set_random_tasks')
        self._sampling = ('random', int(seed), kw) if enabled else None
        self.config_epoch += 1
        if enabled:
            self._have_tasks = True
            self._tasks_filled = max(self._tasks_filled, self.num_envs)
        for c in self._children:
            c._inherit_sampling()

    def targets(self):
        """Current synthetic target (target - start) of every env, [N, 9, 11, 11] int8 (a gather from the task table)."""
        rows = self.task_target[self.env_task.long()]
        return rows[:, :L.CELLS].reshape(self.num_envs, 9, 11, 11)

    # ---- episode log (the reference's Logged wrapper, gridworld/wrappers.py:66-134, without video) ----
    def enable_trajectory_log(self, n_envs=1, capacity=None):
        """The first n_envs envs record every step on the device (one 64-byte record per step, two episode slots
        per env); see gridworld_amd.wrappers.EpisodeLogger for the npz dumps."""
        n_envs = int(n_envs)
        cap = int(capacity or self.max_steps)
        rec = torch.zeros((n_envs, 2, cap, L.TRAJ_BYTES), dtype=torch.uint8, device=self.device)
        heads = torch.zeros((n_envs, 2, 4), dtype=torch.int32, device=self.device)
        L.check(self.lib.igw_set_trajectory_log(self.ctx, rec.data_ptr(), heads.data_ptr(), n_envs, cap),
                'igw_set_trajectory_log')
        self._traj = (rec, heads, n_envs, cap)
        self.config_epoch += 1
        return rec, heads

    def disable_trajectory_log(self):
        L.check(self.lib.igw_set_trajectory_log(self.ctx, None, None, 0, 0), 'igw_set_trajectory_log')
        self._traj = None
        self.config_epoch += 1

    _STATE_KEYS = ('grid_buf', 'occ_buf', 'hist_buf', 'agent_buf', 'aux_buf', 'out_buf', 'task_target', 'task_start',
                   'task_start_occ', 'task_meta', 'task_index', 'stats_buf')

    def state_dict(self):
        """Snapshot of the complete env state (tensors are cloned): resume / parity debugging."""
        d = {k: getattr(self, k).clone() for k in self._STATE_KEYS}
        d['abi_version'] = L.VERSION
        return d

    def load_state_dict(self, state):
        """Restores a state_dict() snapshot.  The key set must be complete and of this ABI version: the step kernels
        read every one of these buffers (a snapshot without the colour index, say, would load and then drift)."""
        if state.get('abi_version') != L.VERSION:
            raise ValueError(f'state_dict is of ABI version {state.get("abi_version")!r}, this build is version {L.VERSION}')
        missing = [k for k in self._STATE_KEYS if k not in state]
        extra = [k for k in state if k not in self._STATE_KEYS and k != 'abi_version']
        if missing or extra:
            raise ValueError(f'state_dict keys do not match: missing {missing}, unexpected {extra}')
        for k in self._STATE_KEYS:
            if tuple(state[k].shape) != tuple(getattr(self, k).shape):
                raise ValueError(f'state_dict[{k!r}] has shape {tuple(state[k].shape)}, expected {tuple(getattr(self, k).shape)}')
        for k in self._STATE_KEYS:
            getattr(self, k).copy_(state[k])
        self._have_tasks = True

    # ---- reset / step ----
    def _need_tasks(self):
        if not self._have_tasks:
            raise ValueError('Task is not initialized! Initialize task before working with the environment '
                             'using .set_tasks')

    def obs(self):
        return self._obs.copy()

    def dense(self):
        """Packed COPIES of the per-step results for consumers that cannot take strided views (see _make_views):
        dict(agentPos [N,5], inventory [N,6], compass [N], reward [N], done [N] uint8), each contiguous."""
        return {'agentPos': self.agent_pos.contiguous(), 'inventory': self.inventory.contiguous(),
                'compass': self.compass.contiguous(), 'reward': self.reward.contiguous(), 'done': self.done.contiguous()}

    def reset(self, mask=None, keep_size=False):
        self._need_tasks()
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        L.check(self.lib.igw_reset(self.ctx, None if m is None else C.c_void_p(m.data_ptr()),
                                   L.RESET_KEEP_SIZE if keep_size else 0, self._stream()), 'igw_reset')
        self._mask_keep = m
        return self.obs()

    @staticmethod
    def _check_camera(cam):
        """Host-side mirror of the kernels' camera screen (IGW_CAMERA_MAX): camera deltas that arrive as HOST data
        are checked here and raise; device tensors are not read back (that would synchronise every step) -- the
        kernels run an offending component as a no-op and count it in stats()['bad_actions']."""
        if torch.is_tensor(cam):
            if cam.is_cuda:
                return
            c = cam.detach().numpy()
        else:
            c = np.asarray(cam)
        c = c.astype(np.float64, copy=False)
        if c.size and not (np.isfinite(c).all() and np.abs(c).max() <= L.CAMERA_MAX):
            raise ValueError(f'camera deltas must be finite with |value| <= {L.CAMERA_MAX:g} degrees')

    def step(self, actions):
        """walking: int32 tensor [N]; walking with discretize=False: dict(buttons u8[N,8] = forward, back,
        left, right, jump, attack, use, hotbar -- or those eight keys separately -- and camera f32[N,2]);
        flying: dict(movement f32[N,3], camera f32[N,2], inventory i32[N], placement i32[N]).
        Returns (obs, reward, done, info) of tensors living in HBM -- STRIDED views of the kernels' 64-byte output
        record (reward: float32, stride 16 elements; done: uint8, stride 64; see _make_views / dense()), the same
        tensors every call.  Actions that already are contiguous device tensors of the kernel's dtypes (walking: int32
        [N]; flying: float32 / int32) go straight to the C ABI: no conversion, no copy, one ctypes call."""
        if not self._have_tasks:
            self._need_tasks()
        N = self.num_envs
        if not (self.walk_dict or self.flying):
            a = actions
            if not (type(a) is torch.Tensor and a.dtype is torch.int32 and a.is_cuda and a.is_contiguous()):
                a = torch.as_tensor(actions, device=self.device).to(torch.int32).contiguous()
            if a.numel() != N:
                raise ValueError(f'walking action needs {N} entries, got {a.numel()}')
            rc = self.lib.igw_step_walking(self.ctx, a.data_ptr(), torch._C._cuda_getCurrentRawStream(self._dev_index))
            if rc:
                L.check(rc, 'igw_step_walking')
            self._act_keep = a
            return self._obs.copy(), self.reward, self.done, {}
        dev = self.device
        self._check_camera(actions['camera'])

        def dev_t(x, dt):   # already a contiguous device tensor of the kernel's dtype: as is
            if type(x) is torch.Tensor and x.dtype is dt and x.is_cuda and x.is_contiguous():
                return x
            return torch.as_tensor(x, device=dev).to(dt).contiguous()
        if self.walk_dict:
            if 'buttons' in actions:
                b = dev_t(actions['buttons'], torch.uint8).reshape(-1, 8)
            else:  # the reference's Dict keys (env.py:60-70), one array per key
                b = torch.stack([torch.as_tensor(np.asarray(actions[k]), device=dev).to(torch.uint8).reshape(-1)
                                 for k in ('forward', 'back', 'left', 'right', 'jump', 'attack', 'use', 'hotbar')],
                                dim=1).contiguous()
            if b.shape[0] != N:
                raise ValueError(f'walking Dict action needs {N} rows of 8 buttons, got {tuple(b.shape)}')
            cam = dev_t(actions['camera'], torch.float32)
            if cam.numel() != 2 * N:
                raise ValueError(f'walking Dict action needs camera [{N},2], got {tuple(cam.shape)}')
            L.check(self.lib.igw_step_walking_dict(self.ctx, b.data_ptr(), cam.data_ptr(), self._stream()),
                    'igw_step_walking_dict')
            self._act_keep = (b, cam)
        else:
            mv, cam = dev_t(actions['movement'], torch.float32), dev_t(actions['camera'], torch.float32)
            inv, pl = dev_t(actions['inventory'], torch.int32), dev_t(actions['placement'], torch.int32)
            if mv.numel() != 3 * N or cam.numel() != 2 * N or inv.numel() != N or pl.numel() != N:
                raise ValueError(f'flying action needs movement [{N},3], camera [{N},2], inventory [{N}], placement [{N}]')
            rc = self.lib.igw_step_flying(self.ctx, mv.data_ptr(), cam.data_ptr(), inv.data_ptr(), pl.data_ptr(),
                                          torch._C._cuda_getCurrentRawStream(self._dev_index))
            if rc:
                L.check(rc, 'igw_step_flying')
            self._act_keep = (mv, cam, inv, pl)
        return self._obs.copy(), self.reward, self.done, {}

    # ---- a captured step loop (the loop of examples/run_env.py:18-26 as ONE HIP-graph launch) ----
    def capture_steps(self, actions, record=False, chains=1):
        """Captures `for t in range(T): env.step(actions[t])` into a HIP graph and returns a StepGraph; replay() launches
        the T steps in one call (no per-step host work at all), bit-identical to the eager loop.  chains = P > 1 captures
        the T steps as P INDEPENDENT chains of launches, one per contiguous sub-batch of N / P envs (envs never read one
        another's state, so the results are the same bytes), each a linear graph replayed on its own stream: the tail
        of one sub-batch's step t -- the wait for its slowest wavefront -- and the ramp of its step t + 1 can overlap the
        other sub-batches' work instead of idling the chip between two whole-batch launches (all action spaces; not
        with the episode log or the RandomTasks generator, which index by context).  `actions`: walking int32
        device tensor [T, N]; flying dict of device tensors movement f32[T,N,3], camera f32[T,N,2], inventory i32[T,N],
        placement i32[T,N]; walking Dict: buttons u8[T,N,8], camera f32[T,N,2].  The graph reads the action BUFFERS at
        replay time: refill them in place (copy_) between replays to step with new actions.  record=True also copies
        every step's output record (StepGraph.outs uint8 [T, N, 64]; .rewards / .dones are views of it).  What varies
        between replays -- actions, env state, task table, the samplers' episode counters, the log's heads -- lives in
        device memory (include/igw.h), so samplers, auto-resets and the episode log advance inside the replayed graph
        exactly as they do eagerly.  What does NOT: the sampler SETTINGS and the episode-log buffers are kernel
        parameters, frozen at capture; after set_task_sampling / set_random_tasks / enable_ / disable_trajectory_log
        replay() raises (capture again)."""
        self._need_tasks()
        return StepGraph(self, actions, record, chains)

    def step_walking_ptr(self, actions_i32):
        """Hot-loop variant: `actions_i32` is already a contiguous int32 device tensor [N]."""
        if actions_i32.numel() != self.num_envs:
            raise ValueError(f'walking action needs {self.num_envs} entries, got {actions_i32.numel()}')
        L.check(self.lib.igw_step_walking(self.ctx, actions_i32.data_ptr(), self._stream()), 'igw_step_walking')

    def rollout(self, T, seed, t0=0, env_offset=0):
        """T fused walking steps per env with counter-RNG actions and auto-reset (one launch)."""
        self._need_tasks()
        L.check(self.lib.igw_rollout_walking(self.ctx, int(T), int(seed), int(t0), int(env_offset),
                                             self._stream()), 'igw_rollout_walking')

    def rollout_actions(self, actions, return_rewards=False):
        """Fused replay of a recorded action sequence: `actions` int32 [T, N] (Discrete(18) ids), or for the flying action
        space a dict of [T, N, ...] arrays (movement, camera, inventory, placement) -- bit-identical to
        T calls of step(), in one launch without a barrier between the steps of different envs (the context's
        autoreset setting applies).  With return_rewards: (rewards float32 [T, N], dones uint8 [T, N])."""
        self._need_tasks()
        N, dev = self.num_envs, self.device
        if self.walk_dict:
            raise L.IgwError('rollout_actions: Discrete(18) walking and flying only')
        if self._traj is not None:
            raise L.IgwError('rollout / rollout_actions do not write the episode log: disable_trajectory_log() first')
        if self.flying:   # dict(movement f32[T,N,3], camera f32[T,N,2], inventory i32[T,N], placement i32[T,N])
            self._check_camera(actions['camera'])
            mv = torch.as_tensor(actions['movement'], device=dev).to(torch.float32).contiguous()
            cam = torch.as_tensor(actions['camera'], device=dev).to(torch.float32).contiguous()
            inv = torch.as_tensor(actions['inventory'], device=dev).to(torch.int32).contiguous()
            plc = torch.as_tensor(actions['placement'], device=dev).to(torch.int32).contiguous()
            T = mv.shape[0]
            if tuple(mv.shape) != (T, N, 3) or tuple(cam.shape) != (T, N, 2) or tuple(inv.shape) != (T, N) or tuple(plc.shape) != (T, N):
                raise ValueError(f'flying actions must be [T,{N},3], [T,{N},2], [T,{N}], [T,{N}]')
            keep = (mv, cam, inv, plc)
        else:
            a = torch.as_tensor(actions, device=dev)
            if a.dim() != 2 or a.shape[1] != N:
                raise ValueError(f'actions must be [T, {N}], got {tuple(a.shape)}')
            a = a.to(torch.int32).contiguous()
            T = a.shape[0]
            keep = (a,)
        rw = dn = None
        if return_rewards:
            rw = torch.empty((T, N), dtype=torch.float32, device=dev)
            dn = torch.empty((T, N), dtype=torch.uint8, device=dev)
        rp, dp = (rw.data_ptr() if rw is not None else None), (dn.data_ptr() if dn is not None else None)
        if self.flying:
            L.check(self.lib.igw_rollout_flying_actions(self.ctx, mv.data_ptr(), cam.data_ptr(), inv.data_ptr(), plc.data_ptr(),
                                                        int(T), rp, dp, self._stream()), 'igw_rollout_flying_actions')
        else:
            L.check(self.lib.igw_rollout_walking_actions(self.ctx, a.data_ptr(), int(T), rp, dp, self._stream()),
                    'igw_rollout_walking_actions')
        self._keep = keep  # the launch reads them asynchronously
        return (rw, dn) if return_rewards else None

    def fill_actions(self, n_steps, seed, t0=0, env_offset=0):
        a = torch.empty((n_steps, self.num_envs), dtype=torch.int32, device=self.device)
        L.check(self.lib.igw_fill_actions_walking(self.ctx, a.data_ptr(), int(n_steps), int(t0), int(seed),
                                                  int(env_offset), self._stream()), 'igw_fill_actions_walking')
        return a

    # ---- asynchronous sub-batches ----
    def split(self, parts, streams=None):
        """`parts` contiguous sub-batches that SHARE this env's tensors (each is a view of rows
        [lo, hi)) but have their own context and HIP stream, so they can be stepped independently --
        e.g. policy inference on one half overlaps env stepping of the other (EnvPool-style async mode).
        Envs are independent, so results are identical to stepping the whole batch.  Every sub-batch stream
        first waits for the work already queued on the current stream (set_tasks / reset / steps of the parent);
        SubBatch.synchronize() / join() order the current stream after the sub-batch again."""
        if self.num_envs % parts:
            raise ValueError('num_envs must be divisible by parts')
        n = self.num_envs // parts
        streams = streams or [torch.cuda.Stream(device=self.device) for _ in range(parts)]
        cur = torch.cuda.current_stream(self.device)
        subs = []
        for k in range(parts):
            streams[k].wait_stream(cur)
            subs.append(SubBatch(self, k * n, n, streams[k]))
        self._children.extend(subs)
        return subs

    def _release_children(self, subs):
        """Sub-batches that are no longer needed: their counters move into this env's own buffer."""
        for c in subs:
            if c in self._children:
                self.stats_buf += c.stats_buf
                self._children.remove(c)

    # ---- introspection ----
    def stats_tensor(self):
        """The device counters [8] int64 of this env and its sub-batches as a DEVICE tensor (no synchronisation)."""
        s = self.stats_buf.sum(0)
        for c in self._children:
            s = s + c.stats_buf.sum(0)
        return s

    def stats(self):
        """Device counters of this env and of its sub-batches (VecGridWorld.split).  `steps`: env-steps executed, counted
        on the device by every step launch and fused rollout (`rollout_steps` is the same counter's old name)."""
        s = self.stats_buf.sum(0)
        for c in self._children:
            s = s + c.stats_buf.sum(0)
        s = s.cpu()
        return {'changed': int(s[L.STAT_CHANGED]), 'resets': int(s[L.STAT_RESETS]),
                'steps': int(s[L.STAT_STEPS]), 'rollout_steps': int(s[L.STAT_STEPS]), 'rescans': int(s[L.STAT_RESCANS]),
                'bad_poses': int(s[L.STAT_BAD_POSE]), 'bad_actions': int(s[L.STAT_BAD_ACTION]),
                'bad_tasks': int(s[L.STAT_BAD_TASK])}

    def internals(self):
        """float64 [N,8]: x, y, z, yaw, pitch, dy, time_int_steps, active_block (debug / parity)."""
        raw = self.agent_buf.cpu().numpy()
        out = np.zeros((self.num_envs, 8), np.float64)
        out[:, :6] = raw[:, :48].copy().view(np.float64)
        pack = raw[:, 62:64].copy().view(np.uint16)[:, 0].astype(np.int64)
        out[:, 6] = np.array([2, 4, 8, 12])[pack & 3]
        out[:, 7] = (pack >> 2) & 7
        return out

    def task_state(self):
        raw, aux = self.agent_buf.cpu().numpy(), self.aux_buf.cpu().numpy()
        i16 = aux[:, :8].copy().view(np.int16).astype(np.int64)
        return {'step_no': raw[:, 60:62].copy().view(np.uint16)[:, 0].astype(np.int64),
                'inventory': raw[:, 48:60].copy().view(np.int16).astype(np.int64),
                'size': i16[:, 0], 'prev_size': i16[:, 1] & 0x7fff, 'dirty': (i16[:, 1] >> 15) & 1,
                'max_int': i16[:, 2], 'target_size': i16[:, 3]}

    def set_step_no(self, step_no):
        """Overwrites GridWorld.step_no of every env (int tensor / array [N]): de-synchronises the episodes of a batch."""
        sn = torch.as_tensor(step_no).to(device=self.agent_buf.device, dtype=torch.int16).reshape(self.num_envs)
        self.agent_buf.view(torch.int16)[:, 30] = sn


class StepGraph:
    """T captured env steps (VecGridWorld.capture_steps).  replay() launches them on the current stream and returns the
    env's (obs, reward, done, info) views, which then hold the LAST step's values."""

    def __init__(self, env, actions, record, chains=1):
        self.env = env
        dev, N = env.device, env.num_envs
        chains = int(chains)
        if chains < 1 or N % chains:
            raise ValueError(f'capture_steps: chains must divide num_envs ({N})')
        if chains > 1 and (env._traj is not None or (env._sampling or ('',))[0] == 'random'):
            raise L.IgwError('capture_steps(chains > 1) cannot be combined with the episode log or the RandomTasks generator')
        self.chains = chains
        # A launch is given the kernel parameters BY VALUE: the graph freezes the sampler settings (set_task_sampling,
        # set_random_tasks) and the episode-log buffers (enable / disable_trajectory_log) of the moment of capture.
        # replay() refuses to run after any of them changed (config_epoch), and the graph keeps the log's buffers
        # alive, so a stale graph can neither run a stale configuration silently nor write into freed memory.
        self.config_epoch = env.config_epoch
        self._held = env._traj

        def need(x, dt, shape, what):
            if not (type(x) is torch.Tensor and x.is_cuda and x.dtype is dt and x.is_contiguous() and tuple(x.shape[1:]) == shape):
                raise ValueError(f'capture_steps: {what} must be a contiguous {dt} device tensor [T, {", ".join(map(str, shape))}]')
            return x
        if env.flying:
            self.buffers = (need(actions['movement'], torch.float32, (N, 3), 'movement'), need(actions['camera'], torch.float32, (N, 2), 'camera'),
                            need(actions['inventory'], torch.int32, (N,), 'inventory'), need(actions['placement'], torch.int32, (N,), 'placement'))
            fn, row_bytes = env.lib.igw_step_flying, (12, 8, 4, 4)
        elif env.walk_dict:
            self.buffers = (need(actions['buttons'], torch.uint8, (N, 8), 'buttons'), need(actions['camera'], torch.float32, (N, 2), 'camera'))
            fn, row_bytes = env.lib.igw_step_walking_dict, (8, 8)
        else:
            self.buffers = (need(actions, torch.int32, (N,), 'actions'),)
            fn, row_bytes = env.lib.igw_step_walking, (4,)
        self.T = T = int(self.buffers[0].shape[0])
        if T < 1 or any(b.shape[0] != T for b in self.buffers):
            raise ValueError('capture_steps: every action buffer needs the same number of steps T >= 1')
        self.outs = torch.zeros((T, N, L.OUT_BYTES), dtype=torch.uint8, device=dev) if record else None
        ptrs = [tuple(b[t].data_ptr() for b in self.buffers) for t in range(T)]
        self.graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        # (thread-local: other threads of the process -- an RCCL watchdog, a data loader -- may touch the runtime)
        # chains > 1: one context, one stream and ONE LINEAR GRAPH per chain, replayed side by side on their streams.
        # (Measured, profiles/r05_chains.txt: captured as parallel BRANCHES of one graph the chains do not overlap --
        # the runtime executes the branches one after the other, 2 / 4 / 8 branches cost 12.3 / 16.6 / 25.6 us per
        # step against 10.9 for the single chain -- while kernels of different STREAMS do run concurrently.)
        self.subs = env.split(chains) if chains > 1 else None   # (kept alive with the graphs)
        self.graphs = [self.graph] + [torch.cuda.CUDAGraph() for _ in range(chains - 1)]
        self.streams = [cap] + [torch.cuda.Stream(device=dev) for _ in range(chains - 1)]
        for k in range(chains):
            st = self.streams[k]
            st.wait_stream(torch.cuda.current_stream(dev))
            ctx = env.ctx if chains == 1 else self.subs[k].ctx
            lo, n = (0, N) if chains == 1 else (self.subs[k].lo, self.subs[k].num_envs)
            offs = tuple(b * lo for b in row_bytes)
            with torch.cuda.graph(self.graphs[k], stream=st, capture_error_mode='thread_local'):
                h = C.c_void_p(st.cuda_stream)
                for t in range(T):
                    L.check(fn(ctx, *(p + o for p, o in zip(ptrs[t], offs)), h), 'step (capture)')
                    if record:
                        self.outs[t, lo:lo + n].copy_(env.out_buf[lo:lo + n])
            torch.cuda.current_stream(dev).wait_stream(st)
        if record:
            f = self.outs.view(torch.float32)
            self.rewards, self.dones = f[:, :, 12], self.outs[:, :, 52]

    def __del__(self):
        # the chains' contexts were registered with the env for its counters: fold their counts into the env's own
        # buffer and let them go (a bench that captures many graphs must not accumulate contexts)
        subs, env = getattr(self, 'subs', None), getattr(self, 'env', None)
        if subs and env is not None:
            try:
                env._release_children(subs)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass

    def replay(self):
        env = self.env
        if env.config_epoch != self.config_epoch:
            raise L.IgwError('StepGraph is stale: set_task_sampling / set_random_tasks / enable_trajectory_log / '
                             'disable_trajectory_log was called after capture_steps (a captured launch carries those '
                             'settings by value); capture the steps again')
        if self.chains == 1:
            self.graph.replay()
        else:   # fork: every chain's stream waits for the caller's stream, runs its graph; the caller's stream joins
            cur = torch.cuda.current_stream(env.device)
            for g, st in zip(self.graphs, self.streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    g.replay()
            for st in self.streams:
                cur.wait_stream(st)
        return env._obs.copy(), env.reward, env.done, {}


class SubBatch:
    """Rows [lo, lo + n) of a VecGridWorld behind their own igw context and stream (VecGridWorld.split).
    The context carries the parent's global env offset, so device-side task generators draw the streams the
    whole batch would draw, and inherits the parent's generator settings."""

    def __init__(self, parent, lo, n, stream):
        self.parent, self.lo, self.num_envs, self.stream = parent, lo, n, stream
        self.lib, self.device = parent.lib, parent.device
        cfg = L.Config.from_buffer_copy(parent.cfg)
        cfg.num_envs = n
        cfg.env_index_base = parent.env_index_base + lo
        if cfg.lanes_per_env == 0:   # the group width the library chose for the WHOLE batch (include/igw.h: IGW_AUTO_*)
            cfg.lanes_per_env = L.auto_lanes(parent.num_envs)
        self.cfg = cfg
        self.ctx = C.c_void_p()
        L.check(self.lib.igw_create(C.byref(cfg), C.byref(self.ctx)), 'igw_create')
        sl = slice(lo, lo + n)
        self.stats_buf = torch.zeros_like(parent.stats_buf)
        rows = lambda t: t[sl].data_ptr()  # noqa: E731
        ptrs = [rows(parent.grid_buf), rows(parent.occ_buf), rows(parent.hist_buf), rows(parent.agent_buf), rows(parent.aux_buf),
                parent.task_target.data_ptr(), parent.task_start.data_ptr(), parent.task_start_occ.data_ptr(),
                parent.task_meta.data_ptr(), parent.task_index.data_ptr(), rows(parent.out_buf), self.stats_buf.data_ptr()]
        L.check(self.lib.igw_bind_buffers(self.ctx, C.byref(L.Buffers(*ptrs))), 'igw_bind_buffers')
        self.agent_pos, self.inventory = parent.agent_pos[sl], parent.inventory[sl]
        self.compass, self.reward, self.done = parent.compass[sl], parent.reward[sl], parent.done[sl]
        self.grid = parent.grid[sl]
        self._inherit_sampling()

    def _inherit_sampling(self):
        sp = self.parent._sampling
        if sp is None:
            L.check(self.lib.igw_set_task_sampling(self.ctx, 0, 0, 0), 'igw_set_task_sampling')
            L.check(self.lib.igw_set_random_tasks(self.ctx, 0, 0, 1, 1, 1, 1, None), 'igw_set_random_tasks')
        elif sp[0] == 'table':
            L.check(self.lib.igw_set_task_sampling(self.ctx, 1, sp[1], sp[2]), 'igw_set_task_sampling')
        else:  # generated rows are indexed by the context's local env, a sub-batch would overwrite rows of another
            raise L.IgwError('the RandomTasks generator cannot be combined with sub-batches')

    def __del__(self):
        if getattr(self, 'ctx', None):
            self.lib.igw_destroy(self.ctx)
            self.ctx = None

    def obs(self):
        return {'agentPos': self.agent_pos, 'inventory': self.inventory, 'compass': self.compass.unsqueeze(1),
                'grid': self.grid}

    def step_walking_ptr(self, actions_i32):
        """actions_i32: contiguous int32 device tensor [n]; launched on this sub-batch's stream (the tensor is
        marked as in use there, so the caching allocator does not recycle it while the kernel reads it)."""
        if actions_i32.numel() != self.num_envs:
            raise ValueError(f'walking action needs {self.num_envs} entries, got {actions_i32.numel()}')
        actions_i32.record_stream(self.stream)
        L.check(self.lib.igw_step_walking(self.ctx, actions_i32.data_ptr(), C.c_void_p(self.stream.cuda_stream)),
                'igw_step_walking')

    def reset(self):
        L.check(self.lib.igw_reset(self.ctx, None, 0, C.c_void_p(self.stream.cuda_stream)), 'igw_reset')
        return self.obs()

    def synchronize(self):
        self.stream.synchronize()

    def join(self):
        """Orders the current stream after everything queued on this sub-batch (no host wait)."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)


def task_eval(targets, grids, full_grids=None, invariant=None, device='cuda:0'):
    """Task(target, full_grid, invariant).maximal_intersection / argmax_intersection on `grid`
    for n pairs (tasks/task.py:121-161), on the GPU.  Returns numpy (max_int, argmax[n,3], target_size)."""
    if not torch.cuda.is_available():
        raise L.IgwError('task_eval needs a HIP device (no CPU fallback)')
    lib = L.load()
    dev = torch.device(device)
    t, g, f = _as_rows(targets, dev), _as_rows(grids, dev), _as_rows(full_grids, dev)
    n = t.shape[0]
    inv = None
    if invariant is not None:
        inv = torch.as_tensor(np.broadcast_to(np.asarray(invariant, dtype=np.uint8), (n,)).copy(), device=dev)
    mi = torch.zeros(n, dtype=torch.int32, device=dev)
    am = torch.zeros((n, 3), dtype=torch.int32, device=dev)
    ts = torch.zeros(n, dtype=torch.int32, device=dev)
    ptr = lambda x: None if x is None else C.c_void_p(x.data_ptr())  # noqa: E731
    L.check(lib.igw_task_eval(dev.index or 0, n, ptr(t), ptr(g), ptr(f), ptr(inv), ptr(mi), ptr(am), ptr(ts),
                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'igw_task_eval')
    return mi.cpu().numpy(), am.cpu().numpy(), ts.cpu().numpy()
