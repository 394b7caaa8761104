"""Host-side task objects with the reference's surface (gridworld/tasks/task.py:8-161, 163-308;
gridworld/tasks/task_set.py:23-160).

They carry data and the task protocol; every intersection they report is computed on the GPU through the
C ABI (igw_task_eval -- the same full-vote kernel igw_prepare_tasks uses), never on the host:
`maximal_intersection`, `argmax_intersection`, `get_intersection`, `step_intersection` and `reset` raise
without a HIP device.  What stays on the host is data layout only: sparse <-> dense conversion, the four
rotated copies of the target (`target_grids`) and the admissible translation lists (`admissible`), which are
attributes of the reference's Task that user code reads.
"""
import pickle
import uuid

import numpy as np

BUILD_ZONE_SIZE_X = 11
BUILD_ZONE_SIZE_Z = 11
BUILD_ZONE_SIZE = 9, 11, 11

_EVAL_DEVICE = 'cuda:0'


def set_eval_device(device):
    """Device the host Task objects evaluate intersections on (default cuda:0)."""
    global _EVAL_DEVICE
    _EVAL_DEVICE = device


def _device_eval(targets, grids, full_grids=None, invariant=True):
    from .vec_env import task_eval  # imported lazily: tasks.py itself needs no GPU to build task data
    return task_eval(targets, grids, full_grids, invariant, device=_EVAL_DEVICE)


def _rot90(grid):
    """One reference rotation around the vertical axis: new[:, z, 10 - x] = old[:, x, z] (task.py:47-56)."""
    return np.ascontiguousarray(np.flip(np.swapaxes(grid, 1, 2), axis=2))


def _shifted(grid, dx, dz):
    """out[:, x, z] = grid[:, x + dx, z + dz] inside the zone, 0 elsewhere: the window
    grid[:, max(dx,0):11+min(dx,0), ...] of task.py:126-128 moved onto the grid window it is compared with."""
    out = np.zeros_like(grid)
    xs, xd = slice(max(dx, 0), BUILD_ZONE_SIZE_X + min(dx, 0)), slice(max(-dx, 0), BUILD_ZONE_SIZE_X + min(-dx, 0))
    zs, zd = slice(max(dz, 0), BUILD_ZONE_SIZE_Z + min(dz, 0)), slice(max(-dz, 0), BUILD_ZONE_SIZE_Z + min(-dz, 0))
    out[:, xd, zd] = grid[:, xs, zs]
    return out


class Tasks:
    """Task-set protocol (task.py:163-206)."""

    @classmethod
    def to_dense(cls, blocks):
        """sparse [(x, y, z, id)] -> dense [y+1, x+5, z+5] (task.py:168-175); a dense array passes through;
        None (the reference's F2 TypeError) is treated as the empty list."""
        if blocks is None:
            return np.zeros(BUILD_ZONE_SIZE, dtype=np.int32)
        if isinstance(blocks, (list, tuple)):
            grid = np.zeros(BUILD_ZONE_SIZE, dtype=np.int32)
            for x, y, z, block_id in blocks:
                grid[int(y) + 1, int(x) + BUILD_ZONE_SIZE_X // 2, int(z) + BUILD_ZONE_SIZE_Z // 2] = int(block_id)
            return grid
        return np.asarray(blocks)

    @classmethod
    def to_sparse(cls, blocks):
        """task.py:177-187, as the reference returns it: a sparse list passes through unchanged (what every
        reference caller hands over, Subtasks.create_task included); for a dense ARRAY the reference unpacks the
        nonzero() indices -- (y, x, z) order -- as (x, y, z), so the result is
        (y_idx - 5, x_idx - 1, z_idx - 5, id) in nonzero() order, which its own to_dense does not read back.
        Kept bit-for-bit (fixture s10_task_protocol: to_sparse_in / to_sparse_out); `dense_to_sparse` is the
        inverse of to_dense."""
        if isinstance(blocks, np.ndarray):
            idx = blocks.nonzero()
            return [(int(i0) - BUILD_ZONE_SIZE_X // 2, int(i1) - 1, int(i2) - BUILD_ZONE_SIZE_Z // 2,
                     blocks[i0, i1, i2]) for i0, i1, i2 in zip(*idx)]
        return blocks

    @classmethod
    def dense_to_sparse(cls, blocks):
        """dense [y+1, x+5, z+5] -> [(x, y, z, id)] in row-major order of (y, x, z): the inverse of to_dense
        (not part of the reference's surface; see to_sparse)."""
        if isinstance(blocks, np.ndarray):
            ys, xs, zs = blocks.nonzero()
            return [(int(x) - BUILD_ZONE_SIZE_X // 2, int(y) - 1, int(z) - BUILD_ZONE_SIZE_Z // 2,
                     int(blocks[y, x, z])) for y, x, z in zip(ys, xs, zs)]
        return blocks

    def reset(self):
        raise NotImplementedError

    def __len__(self):
        return NotImplemented

    def __iter__(self):
        return NotImplemented

    def set_task(self, task_id):
        return NotImplemented

    def get_target(self):
        return NotImplemented

    def set_task_obj(self, task):
        return NotImplemented


class Task(Tasks):
    """One goal structure (task.py:8-161).  Attributes follow the reference: `target_grid`, `target_grids`
    (4 rotations), `admissible` (per rotation the translations that keep the whole structure inside the zone;
    `[[(0, 0)]]` when not invariant), `target_size`, `full_size`, `max_int`, `prev_grid_size`,
    `right_placement`, `wrong_placement`."""

    def __init__(self, chat, target_grid, last_instruction=None, starting_grid=None, full_grid=None,
                 invariant=True):
        self.chat = chat
        self.starting_grid = starting_grid  # sparse list (or None)
        self.last_instruction = last_instruction
        self.target_grid = np.asarray(self.to_dense(target_grid))
        self.full_grid = None if full_grid is None else np.asarray(full_grid)
        self.invariant = bool(invariant)
        self.target_size = int((self.target_grid != 0).sum())
        self.full_size = self.target_size if self.full_grid is None else int((self.full_grid != 0).sum())
        self._max_int = 0
        self._max_int_of = None  # grid whose maximal intersection max_int stands for, until it is first read
        self.prev_grid_size = 0
        self.right_placement = 0
        self.wrong_placement = 0
        self._target_grids = None
        self._admissible = None

    @property
    def max_int(self):
        """reset() defines max_int as the intersection with the starting grid (task.py:79-82); the device
        evaluates it when the value is first read, so task data can be built on a host without a GPU."""
        if self._max_int_of is not None:
            grid, self._max_int_of = self._max_int_of, None
            self._max_int = self.maximal_intersection(grid)
        return self._max_int

    @max_int.setter
    def max_int(self, value):
        self._max_int_of = None
        self._max_int = value

    # -- derived data the reference builds eagerly in __init__ (task.py:40-72); built on first use here
    @property
    def target_grids(self):
        if self._target_grids is None:
            rots = [self.target_grid]
            for _ in range(3):
                rots.append(_rot90(rots[-1]).astype(np.int32))
            self._target_grids = rots
        return self._target_grids

    @property
    def admissible(self):
        """Translations (dx, dz) whose window keeps all `full_size` blocks (task.py:58-72).  The window
        [max(dx,0), 11+min(dx,0)) keeps every block iff dx <= xmin (dx >= 0) or dx >= xmax - 10 (dx <= 0):
        the bounding-box rule the kernels use; listed in the reference's (dx, dz) order."""
        if self._admissible is None:
            if not self.invariant:
                self._admissible = [[(0, 0)]]
            else:
                base = self.target_grid if self.full_grid is None else self.full_grid
                adm = []
                for _ in range(4):
                    _, xs, zs = np.nonzero(base)
                    if len(xs) == 0:
                        ok = [(dx, dz) for dx in range(-10, 11) for dz in range(-10, 11)]
                    else:
                        ok = [(dx, dz) for dx in range(int(xs.max()) - 10, int(xs.min()) + 1)
                              for dz in range(int(zs.max()) - 10, int(zs.min()) + 1)]
                    adm.append(ok)
                    base = _rot90(base)
                self._admissible = adm
        return self._admissible

    # -- protocol
    def reset(self):
        """task.py:74-86: max_int on the starting grid, prev_grid_size = number of starting blocks."""
        self.max_int = 0
        if self.starting_grid is not None and len(self.starting_grid) > 0:
            self._max_int_of = np.asarray(Tasks.to_dense(self.starting_grid))
        self.prev_grid_size = len(self.starting_grid) if self.starting_grid is not None else 0
        self.right_placement = 0
        self.wrong_placement = 0
        return self

    def __len__(self):
        return 1

    def __iter__(self):
        yield self

    def __repr__(self):
        ins = self.last_instruction or ''
        return f'Task(instruction={ins if len(ins) < 20 else ins[:20] + "..."})'

    # -- intersections: device
    def _eval(self, grid):
        fg = None if self.full_grid is None else self.full_grid[None]
        return _device_eval(self.target_grid[None], np.asarray(grid)[None], fg, self.invariant)

    def maximal_intersection(self, grid):
        """task.py:147-161."""
        return int(self._eval(grid)[0][0])

    def argmax_intersection(self, grid):
        """task.py:121-136: (dx, dz, rotation) of the first strict maximum; (0, 0, 0) when nothing matches."""
        return tuple(int(v) for v in self._eval(grid)[1][0])

    def get_intersection(self, grid, dx, dz, rot):
        """task.py:138-145: matches at one given translation and rotation.  The rotated target window is moved
        onto the grid window on the host (a copy), the count is the device's (0, 0) intersection of that pair."""
        moved = _shifted(self.target_grids[rot], int(dx), int(dz))
        return int(_device_eval(moved[None], np.asarray(grid)[None], None, False)[0][0])

    def step_intersection(self, grid):
        """task.py:103-119: (right_placement, wrong_placement, done); recomputes only when the block count
        changed, as the reference does."""
        grid = np.asarray(grid)
        grid_size = int((grid != 0).sum())
        wrong_placement = self.prev_grid_size - grid_size
        max_int = self.maximal_intersection(grid) if wrong_placement != 0 else self.max_int
        done = max_int == self.target_size
        self.prev_grid_size = grid_size
        right_placement = max_int - self.max_int
        self.max_int = max_int
        self.right_placement = right_placement
        self.wrong_placement = wrong_placement
        return right_placement, wrong_placement, done


class _Delegating(Tasks):
    """Task sets answer unknown attributes with those of their current task (task_set.py:43-44, 90-91;
    task.py:224-227)."""

    def __getattr__(self, name):
        if name == 'current':
            raise AttributeError(name)
        return getattr(self.current, name)


class CustomTasks(_Delegating):
    """User-defined goal structures, one drawn uniformly per reset (task_set.py:23-57)."""

    def __init__(self, goals, task_kwargs=None):
        self.task_kwargs = task_kwargs or {}
        self.tasks = {uuid.uuid4().hex: Task(conv, self.to_dense(grid), **self.task_kwargs) for conv, grid in goals}
        self.task_ids = list(self.tasks.keys())
        self.reset()

    def __len__(self):
        return len(self.task_ids)

    def __iter__(self):
        for task in self.tasks.values():
            yield from iter(task)

    def reset(self):
        # np.random.choice over the id list consumes the stream like choice(len(ids)) (task_set.py:54)
        self.current = self.tasks[self.task_ids[int(np.random.choice(len(self.task_ids)))]].reset()
        return self.current


class RandomTasks(_Delegating):
    """Randomly generated targets (task_set.py:59-157): same sampling procedure and np.random consumption as
    the reference, with a guard against its endless rejection loop.  The device-side
    equivalent (no host in the reset path) is VecGridWorld.set_random_tasks."""

    def __init__(self, max_blocks=4, height_levels=1, allow_float=False, max_dist=2, num_colors=1, max_cache=0):
        self.height_levels, self.max_blocks, self.allow_float = height_levels, max_blocks, allow_float
        self.max_dist, self.num_colors, self.max_cache = max_dist, num_colors, max_cache
        self.tasks = {}
        self.current = None
        for _ in range(self.max_cache):
            self.tasks[uuid.uuid4().hex] = self.sample_task()
        self.reset()

    def dump(self, path):
        """task_set.py:93-95: {uid: target_grid} pickle."""
        with open(path, 'wb') as f:
            pickle.dump({uid: t.target_grid for uid, t in self.tasks.items()}, f)

    def load(self, path):
        """task_set.py:97-100."""
        with open(path, 'rb') as f:
            grids = pickle.load(f)
        self.tasks = {uid: Task('', g) for uid, g in grids.items()}

    def __len__(self):
        return self.max_cache

    def __iter__(self):
        for task in self.tasks.values():
            yield task

    def __repr__(self):
        hps = dict(max_blocks=self.max_blocks, height_levels=self.height_levels, allow_float=self.allow_float,
                   max_dist=self.max_dist, num_colors=self.num_colors, max_cache=self.max_cache)
        return 'RandomTasks(' + ', '.join(f'{k}={v}' for k, v in hps.items()) + ')'

    def reset(self):
        if self.max_cache > 0:
            ids = list(self.tasks.keys())
            self.current_id = ids[int(np.random.choice(len(ids)))]
            self.current = self.tasks[self.current_id]
        else:
            self.current = self.sample_task()
        return self.current

    def set_task(self, task_id):
        self.current = self.tasks[task_id]
        return self.current

    def sample_task(self):
        target = np.zeros(BUILD_ZONE_SIZE, dtype=np.int32)
        for height in range(self.height_levels):
            bx = np.random.choice(BUILD_ZONE_SIZE_X)
            bz = np.random.choice(BUILD_ZONE_SIZE_Z)
            target[height, bx, bz] = np.random.choice(self.num_colors) + 1
            for _ in range(self.max_blocks - 1):
                free = [(dx, dz) for dx in range(-self.max_dist, self.max_dist + 1)
                        for dz in range(-self.max_dist, self.max_dist + 1)
                        if (dx or dz) and 0 <= bx + dx < BUILD_ZONE_SIZE_X and 0 <= bz + dz < BUILD_ZONE_SIZE_Z
                        and target[height, bx + dx, bz + dz] == 0]
                if not free:
                    break  # the reference would loop forever here
                dx, dz, colour = 0, 0, 1
                while (dx == 0 and dz == 0) or (dx, dz) not in free:
                    dx = np.random.choice(2 * self.max_dist + 1) - self.max_dist
                    dz = np.random.choice(2 * self.max_dist + 1) - self.max_dist
                    colour = np.random.choice(self.num_colors) + 1
                target[height, bx + dx, bz + dz] = colour
        return Task('', target)


class Subtasks(_Delegating):
    """A dialog with the structure after each turn (task.py:208-308).  reset() picks a turn: the structure
    before it is the starting grid, the structure after it the target, the final structure the `full_grid`
    that defines the admissible translations.  With `progressive`, step_intersection() moves on to the next
    turn's structure as soon as the current one is complete.  As in the reference the `invariant` argument
    is stored but the created Task is always invariant (task.py:278-283 does not forward it)."""

    def __init__(self, dialog, structure_seq, invariant=False, progressive=True):
        self.dialog, self.structure_seq = dialog, structure_seq
        self.invariant, self.progressive = invariant, progressive
        self.next, self.full = None, False
        self.task_start = self.task_goal = 0
        self.full_structure = self.to_dense(structure_seq[-1])
        self.current = self.reset()

    def __len__(self):
        return len(self.structure_seq)

    def __iter__(self):
        return (self.create_task(turn - 1, turn) for turn in range(len(self)))

    def __repr__(self):
        return (f'Subtasks(total_steps={len(self.structure_seq)}, current_task_start={self.task_start}, '
                f'current_task_end={self.task_goal})')

    def reset(self):
        """task.py:229-248: a forced turn (`next`) or a uniformly drawn one; one np.random.choice unless the
        sequence has a single structure."""
        if self.next is not None:
            before = self.next
        elif len(self.structure_seq) == 1:
            before = -1
        else:
            before = int(np.random.choice(len(self.structure_seq))) - 1
        self.task_start, self.task_goal = before, before + 1
        self.current = self.create_task(before, before + 1)
        return self.current

    def _utterance(self, turn):
        return '\n'.join(turn) if isinstance(turn, list) else turn

    def create_task(self, turn_start, turn_goal):
        """task.py:260-286."""
        last = len(self.structure_seq) - 1
        goal = -1 if self.full else min(turn_goal, last)
        history = [self._utterance(t) for t in self.dialog[:turn_goal + 1]]
        chat = history[0] if history else ''
        for line in history[1:]:
            chat = chat + '\n' + line if chat else line
        blocks_before = [] if turn_start == -1 else self.structure_seq[turn_start]
        task = Task(chat, target_grid=self.to_dense(self.structure_seq[goal]),
                    starting_grid=self.to_sparse(blocks_before),
                    full_grid=self.full_structure, last_instruction='\n'.join(self.dialog[goal]))
        return task.reset()  # max_int on the starting grid, prev_grid_size (task.py:284-285)

    def step_intersection(self, grid):
        """task.py:288-298.  GridWorld.step never calls this (it scores its own synthetic Task, env.py:291);
        it is the protocol for callers that score a Subtasks object themselves."""
        right, wrong, done = self.current.step_intersection(grid)
        if done and self.progressive and len(self.structure_seq) > self.task_goal:
            self.task_goal += 1
            self.current = self.create_task(self.task_start, self.task_goal)
            self.current.prev_grid_size = 0
            done = self.current.step_intersection(grid)[2]
        return right, wrong, done

    def set_task(self, task_id):
        """task.py:300-303 calls create_task(task_id) with one argument, a TypeError in the reference; here the
        turn `task_id` becomes the goal with the turn before it as the start."""
        self.task_id = task_id
        self.task_start, self.task_goal = task_id - 1, task_id
        self.current = self.create_task(task_id - 1, task_id)
        return self.current

    def set_task_obj(self, task):
        """task.py:305-308."""
        self.task_id = None
        self.current = task
        return self.current


def dummy_task():
    """DUMMY_TASK (task_set.py:160): one blue block at sparse (5, 7, 5); starting grid None == []."""
    return CustomTasks(goals=[('', [(5, 7, 5, 1)])], task_kwargs={'invariant': False})
