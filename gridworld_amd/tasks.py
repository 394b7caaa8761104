"""Host-side task containers with the reference's constructor surface
(gridworld/tasks/task.py:8-45, 163-187; gridworld/tasks/task_set.py:23-57, 59-160).

They only carry data to the device: rotations, admissible translations and every
intersection are computed by the HIP kernels (igw_prepare_tasks / igw_task_eval)."""
import numpy as np

BUILD_ZONE_SIZE_X = 11
BUILD_ZONE_SIZE_Z = 11
BUILD_ZONE_SIZE = 9, 11, 11


class Tasks:
    @classmethod
    def to_dense(cls, blocks):
        """sparse [(x, y, z, id)] -> dense [y+1, x+5, z+5] (tasks/task.py:168-175); None == []."""
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
        """dense -> [(x, y, z, id)] in row-major order of (y, x, z) (tasks/task.py:178-187)."""
        if isinstance(blocks, np.ndarray):
            ys, xs, zs = blocks.nonzero()
            return [(int(x) - BUILD_ZONE_SIZE_X // 2, int(y) - 1, int(z) - BUILD_ZONE_SIZE_Z // 2,
                     int(blocks[y, x, z])) for y, x, z in zip(ys, xs, zs)]
        return blocks

    def reset(self):
        raise NotImplementedError


class Task(Tasks):
    def __init__(self, chat, target_grid, last_instruction=None, starting_grid=None, full_grid=None,
                 invariant=True):
        self.chat = chat
        self.target_grid = np.asarray(self.to_dense(target_grid))
        self.last_instruction = last_instruction
        self.starting_grid = starting_grid  # sparse list (or None == [])
        self.full_grid = None if full_grid is None else np.asarray(full_grid)
        self.invariant = bool(invariant)
        self.target_size = int((self.target_grid != 0).sum())

    def reset(self):
        return self

    def __len__(self):
        return 1

    def __iter__(self):
        yield self

    def __repr__(self):
        ins = self.last_instruction or ''
        return f'Task(instruction={ins if len(ins) < 20 else ins[:20] + "..."})'


class CustomTasks(Tasks):
    """User-defined goal structures, uniformly sampled on reset (task_set.py:23-57)."""

    def __init__(self, goals, task_kwargs=None):
        self.task_kwargs = task_kwargs or {}
        self.tasks = [Task(conv, self.to_dense(grid), **self.task_kwargs) for conv, grid in goals]
        self.reset()

    def __getattr__(self, name):  # task_set.py:43-44: attribute access falls through to the current task
        if name == 'current':
            raise AttributeError(name)
        return getattr(self.current, name)

    def __len__(self):
        return len(self.tasks)

    def reset(self):
        self.current = self.tasks[int(np.random.choice(len(self.tasks)))]
        return self.current


class RandomTasks(Tasks):
    """Randomly generated targets (task_set.py:59-157); same sampling procedure and np.random stream
    use as the reference, with a guard against its infinite rejection loop."""

    def __init__(self, max_blocks=4, height_levels=1, allow_float=False, max_dist=2, num_colors=1, max_cache=0):
        self.height_levels, self.max_blocks, self.allow_float = height_levels, max_blocks, allow_float
        self.max_dist, self.num_colors, self.max_cache = max_dist, num_colors, max_cache
        self.tasks = [self.sample_task() for _ in range(max_cache)]
        self.reset()

    def __getattr__(self, name):  # task_set.py:90-91
        if name == 'current':
            raise AttributeError(name)
        return getattr(self.current, name)

    def __len__(self):
        return self.max_cache

    def reset(self):
        if self.max_cache > 0:
            self.current = self.tasks[int(np.random.choice(len(self.tasks)))]
        else:
            self.current = self.sample_task()
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


class Subtasks(Tasks):
    """Staged task: a dialog and a sequence of structures; reset() samples one turn as the goal with the
    previous structure as the starting grid (gridworld/tasks/task.py:208-308).  As in the reference the
    created Task is invariant with the full structure as `full_grid` (the `invariant` argument is stored
    but not forwarded, task.py:278-283)."""

    def __init__(self, dialog, structure_seq, invariant=False, progressive=True):
        self.dialog = dialog
        self.invariant = invariant
        self.progressive = progressive
        self.structure_seq = structure_seq
        self.next = None
        self.full = False
        self.task_start = 0
        self.task_goal = 0
        self.full_structure = self.to_dense(self.structure_seq[-1])
        self.current = self.reset()

    def __getattr__(self, name):
        if name == 'current':
            raise AttributeError(name)
        return getattr(self.current, name)

    def reset(self):
        if self.next is None:
            if len(self.structure_seq) == 1:
                turn = -1
            else:
                turn = int(np.random.choice(len(self.structure_seq))) - 1
            turn_goal = turn + 1
        else:
            turn = self.next
            turn_goal = self.next + 1
        self.task_start = turn
        self.task_goal = turn_goal
        self.current = self.create_task(self.task_start, self.task_goal)
        return self.current

    def __len__(self):
        return len(self.structure_seq)

    def __iter__(self):
        for i in range(len(self)):
            yield self.create_task(i - 1, i)

    def _dialog_until(self, last_turn):
        """Utterances 0..last_turn joined by newlines; a multi-line turn is flattened first and an empty
        running text takes the next turn without a separator (task.py:266-270)."""
        text = ''
        for turn in self.dialog[:last_turn + 1]:
            line = '\n'.join(turn) if isinstance(turn, list) else turn
            text = line if not text else text + '\n' + line
        return text

    def create_task(self, turn_start, turn_goal):
        goal_idx = -1 if self.full else min(turn_goal, len(self.structure_seq) - 1)
        start_blocks = self.structure_seq[turn_start] if turn_start >= 0 else []
        instruction = self.dialog[goal_idx]
        task = Task(self._dialog_until(turn_goal),
                    target_grid=self.to_dense(self.structure_seq[goal_idx]),
                    starting_grid=self.to_sparse(np.asarray(self.to_dense(start_blocks))),
                    full_grid=self.full_structure,
                    last_instruction='\n'.join(instruction) if isinstance(instruction, list) else instruction)
        return task.reset()


def dummy_task():
    """DUMMY_TASK (task_set.py:160): one blue block at sparse (5, 7, 5); starting grid None == []."""
    return CustomTasks(goals=[('', [(5, 7, 5, 1)])], task_kwargs={'invariant': False})
