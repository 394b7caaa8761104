"""MI355X-native vectorised IGLU gridworld: the env.step()/reset() hot path as HIP kernels.

    from gridworld_amd import VecGridWorld          # N envs, tensor obs (the fast path)
    import gridworld_amd; env = gridworld_amd.make('IGLUGridworld-v0', vector_state=True, render=False)
"""
from ._lib import IgwError  # noqa: F401
from .vec_env import SubBatch, VecGridWorld, task_eval  # noqa: F401

__version__ = '0.1.0'
from .env import GridWorld, SizeReward, Wrapper, create_env, make, make_vec, register  # noqa: F401,E402
from .tasks import Task, Tasks, CustomTasks, RandomTasks, Subtasks, dummy_task  # noqa: F401,E402
from . import workloads  # noqa: F401,E402
