"""ctypes binding of the C ABI in include/igw.h (libigw_hip.so, built by gridworld_amd/build.py).

There is NO CPU fallback: if the library cannot be loaded or no HIP device exists, every
entry point raises.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libigw_hip.so')

GRID_STRIDE = 1104
CELLS = 1089
AGENT_BYTES = 64
AUX_BYTES = 16
OUT_BYTES = 64
TASK_META_BYTES = 128
OCC_WORDS = 48
TRAJ_BYTES = 64
HIST_ROW = 512
STAT_STRIPES = 64
STAT_CHANGED, STAT_RESETS, STAT_STEPS, STAT_RESCANS, STAT_BAD_POSE, STAT_BAD_ACTION, STAT_BAD_TASK = 0, 1, 2, 3, 4, 5, 6
VERSION = 5
LEVEL_INDEX_BYTES = 160
TASK_INDEX_BYTES = 9 * LEVEL_INDEX_BYTES
WALKING_DISCRETE, FLYING, WALKING_DICT = 0, 1, 2
RESET_KEEP_SIZE = 1
CAMERA_MAX = 1e6   # IGW_CAMERA_MAX
AUTO_32_MAX, AUTO_16_MAX, AUTO_8_MAX = 1024, 4096, 24576   # IGW_AUTO_*_MAX (checked against the header by tests/test_abi.py)


def auto_lanes(num_envs):
    """The group width igw_create chooses for lanes_per_env = 0 (include/igw.h: IGW_AUTO_*_MAX) -- the ONE host-side
    copy of that rule (VecGridWorld.split, bench.py)."""
    return 32 if num_envs <= AUTO_32_MAX else 16 if num_envs <= AUTO_16_MAX else 8 if num_envs <= AUTO_8_MAX else 4

# every symbol include/igw.h declares (checked by tests/test_abi.py)
EXPORTS = ['igw_version', 'igw_build_id', 'igw_last_error', 'igw_device_count', 'igw_create', 'igw_destroy', 'igw_debug_set_stamps',
           'igw_bind_buffers', 'igw_prepare_tasks', 'igw_set_task_sampling', 'igw_set_random_tasks',
           'igw_set_trajectory_log', 'igw_reset', 'igw_step_walking', 'igw_step_flying', 'igw_step_walking_dict',
           'igw_rollout_walking', 'igw_rollout_walking_actions', 'igw_rollout_flying_actions', 'igw_fill_actions_walking', 'igw_task_eval', 'igw_debug_trig']


class IgwError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [('device', C.c_int32), ('num_envs', C.c_int32), ('num_tasks', C.c_int32),
                ('action_space', C.c_int32), ('select_and_place', C.c_int32), ('size_reward', C.c_int32),
                ('max_steps', C.c_int32), ('autoreset', C.c_int32),
                ('right_placement_scale', C.c_double), ('wrong_placement_scale', C.c_double),
                ('lanes_per_env', C.c_int32), ('reserved', C.c_int32), ('env_index_base', C.c_int64)]


class Buffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('grid', 'occ', 'hist', 'agent', 'aux', 'task_target', 'task_start',
                                          'task_start_occ', 'task_meta', 'task_index', 'out', 'stats')]


_lib = None


def load(build_if_missing=True):
    """Loads libigw_hip.so (building it with hipcc first if it is missing or stale).  IGW_DIAG=1 in the
    environment selects the diagnostic build (phase stamps, ablation switches) -- tools/ only."""
    global _lib
    if _lib is not None:
        return _lib
    diag = os.environ.get('IGW_DIAG') == '1'
    path = os.path.join(HERE, 'libigw_hip_diag.so') if diag else LIB_PATH
    if os.environ.get('IGW_LIB'):  # an explicitly built variant (tools/ab_variants.sh): A/B runs on one GPU box
        path, build_if_missing = os.environ['IGW_LIB'], False
    if build_if_missing:
        from . import build as _build
        try:
            _build.build(diag=diag)
        except subprocess.CalledProcessError as e:  # a compile error must never fall back to a stale binary
            raise IgwError(f'hipcc failed to build {os.path.basename(path)}: {e}') from e
        except Exception as e:  # no hipcc: an existing library is still usable, a missing one is fatal
            if not os.path.exists(path):
                raise IgwError(f'{os.path.basename(path)} is missing and could not be built: {e}') from e
    if not os.path.exists(path):
        raise IgwError(f'{os.path.basename(path)} not found; run `python -m gridworld_amd.build`')
    L = C.CDLL(path)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    L.igw_version.restype = C.c_int
    L.igw_last_error.restype = C.c_char_p
    L.igw_build_id.restype = C.c_char_p
    L.igw_device_count.restype = C.c_int
    L.igw_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.igw_destroy.argtypes = [vp]
    L.igw_debug_set_stamps.argtypes = [vp, vp]
    L.igw_bind_buffers.argtypes = [vp, C.POINTER(Buffers)]
    L.igw_prepare_tasks.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.igw_reset.argtypes = [vp, vp, i32, vp]
    L.igw_set_task_sampling.argtypes = [vp, i32, u64, i32]
    L.igw_set_random_tasks.argtypes = [vp, i32, u64, i32, i32, i32, i32, vp]
    L.igw_set_trajectory_log.argtypes = [vp, vp, vp, i32, i32]
    L.igw_step_walking.argtypes = [vp, vp, vp]
    L.igw_step_flying.argtypes = [vp, vp, vp, vp, vp, vp]
    L.igw_step_walking_dict.argtypes = [vp, vp, vp, vp]
    L.igw_rollout_walking.argtypes = [vp, i64, u64, i64, i64, vp]
    L.igw_rollout_walking_actions.argtypes = [vp, vp, i64, vp, vp, vp]
    L.igw_rollout_flying_actions.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp, vp]
    L.igw_fill_actions_walking.argtypes = [vp, vp, i64, i64, u64, i64, vp]
    L.igw_task_eval.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.igw_debug_trig.argtypes = [i32, i64, vp, vp, vp, vp, vp, vp, vp]
    for name in EXPORTS:
        getattr(L, name)
        if name not in ('igw_last_error', 'igw_build_id'):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def build_id():
    """igw_build_id() of the loaded library (hash of the kernel sources it was compiled from)."""
    return load().igw_build_id().decode()


def check(code, what=''):
    if code != 0:
        msg = load().igw_last_error()
        raise IgwError(f'{what} failed ({code}): {msg.decode() if msg else ""}')
