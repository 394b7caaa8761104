"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/igw.h declares, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'igw.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(igw_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from gridworld_amd import _lib
    L = _lib.load()
    declared = _declared_symbols()
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(L, name), f'{name} declared in include/igw.h but not exported'
    assert sorted(_lib.EXPORTS) == declared
    assert L.igw_version() == _lib.VERSION == 5
    # the library is stamped with the hash of the sources it was built from (bench.py compares profiles with it)
    from gridworld_amd import build as B
    assert _lib.build_id() == B.source_hash() == B.built_id()


def test_layout_constants_match_header():
    from gridworld_amd import _lib
    src = open(os.path.join(ROOT, 'include', 'igw.h')).read()
    for name, val in (('IGW_GRID_STRIDE', _lib.GRID_STRIDE), ('IGW_CELLS', _lib.CELLS),
                      ('IGW_AGENT_BYTES', _lib.AGENT_BYTES), ('IGW_AUX_BYTES', _lib.AUX_BYTES), ('IGW_OUT_BYTES', _lib.OUT_BYTES),
                      ('IGW_TASK_META_BYTES', _lib.TASK_META_BYTES),
                      ('IGW_STAT_STRIPES', _lib.STAT_STRIPES), ('IGW_OCC_WORDS', _lib.OCC_WORDS),
                      ('IGW_TRAJ_BYTES', _lib.TRAJ_BYTES), ('IGW_VERSION', _lib.VERSION),
                      ('IGW_STAT_BAD_POSE', _lib.STAT_BAD_POSE), ('IGW_STAT_BAD_ACTION', _lib.STAT_BAD_ACTION),
                      ('IGW_STAT_BAD_TASK', _lib.STAT_BAD_TASK), ('IGW_LEVEL_INDEX_BYTES', _lib.LEVEL_INDEX_BYTES),
                      ('IGW_AUTO_32_MAX', _lib.AUTO_32_MAX), ('IGW_AUTO_16_MAX', _lib.AUTO_16_MAX), ('IGW_AUTO_8_MAX', _lib.AUTO_8_MAX),
                      ('IGW_STAT_STEPS', _lib.STAT_STEPS)):
        m = re.search(r'#define\s+%s\s+(\d+)' % name, src)
        assert m and int(m.group(1)) == val, name
    assert ctypes.sizeof(_lib.Config) == 64
    assert ctypes.sizeof(_lib.Buffers) == 12 * ctypes.sizeof(ctypes.c_void_p)
    assert _lib.TASK_INDEX_BYTES == 9 * _lib.LEVEL_INDEX_BYTES
    assert [_lib.auto_lanes(n) for n in (1, 1024, 1025, 4096, 4097, 24576, 24577, 65536)] == [32, 32, 16, 16, 8, 8, 4, 4]


def test_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from gridworld_amd import _lib, VecGridWorld, task_eval
    L = _lib.load()
    assert L.igw_device_count() == 0
    cfg = _lib.Config(0, 4, 4, 0, 1, 1, 250, 0, 1.0, 0.1, 0, 0)
    ctx = ctypes.c_void_p()
    assert L.igw_create(ctypes.byref(cfg), ctypes.byref(ctx)) == -2
    assert b'no CPU fallback' in L.igw_last_error()
    with pytest.raises(_lib.IgwError):
        VecGridWorld(4)
    with pytest.raises(_lib.IgwError):
        task_eval([[0] * 1089], [[0] * 1089])


def test_config_validation_messages():
    from gridworld_amd import _lib
    L = _lib.load()
    ctx = ctypes.c_void_p()
    bad = _lib.Config(0, 0, 1, 0, 1, 1, 250, 0, 1.0, 0.1, 0, 0)
    assert L.igw_create(ctypes.byref(bad), ctypes.byref(ctx)) == -1
    bad = _lib.Config(0, 4, 1, 0, 1, 1, 70000, 0, 1.0, 0.1, 0, 0)
    assert L.igw_create(ctypes.byref(bad), ctypes.byref(ctx)) == -1
    bad = _lib.Config(0, 4, 1, 7, 1, 1, 250, 0, 1.0, 0.1, 0, 0)
    assert L.igw_create(ctypes.byref(bad), ctypes.byref(ctx)) == -1
    assert L.igw_step_walking(None, None, None) == -1
