"""A short run of the randomised parity fuzz (tests/fuzz_parity.py): random ragged batch sizes, lanes per env,
reward modes, max_steps, start grids, scales and action mixes -- HIP path vs the CPU oracle at every step,
walking and flying (oracle in device-trig mode).  The tool runs hundreds of cases in minutes; this is the smoke."""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('seed', [11, 12])
def test_fuzz_parity_smoke(seed, monkeypatch):
    spec = importlib.util.spec_from_file_location('fuzz_parity', os.path.join(ROOT, 'tests', 'fuzz_parity.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, 'argv', ['fuzz_parity.py', '10', str(seed)])
    mod.main()
