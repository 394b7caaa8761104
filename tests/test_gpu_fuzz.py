"""The randomised parity fuzz (tests/fuzz_parity.py) under the driver: 3 seeds x 100 cases -- random ragged batch sizes,
lanes per env, all three action spaces (Discrete(18), walking Dict, flying), reward modes, max_steps, start grids (also
whole floors of one colour: inventories far below zero), full grids, scales, initial poses and action mixes; eager
steps, chunks through the fused replay, the EXTRA kernel variant (episode log decoded and compared, on-device task
sampler, on-device RandomTasks generator), split(2) sub-batches on two streams, mid-run state_dict round trips and
host-side masked resets -- HIP path vs the CPU oracle after every step (continuous spaces: oracle in device-trig mode)."""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_fuzz_parity_100_cases(seed, monkeypatch):
    spec = importlib.util.spec_from_file_location('fuzz_parity', os.path.join(ROOT, 'tests', 'fuzz_parity.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, 'argv', ['fuzz_parity.py', '100', str(seed)])
    tally = mod.main()
    # every dimension the fuzz advertises was actually drawn
    for key in ('mode:walking', 'mode:flying', 'mode:walking_dict', 'extra:log', 'extra:sampling', 'extra:log+sampling',
                'extra:random_tasks', 'split', 'snapshot', 'fused', 'host_resets'):
        assert tally.get(key, 0) > 0, key
    assert tally['logged_episodes'] > 0
