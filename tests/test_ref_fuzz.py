"""Reference <-> oracle randomised slice (build container only: needs /root/reference; skipped on the GPU box).

200 fresh scenarios of tests/golden/fuzz_ref_vs_oracle.py -- random create_env kwargs, task sources (Task,
Subtasks, CustomTasks, RandomTasks), starting grids, initialize_world poses, all three action spaces, glibc and
correctly rounded trig -- stepped in lock-step through the imported Python reference and the CPU oracle; every
observation, reward, done, grid, float64 internal and task counter must agree on every step.  The long run
(>= 5 M reference steps) is logged in profiles/r05_fuzz_ref_vs_oracle.json."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference'),
                                reason='the Python reference only exists in the build container')


def test_reference_vs_oracle_200_scenarios():
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    import fuzz_ref_vs_oracle as F
    s = F.run(range(77000, 77200), procs=min(4, os.cpu_count() or 1), max_T=150)   # (scenario generator: tests/scenario_fuzz.py)
    assert s['scenarios'] == 200 and s['reference_env_steps'] > 15000
    assert s['mismatches'] == 0, s['first_mismatches'][:2]
    # every scenario family was drawn
    for key in ('space:walking', 'space:walking_dict', 'space:flying', 'source:task', 'source:subtasks',
                'source:custom', 'source:random', 'trig:glibc', 'trig:cr_libm'):
        assert s['by'][key]['steps'] > 0, key
