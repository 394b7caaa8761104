"""The product's 1-env gym facade against the CPU oracle on the SAME random scenarios the Python reference was run on
(tests/scenario_fuzz.py; reference side: tests/golden/fuzz_ref_vs_oracle.py, 31.3 M reference env-steps logged under
profiles/).  A scenario draws create_env kwargs, a task source -- Task, or Subtasks / CustomTasks / RandomTasks with
seeded np.random, which gridworld_amd.tasks consumes exactly as the reference's generators do --, starting grids,
full_grid, initialize_world poses, one of the three action spaces and an action mix; the facade
(gridworld_amd.make('IGLUGridworld-v0', ...): SizeReward(GridWorld) on the HIP path) and an OracleEnv are stepped in
lock-step and every observation, reward (the Python value), done, float64 internal (env.unwrapped.agent), task
counter, GridWorld.max_int and step_no must agree on every step, every reset observation too.  Seeds 1,000,000+ are
the first scenarios of the logged reference run: what both sides agree with the oracle on, they agree on with each
other."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_facade_vs_oracle_on_the_reference_fuzz_scenarios():
    import scenario_fuzz as S
    s = S.run(range(1000000, 1000160), S.ProductBackend(), max_T=150)
    assert s['scenarios'] == 160 and s['env_steps'] > 12000 and s['resets'] > 300
    assert s['mismatches'] == 0, s['first_mismatches'][:2]
    for key in ('space:walking', 'space:walking_dict', 'space:flying', 'source:task', 'source:subtasks',
                'source:custom', 'source:random'):
        assert s['by'][key]['steps'] > 0, key
    print('\nfacade vs oracle:', {k: s[k] for k in ('scenarios', 'env_steps', 'resets', 'mismatches', 'wall_s')})


def test_facade_vs_glibc_oracle_on_the_continuous_scenarios():
    """The independent-trig leg of the facade fuzz: the same scenarios with the oracle computing with GLIBC trig (the
    row above runs it on the host compile of the product's own trig header -- for the two continuous action spaces that
    compares the product's 1-env path with its own arithmetic).  Contract: integers and float32 observations bit-exact on
    every step, float64 internals within last-bit differences (counted).  Match: gridworld/core/world.py:396-414, 163-201."""
    import scenario_fuzz as S
    s = S.run(range(1000000, 1000160), S.ProductBackendGlibc(), max_T=150)
    assert s['scenarios'] == 160 and s['env_steps'] > 12000
    assert s['mismatches'] == 0, s['first_mismatches'][:2]
    for key in ('space:walking_dict', 'space:flying'):
        assert s['by'][key]['steps'] > 2000, key
    assert s.get('max_abs_float64_deviation', 0.0) < 1e-9
    print('\nfacade vs glibc oracle:', {k: s.get(k) for k in ('scenarios', 'env_steps', 'mismatches', 'scenarios_with_a_float64_difference',
                                                              'steps_with_a_float64_difference', 'max_abs_float64_deviation', 'wall_s')})
