#!/usr/bin/env python3
"""Randomised lock-step comparison of the Python reference with the CPU oracle (BUILD CONTAINER ONLY).

The fixtures under tests/golden/ pin the oracle (oracle/igw_oracle.c) to ~70 k recorded reference steps on fixed
seeds.  This script widens that link: it imports the reference from /root/reference (through ref_harness, so it
never travels to the GPU box), draws a fresh scenario per seed and steps the reference env and an
`oracle.OracleEnv` side by side, comparing EVERY step

    float32 observations (agentPos, compass: bit patterns), inventory, grid, reward (the Python value), done,
    the float64 internals (position, rotation, dy, time_int_steps, active_block: bit patterns),
    the synthetic task's max_int / prev_grid_size, GridWorld.max_int, step_no

and the reset observation / internals at every reset.  A scenario draws

  * create_env kwargs: action_space (walking Discrete(18) | walking Dict | flying), size_reward, select_and_place,
    right / wrong placement scales, max_steps (gridworld/env.py:333-350);
  * the task source: a plain Task (rt20 / uniform / CDM / multi-level / empty / DUMMY targets; empty, subset,
    foreign-block and wide starting grids; full_grid; invariant), or a generator with seeded np.random --
    Subtasks, CustomTasks, RandomTasks (tasks/task.py:208-308, tasks/task_set.py:22-157) -- whose drawn task is
    read back from the env at every reset and handed to the oracle;
  * initialize_world poses and overwritten starting grids (env.py:177-193);
  * the trig the reference computes with: glibc (CPython's math) against the oracle's libm hooks, or correctly
    rounded mpmath trig (ref_harness.cr_libm) against the oracle running the product's own sincos / atan2
    (oracle.use_device_trig) -- the latter also tests that igw_trig.h is correctly rounded on fresh arguments;
  * the action mix: uniform or builder-biased; continuous values as float32 widened to double (what the product
    accepts) or as raw doubles.

    python tests/golden/fuzz_ref_vs_oracle.py --scenarios 20000 --procs 8 --out profiles/r05_fuzz_ref_vs_oracle.json

tests/test_ref_fuzz.py runs a 200-scenario slice of this when /root/reference exists.  The scenario generator itself
is tests/scenario_fuzz.py, shared with the product-side run of the SAME seeds on the GPU (tests/test_gpu_facade_fuzz.py:
gridworld_amd.make(...) against the oracle).
"""
import argparse
import contextlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, os.path.dirname(HERE), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import ref_harness as H  # noqa: E402
import scenario_fuzz as S  # noqa: E402

FIELDS = S.FIELDS


class ReferenceBackend:
    """gym.make('IGLUGridworld-v0', ...) of the imported Python reference (through ref_harness' shims)."""
    name = 'reference'
    float32_actions = False      # the reference takes Python floats: raw doubles are drawn for a fifth of the scenarios

    def __init__(self):
        self.gym, self.Task, _ = H.load_reference()
        from gridworld.tasks.task import Subtasks
        from gridworld.tasks.task_set import CustomTasks, RandomTasks
        self.Subtasks, self.CustomTasks, self.RandomTasks = Subtasks, CustomTasks, RandomTasks
        g = H.load_cdm_goals()
        self._goals = [g[k] for k in sorted(g, key=lambda n: int(n[1:]))]

    def make(self, kw):
        return self.gym.make('IGLUGridworld-v0', vector_state=True, render=False, **kw)

    def goals(self):
        return self._goals

    def trig(self, crlibm):
        """glibc (CPython's math), or correctly rounded mpmath trig patched into gridworld.core.world"""
        return H.cr_libm() if crlibm else contextlib.nullcontext()

    def oracle_device_trig(self, crlibm):
        return bool(crlibm)      # correctly rounded reference <-> the product's own sincos / atan2 on the oracle's side

    def internals(self, env):
        return H._internals(env)

    def syn(self, env):
        s = env.unwrapped._synthetic_task
        return int(s.max_int), int(s.prev_grid_size)


_backend = None


def backend():
    global _backend
    if _backend is None:
        _backend = ReferenceBackend()
    return _backend


def run_scenario(seed, max_T=None):
    return S.run_scenario(seed, backend(), max_T)


def _worker(args):
    seed, max_T = args
    return S.run_scenario_safe(seed, backend(), max_T)


def run(seeds, procs=1, max_T=None, progress=None):
    t0 = time.time()
    jobs = [(int(s), max_T) for s in seeds]
    results = []
    if procs <= 1:
        for j in jobs:
            results.append(_worker(j))
    else:
        import multiprocessing as mp
        with mp.get_context('fork').Pool(procs) as pool:
            for i, r in enumerate(pool.imap_unordered(_worker, jobs, chunksize=4)):
                results.append(r)
                if progress and (i + 1) % progress == 0:
                    s = sum(x['steps'] for x in results)
                    b = sum(x['mismatch'] is not None for x in results)
                    print(f'{i + 1}/{len(jobs)} scenarios, {s} steps, {b} mismatches, {time.time() - t0:.0f} s',
                          flush=True)
    return S.summarize(results, time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenarios', type=int, default=200)
    ap.add_argument('--seed0', type=int, default=500000)
    ap.add_argument('--procs', type=int, default=1)
    ap.add_argument('--max-T', type=int, default=None)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    s = run(range(a.seed0, a.seed0 + a.scenarios), a.procs, a.max_T, progress=500)
    s['seed0'] = a.seed0
    s['command'] = ' '.join(sys.argv)
    text = json.dumps(s, indent=1)
    if a.out:
        with open(a.out, 'w') as f:
            f.write(text + '\n')
    print(text if len(text) < 6000 else json.dumps({k: v for k, v in s.items() if k != 'first_mismatches'}, indent=1))
    return 1 if s['mismatches'] else 0


if __name__ == '__main__':
    sys.exit(main())
