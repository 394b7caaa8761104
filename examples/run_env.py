#!/usr/bin/env python3
"""The reference's examples/run_env.py loop (the shape the steps/sec metric is defined on), on the HIP path.

    python examples/run_env.py                 # 1-env gym-shaped facade (API parity; latency-bound)
    python examples/run_env.py --vec 65536     # N envs, tensor observations (the fast path)
    python examples/run_env.py --vec 65536 --random-tasks      # RandomTasks.sample_task on the device at every reset
    python examples/run_env.py --vec 4096 --log 2 episodes/    # dump the episodes of the first 2 envs (Logged-style npz)
    python examples/run_env.py --vec 65536 --replay            # the same loop over a recorded action sequence, fused into one launch
"""
import argparse
import os
import sys
from time import perf_counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import gridworld_amd as G  # noqa: E402


def single(episodes):
    env = G.make('IGLUGridworldVector-v0')      # vector_state=True, render=False
    env.set_task(G.dummy_task())                # DUMMY_TASK
    print(f'Action space: {env.action_space}')
    time, steps = 0.0, 0
    for _ in range(episodes):
        done = False
        env.reset()
        while not done:
            action = env.action_space.sample()
            t = perf_counter()
            obs, reward, done, info = env.step(action)
            time += perf_counter() - t
            steps += 1
    print(f'steps per second: {steps / time:.4f}')


def vec(n, steps, random_tasks=False, log=None, replay=False):
    env = G.make_vec(n, size_reward=False, autoreset=True)
    if random_tasks:   # RandomTasks(max_blocks=20, max_dist=2, num_colors=6), sampled on the device at every reset
        env.set_random_tasks(True, seed=0, max_blocks=20, height_levels=1, max_dist=2, num_colors=6)
    else:
        env.set_tasks(G.workloads.rt20(n, seed=0, device=env.device))
    logger = None
    if log:
        from gridworld_amd.wrappers import EpisodeLogger
        logger = EpisodeLogger(env, n_envs=int(log[0]), path=log[1], desc='example')
    env.reset()
    actions = torch.randint(0, 18, (steps, n), dtype=torch.int32, device=env.device)
    torch.cuda.synchronize()
    t = perf_counter()
    if replay:   # open-loop: all steps in one launch, every step's reward / done returned
        rewards, dones = env.rollout_actions(actions, return_rewards=True)
    else:
        for k in range(steps):
            obs, reward, done, info = env.step(actions[k])   # tensors in HBM; a policy would read obs here
    torch.cuda.synchronize()
    dt = perf_counter() - t
    print(f'{n} envs x {steps} steps: {n * steps / dt / 1e6:.1f} M env-steps per second; counters {env.stats()}')
    if logger:
        files = [ep['file'] for ep in logger.collect()]
        print(f'{len(files)} finished episodes of the logged envs written, e.g. {files[:2]}')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--vec', type=int, default=0)
    ap.add_argument('--episodes', type=int, default=4)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--random-tasks', action='store_true')
    ap.add_argument('--log', nargs=2, metavar=('N_ENVS', 'DIR'))
    ap.add_argument('--replay', action='store_true')
    a = ap.parse_args()
    vec(a.vec, a.steps, a.random_tasks, a.log, a.replay) if a.vec else single(a.episodes)
