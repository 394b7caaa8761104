#!/usr/bin/env python3
"""The reference's examples/run_env.py loop (the shape the steps/sec metric is defined on), on the HIP path.

    python examples/run_env.py                 # 1-env gym-shaped facade (API parity; latency-bound)
    python examples/run_env.py --vec 65536     # N envs, tensor observations (the fast path)
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


def vec(n, steps):
    env = G.make_vec(n, size_reward=False, autoreset=True)
    env.set_tasks(G.workloads.rt20(n, seed=0, device=env.device))
    env.reset()
    actions = torch.randint(0, 18, (steps, n), dtype=torch.int32, device=env.device)
    torch.cuda.synchronize()
    t = perf_counter()
    for k in range(steps):
        obs, reward, done, info = env.step(actions[k])   # tensors in HBM; a policy would read obs here
    torch.cuda.synchronize()
    dt = perf_counter() - t
    print(f'{n} envs x {steps} steps: {n * steps / dt / 1e6:.1f} M env-steps per second')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--vec', type=int, default=0)
    ap.add_argument('--episodes', type=int, default=4)
    ap.add_argument('--steps', type=int, default=500)
    a = ap.parse_args()
    vec(a.vec, a.steps) if a.vec else single(a.episodes)
