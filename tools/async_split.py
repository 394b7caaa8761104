"""Runs on the GPU box: throughput of 65,536 envs stepped as P independent sub-batches on P streams
(VecGridWorld.split, EnvPool-style async mode) versus one whole-batch launch per step.  With one launch per
step the chip holds exactly one set of co-resident waves and every launch ends in a tail where only the
youngest waves are left; sub-batches on separate streams let the next step of one sub-batch fill the tail
of another."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
from gridworld_amd import VecGridWorld, workloads  # noqa: E402

N, K = 65536, 500
dev = torch.device('cuda', 0)


def run(parts):
    env = VecGridWorld(N, device=dev, action_space='walking', size_reward=False, max_steps=250, autoreset=True)
    env.set_tasks(workloads.rt20(N, seed=0, device=dev))
    env.reset()
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    sn = torch.randint(0, 250, (N,), generator=g, device=dev, dtype=torch.int32)
    env.agent_buf[:, 48] = (sn & 0xff).to(torch.uint8)
    env.agent_buf[:, 49] = (sn >> 8).to(torch.uint8)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.3:
        env.rollout(250, seed=17 + n, t0=n)
        torch.cuda.synchronize()
        n += 250
    acts = env.fill_actions(K, seed=0)
    torch.cuda.synchronize()
    fn = env.lib.igw_step_walking
    if parts == 1:
        jobs = [(env.ctx, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), 0, N)]
        subs = []
    else:
        subs = env.split(parts)
        jobs = [(s.ctx, C.c_void_p(s.stream.cuda_stream), s.lo, s.num_envs) for s in subs]
    ptrs = [[acts[t, lo:lo + n_].data_ptr() for (_, _, lo, n_) in jobs] for t in range(K)]
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        a = time.perf_counter()
        for t in range(K):
            pt = ptrs[t]
            for j, (ctx, st, _, _) in enumerate(jobs):
                fn(ctx, pt[j], st)
        torch.cuda.synchronize()
        el = time.perf_counter() - a
        best = el if best is None else min(best, el)
    st = env.stats()
    print('parts %d: %.3f G env-steps/s  (%.2f us per whole-batch step; resets %d)' % (parts, N * K / best / 1e9, 1e6 * best / K, st['resets']))
    del subs, env


for p in (1, 2, 4, 8):
    run(p)
