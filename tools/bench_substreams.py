"""Secondary measurement: the 65,536-env batch split into free-running sub-batches on their own HIP streams
(VecGridWorld.split).  Not the headline number: there is no per-step barrier across the whole batch."""
import sys, time, torch
sys.path.insert(0, '.')
from gridworld_amd import VecGridWorld, workloads
N, K, W = 65536, 400, 20
for parts in (1, 2, 4):
    env = VecGridWorld(N, size_reward=False, autoreset=True)
    env.set_tasks(workloads.rt20(N, seed=0, device=env.device))
    env.reset()
    acts = env.fill_actions(W + K, seed=5)
    subs = env.split(parts)
    m = N // parts
    chunks = [acts[:, k * m:(k + 1) * m].contiguous() for k in range(parts)]
    torch.cuda.synchronize()

    def run(t0, t1):
        for t in range(t0, t1):
            for k, sb in enumerate(subs):
                sb.step_walking_ptr(chunks[k][t])
    run(0, W)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(W, W + K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f'sub-batches {parts}: {N * K / dt / 1e6:.1f} M env-steps/s  ({dt / K * 1e6:.1f} us per step of all 65,536 envs)')
