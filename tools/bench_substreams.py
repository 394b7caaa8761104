import sys, time, torch
sys.path.insert(0, '.')
from gridworld_amd import VecGridWorld, workloads
N, K, W = 65536, 300, 20
for parts in (1, 2, 4, 8):
    for gs in (4, 2, 8):
        n = N // parts
        envs, streams, acts = [], [], []
        for p in range(parts):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                e = VecGridWorld(n, size_reward=False, autoreset=True, lanes_per_env=gs)
                e.set_tasks(workloads.rt20(n, seed=p, device=e.device))
                e.reset()
                a = e.fill_actions(W + K, seed=5, env_offset=p * n)
            envs.append(e); streams.append(s); acts.append(a)
        torch.cuda.synchronize()
        def run(t0, t1):
            for t in range(t0, t1):
                for p in range(parts):
                    with torch.cuda.stream(streams[p]):
                        envs[p].step_walking_ptr(acts[p][t])
        run(0, W); torch.cuda.synchronize()
        t = time.perf_counter(); run(W, W + K); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f'parts {parts} GS {gs}: {N * K / dt / 1e6:.1f} M steps/s  ({dt / K * 1e6:.1f} us/step)')
        del envs, streams, acts
