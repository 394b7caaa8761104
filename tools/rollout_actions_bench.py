import sys, time, torch
sys.path.insert(0, '.')
from gridworld_amd import VecGridWorld, workloads
N, T = 65536, 250
env = VecGridWorld(N, size_reward=False, autoreset=True)
env.set_tasks(workloads.rt20(N, seed=0, device=env.device)); env.reset()
env.rollout(250, seed=1)
acts = env.fill_actions(T, seed=3)
torch.cuda.synchronize()
for name, fn in (('rng rollout', lambda: env.rollout(T, seed=5)), ('recorded actions', lambda: env.rollout_actions(acts)),
                 ('recorded actions + rewards out', lambda: env.rollout_actions(acts, return_rewards=True))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('%-32s %.3f G env-steps/s' % (name, N * T / el / 1e9))
