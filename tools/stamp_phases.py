"""Diagnostic: per-wave phase durations of the walking step kernel from in-kernel s_memtime stamps."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from gridworld_amd import VecGridWorld, workloads
N = 65536
for gs in (4,):
    env = VecGridWorld(N, size_reward=False, autoreset=True, lanes_per_env=gs)
    env.set_tasks(workloads.rt20(N, seed=0, device=env.device)); env.reset()
    acts = env.fill_actions(120, seed=1)
    for t in range(100): env.step_walking_ptr(acts[t])
    waves = N * gs // 64
    st = torch.zeros((waves, 8), dtype=torch.int64, device=env.device)
    env.lib.igw_debug_set_stamps(env.ctx, st.data_ptr())
    acc = []
    for t in range(100, 120):
        st.zero_(); env.step_walking_ptr(acts[t]); torch.cuda.synchronize()
        a = st.cpu().numpy().astype(np.float64)
        acc.append(a)
    a = np.stack(acc)  # [20, waves, 8]
    t0 = a[:, :, 0].min(axis=1, keepdims=True)
    names = ['start->loads', 'act+hit_test', 'physics', 'tail of world_step', 'changes', 'rescan/finish/stores']
    d = np.diff(a[:, :, :7], axis=2)
    print(f'GS {gs}: waves {waves}; clock ticks (100 MHz s_memtime? raw units)')
    print('  first wave start -> last wave end:', (a[:, :, 6].max(1) - a[:, :, 0].min(1)).mean())
    print('  wave start spread (last start - first start):', (a[:, :, 0].max(1) - a[:, :, 0].min(1)).mean())
    print('  wave lifetime mean / p50 / p99 / max:', (a[:, :, 6] - a[:, :, 0]).mean(), np.percentile(a[:, :, 6] - a[:, :, 0], 50), np.percentile(a[:, :, 6] - a[:, :, 0], 99), (a[:, :, 6] - a[:, :, 0]).max())
    for i, nme in enumerate(names):
        print(f'  {nme:24s} mean {d[:, :, i].mean():9.1f}  p99 {np.percentile(d[:, :, i], 99):9.1f}  max {d[:, :, i].max():9.1f}')
    env.lib.igw_debug_set_stamps(env.ctx, None)
