#!/usr/bin/env python3
"""Diagnostic (IGW_DIAG build only): per-wave phase durations of the walking step kernel from in-kernel
s_memtime stamps, in the de-synchronised steady state bench.py times, with per-wave work features
(changed envs, rescans, resets, sub-steps) so the launch's tail can be attributed.

    IGW_DIAG=1 python tools/stamp_phases.py [out.npz] [lanes_per_env]
"""
import os
import sys

os.environ['IGW_DIAG'] = '1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gridworld_amd import VecGridWorld, workloads  # noqa: E402

N = 65536
out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/stamps.npz'
gs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 8: stamps do not drain the memory queues
FLY = os.environ.get('IGW_STAMP_MODE') == 'flying'   # BASELINE configs[3]
env = VecGridWorld(N, size_reward=False, autoreset=True, lanes_per_env=gs, debug_flags=flags,
                   action_space='flying' if FLY else 'walking')
if os.environ.get('IGW_STAMP_WORKLOAD') == 'cdm':   # the real IGLU targets with partial starting grids
    _tg, _st = workloads.cdm(N, 0, np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'cdm_goals.npz'))['dense'])
    env.set_tasks(_tg.to(env.device), _st.to(env.device))
else:
    env.set_tasks(workloads.rt20(N, seed=0, device=env.device))
env.reset()
g = torch.Generator(device=env.device)
g.manual_seed(1)
sn = torch.randint(0, 250, (N,), generator=g, device=env.device, dtype=torch.int32)
env.set_step_no(sn)
if FLY:
    import ctypes
    mvt = torch.empty((60, N, 3), device=env.device).uniform_(-1, 1, generator=g)
    cam = torch.empty((60, N, 2), device=env.device).uniform_(-5, 5, generator=g)
    inv = torch.randint(0, 7, (60, N), generator=g, device=env.device, dtype=torch.int32)
    plc = torch.randint(0, 3, (60, N), generator=g, device=env.device, dtype=torch.int32)

    def step(t):
        rc = env.lib.igw_step_flying(env.ctx, ctypes.c_void_p(mvt[t].data_ptr()), ctypes.c_void_p(cam[t].data_ptr()),
                                     ctypes.c_void_p(inv[t].data_ptr()), ctypes.c_void_p(plc[t].data_ptr()), env._stream())
        assert rc == 0
    for rep in range(5):   # pre-roll to the steady state: 300 steps
        for t in range(60):
            step(t)
else:
    env.rollout(250, seed=3)
    acts = env.fill_actions(60, seed=1)

    def step(t):
        env.step_walking_ptr(acts[t])
for t in range(20):
    step(t)
waves = N * gs // 64
st = torch.zeros((waves, 8), dtype=torch.int64, device=env.device)
env.lib.igw_debug_set_stamps(env.ctx, st.data_ptr())
acc = []
for t in range(20, 60):
    st.zero_()
    step(t)
    torch.cuda.synchronize()
    acc.append(st.cpu().numpy().copy())
env.lib.igw_debug_set_stamps(env.ctx, None)
a = np.stack(acc)  # [launches, waves, 8]
np.savez_compressed(out, stamps=a)
t = a[:, :, :7].astype(np.float64)
feat = a[:, :, 7]
n_ch, n_rs, n_rt, max_m, n_brk = feat & 0xff, (feat >> 8) & 0xff, (feat >> 16) & 0xff, (feat >> 24) & 0xff, (feat >> 32) & 0xff
life = t[:, :, 6] - t[:, :, 0]
names = ['loads', 'action + hit_test', 'physics', 'world_step tail', 'histogram update', 'rescan/finish/stores']
d = np.diff(t, axis=2)
print(f'lanes/env {gs}: {waves} waves, {a.shape[0]} launches; shader cycles; debug flags {flags}')
print(f'  wave lifetime: mean {life.mean():.0f} p50 {np.percentile(life, 50):.0f} p90 {np.percentile(life, 90):.0f} '
      f'p99 {np.percentile(life, 99):.0f} max(mean over launches) {life.max(1).mean():.0f}')
for i, nm in enumerate(names):
    print(f'  {nm:22s} mean {d[:, :, i].mean():8.0f}  p90 {np.percentile(d[:, :, i], 90):8.0f}  p99 {np.percentile(d[:, :, i], 99):8.0f}  max {d[:, :, i].max():8.0f}')
print('  lifetime by work in the wave:')
for label, m in (('no change, no reset', (n_ch == 0) & (n_rt == 0)), ('1 change', (n_ch == 1) & (n_rt == 0)),
                 ('2 changes', (n_ch == 2) & (n_rt == 0)), ('3-4 changes', (n_ch >= 3) & (n_ch <= 4) & (n_rt == 0)),
                 ('5+ changes', n_ch >= 5), ('rescan', n_rs > 0), ('reset', n_rt > 0), ('break', n_brk > 0),
                 ('sub-steps 2', max_m == 2), ('sub-steps 4', max_m == 4), ('sub-steps 8+', max_m >= 8)):
    if m.any():
        print(f'    {label:20s} share {m.mean():6.3f}  lifetime mean {life[m].mean():8.0f}  p99 {np.percentile(life[m], 99):8.0f}'
              f'  hist phase {d[:, :, 4][m].mean():7.0f}  last phase {d[:, :, 5][m].mean():7.0f}')
# by dispatch order: blocks are dealt to the CUs in index order, so block // 256 is the age rank of a wave on its SIMD
grp = (np.arange(waves) // 4) // max(1, (waves // 4) // 4)
print('  by block dispatch order (= wave age on its SIMD; issue is arbitrated by age):')
for g_ in range(4):
    m = grp == g_
    print(f'    blocks {g_ * (waves // 16):5d}+  lifetime mean {life[:, m].mean():7.0f}  max {life[:, m].max(1).mean():7.0f}  ' +
          '  '.join(f'{nm.split()[0]} {d[:, m, i].mean():6.0f}' for i, nm in enumerate(names)))
young = grp == 3
print('  youngest quarter, no reset, by changed envs in the wave:')
for k in range(8):
    m = young[None, :] & (n_ch == k) & (n_rt == 0)
    if m.sum() > 5:
        print(f'    {k} changes: n {int(m.sum()):6d}  lifetime {life[m].mean():7.0f}  histogram phase {d[:, :, 4][m].mean():6.0f}  physics {d[:, :, 2][m].mean():6.0f}')
# The launch ends with its LAST wave.  s_memtime counts per CU (the counters of different CUs are tens of millions of
# cycles apart), so wave times are comparable only within a CU: block b of the 65,536-env launch runs on CU b % 256
# (blocks b, b + 256, b + 512, b + 768 share a clock), and every time below is relative to the first start on the CU.
if waves == 4096:
    tc = t.reshape(t.shape[0], 4, 256, 4, 7)   # [launch, dispatch round, CU, wave of the block, stamp]
    t0 = tc[..., 0].min(axis=(1, 3), keepdims=True)
    end = (tc[..., 6] - t0).reshape(t.shape[0], waves)
    start = (tc[..., 0] - t0).reshape(t.shape[0], waves)
    cu_span = end.reshape(t.shape[0], 4, 256, 4).max(axis=(1, 3))
    print(f'  per-CU span (first start -> last end on the CU): mean {cu_span.mean():.0f}  slowest CU of a launch {cu_span.max(1).mean():.0f}'
          f'  (the launch waits for that one); start spread within a CU {start.max(1).mean():.0f}')
    print('  the slowest CU\'s span if the waves with ... did not exist:')
    for label, msk in (('5+ changes', n_ch >= 5), ('3+ changes', n_ch >= 3), ('a reset', n_rt > 0), ('a break', n_brk > 0),
                       ('sub-steps 4+', max_m >= 4), ('any change', n_ch >= 1),
                       ('the last dispatch round', np.broadcast_to((np.arange(waves) // 4) >= 768, n_ch.shape))):
        print(f'    {label:24s} (share {msk.mean():.4f}): {np.where(msk, 0, end).max(1).mean():.0f}')
    srt = np.sort(end, axis=1)
    print('  end of the k-th last wave, mean over launches: ' + '  '.join(f'k={k}: {srt[:, -k].mean():.0f}' for k in (1, 2, 4, 8, 16, 41, 205, 2048)))
    top = np.argsort(end, axis=1)[:, -16:]
    print('  phases of the 16 last waves: ' + '  '.join(f'{nm.split()[0]} {np.take_along_axis(d[:, :, i], top, 1).mean():.0f}' for i, nm in enumerate(names)))
    for k in range(4):
        msk = (np.arange(waves) // 4) // 256 == k
        print(f'    dispatch round {k}: start {start[:, msk].mean():5.0f}  end {end[:, msk].mean():6.0f}')
# which waves end last
last = life.argmax(1)
print('  slowest wave per launch: changes', n_ch[np.arange(len(last)), last].tolist())
print('                           resets ', n_rt[np.arange(len(last)), last].tolist())
print('                           rescans', n_rs[np.arange(len(last)), last].tolist())
print('                           m      ', max_m[np.arange(len(last)), last].tolist())
