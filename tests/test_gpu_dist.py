"""GPU box: the RCCL leg of the control plane on real hardware.  A one-GPU box cannot host two ranks (RCCL refuses two
ranks on one device), so this is the most the builder's boxes can show: a mixed gloo + nccl group of ONE rank on cuda:0,
gridworld_amd.dist.gather_counts_rccl through RCCL's all_gather, the step counter of a real stepped batch as payload."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from gridworld_amd import VecGridWorld, workloads, dist as gd
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
dist.init_process_group('cpu:gloo,cuda:nccl', init_method='tcp://127.0.0.1:%(port)d', rank=0, world_size=1)
env = VecGridWorld(4096, autoreset=True)
env.set_tasks(workloads.rt20(4096, seed=0, device=env.device))
env.reset()
acts = env.fill_actions(8, seed=3)
for t in range(8):
    env.step(acts[t])
torch.cuda.synchronize()
steps = int(env.stats()["steps"])   # the device-side counter (IGW_STAT_STEPS)
vals, how = gd.gather_counts_rccl(steps, dev)
floats = gd.gather_floats(1.5)
gd.barrier()
gd.shutdown()
print('RESULT', steps, vals, how, floats)
'''


@pytest.mark.gpu
def test_rccl_gather_single_rank_group():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, port=29541)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT')][-1]
    assert line.startswith('RESULT 32768 [32768] rccl all_gather of one int64 per rank [1.5]'), line


@pytest.mark.gpu
def test_two_ranks_through_torchrun_on_one_shared_gpu():
    """The N > 1 CONTROL PLANE on a 1-GPU box: `bench.py --gpus 2` starts its two ranks through
    `python -m torch.distributed.run` (child processes, started before anything touches the GPU) and IGW_SHARE_GPU=1
    puts both on cuda:0.  Checked: rendezvous, the shared-memory barrier of the timing bracket, the per-window max over
    ranks, the gather of every rank's DEVICE-SIDE step counter (IGW_STAT_STEPS delta of the window) with `value`
    computed from their sum, `cpu_baseline` (rank 0, a one-second sample here) and `roofline` on the N > 1 line,
    a clean exit of both ranks.  RCCL refuses two ranks on one device, so the gather takes the
    agreed gloo fallback here (on a real node it is RCCL over xGMI; the one-rank RCCL gather is the test above).
    NO scaling number can be read from this: both ranks share one GPU."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', IGW_SHARE_GPU='1', IGW_RCCL_TIMEOUT_S='30')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    n, k = 8192, 20
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', str(k), '--warmup', '5', '--envs-per-gpu', str(n),
           '--windows', '3', '--rehearsals', '1', '--cpu-sample-scale', '0.05', '--no-secondary', '--no-api', '--no-fused', '--no-async']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    cfg = line['config']
    assert line['n_gpus'] == 2 and cfg['ranks_seen'] == [0, 1] and cfg['total_envs'] == 2 * n
    assert cfg['device_step_counts_per_rank'] == [n * k, n * k]          # what the kernels of each rank counted
    assert abs(line['value'] - 2 * n * k / (line['ms_per_step'] * 1e-3 * k)) < 1e-6 * line['value']
    assert 'rccl' in cfg['step_count_gather'] or 'gloo' in cfg['step_count_gather']
    assert len(cfg['kernel_us_per_rank']) == 2 and line['scaling'] == 'weak'
    # the N > 1 line is complete (north star: the CPU path timed in the same run, the roofline of the kernel):
    cb = line['cpu_baseline']
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['cores'] >= 1 and 'rank 0' in cb['timed_on'] and 'sample' in cb
    rf = line['roofline']
    assert rf['bound'] in ('issue', 'hbm') and rf['kernel_avg_ms'] > 0 and len(cfg['kernel_us_per_rank']) == 2
    hbm = rf.get('hbm', rf)
    assert hbm['bound'] == 'hbm' and hbm['peak'] == 8000.0 and hbm['achieved'] > 0
