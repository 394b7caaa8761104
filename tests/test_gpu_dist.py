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
steps = env.num_envs * 8 + int(env.stats()['bad_actions'])
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
