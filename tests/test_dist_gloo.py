"""Multi-process path on CPU (gloo, world sizes 2 and 8): env sharding, barrier, step-count reduction.
The data path has no collective (envs are independent); this covers what bench.py does around it."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, time, json
    sys.path.insert(0, %r)
    import torch
    from gridworld_amd import dist as gd
    rank, local_rank, world = gd.init(backend='gloo')
    assert world == 2
    total = 65536 * world + 3
    lo, hi = gd.shard_envs(total, rank, world)
    nb = gd.NodeBarrier()
    for i in range(200):            # the shared-memory barrier bench.py brackets its window with
        if i == 100 and rank == 1:
            time.sleep(0.2)         # a late rank holds the other one back
            late = time.perf_counter()
        nb.wait()
        if i == 100 and rank == 0:
            released = time.perf_counter()
    t0 = time.perf_counter()
    for i in range(1000):
        nb.wait()
    nb_us = (time.perf_counter() - t0) / 1000 * 1e6
    stamps = gd.gather_counts(int((released if rank == 0 else late) * 1e6))  # CLOCK_MONOTONIC: one clock for both
    gd.barrier()
    t = time.perf_counter()
    steps = (hi - lo) * 10          # 10 "steps" of this rank's shard
    time.sleep(0.05 * (rank + 1))   # uneven ranks: the window is the max over ranks
    el = time.perf_counter() - t
    gd.barrier()
    tot, mx = gd.reduce_window(steps, el)
    counts = gd.gather_counts(steps)
    via = gd.gather_counts_rccl(steps, None)   # no GPU here: the documented gloo path
    assert via == (counts, 'gloo'), via
    assert gd.reduce_windows([0.1 * (rank + 1), 0.3 - 0.1 * rank]) == [0.2, 0.3]
    assert gd.gather_floats(1.5 + rank) == [1.5, 2.5]
    if rank == 0:
        print(json.dumps(dict(total=tot, max_elapsed=mx, counts=counts, lo=lo, hi=hi, expect=total * 10, nb_us=nb_us,
                              released_after_late_rank=stamps[0] >= stamps[1])), flush=True)
    gd.shutdown()
''') % ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gloo_world2_sharding_and_reduction(tmp_path):
    import json
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for attempt in range(3):  # the free port can be taken between the probe and the rendezvous
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['total'] == d['expect'] == sum(d['counts'])
    assert d['max_elapsed'] >= 0.1
    assert d['lo'] == 0 and d['hi'] == (65536 * 2 + 3) // 2
    assert d['released_after_late_rank'], d
    assert d['nb_us'] < 2000, d  # the node barrier is a spin on shared memory (microseconds on idle cores)


def test_shard_envs_partition():
    from gridworld_amd.dist import shard_envs
    for total in (1, 7, 65536, 524288, 524291):
        for world in (1, 2, 3, 8):
            edges = [shard_envs(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus 2` (no torchrun) must bring up 2 ranks itself: the launcher path, exercised
    without a GPU through --dry-run (ranks rendezvous over gloo and reduce fake counters)."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5',
                          '--dry-run'], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks'] == [0, 1]
    assert d['total_steps'] == 2 * 65536 * 20 and d['steps'] == 20 and d['warmup'] == 5
    assert d['windows_max'] == [2e-3, 4e-3]   # per window the max over the two ranks


def test_bench_dry_run_eight_ranks_on_four_cpus():
    """The 8-rank shape of the driver's scaling run, without GPUs and on FEWER CPUs than ranks (taskset to 4): 8 ranks
    rendezvous, pass the shared-memory barrier of the timing bracket 50 times (its spin backs off to sched_yield, so
    ranks that share a CPU cannot starve each other), reduce the per-window maxima, gather per-rank values, tear the
    group down together, and the launcher relays exactly one JSON line with exit code 0."""
    import json
    import shutil
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'OMP_NUM_THREADS'):
        env.pop(k, None)
    cpus = sorted(os.sched_getaffinity(0))[:4]
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '20', '--warmup', '5', '--dry-run']
    if shutil.which('taskset'):
        cmd = ['taskset', '-c', ','.join(map(str, cpus))] + cmd
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['ranks'] == list(range(8)) == d['config']['ranks_seen']
    assert d['total_steps'] == 8 * 65536 * 20
    assert d['windows_max'] == [8e-3, 16e-3]
    assert d['config']['kernel_us_per_rank'] == [10.0 + r for r in range(8)]
    # the launcher sizes the ranks' thread pools from what the process may really use (affinity / cgroup quota)
    assert 1 <= d['config']['torch_threads'] <= max(1, d['config']['usable_cpus'])
