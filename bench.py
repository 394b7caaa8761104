#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the vectorised env.step() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one igw_step_walking launch over every env of the rank (one env.step() per env, reward
and done included, auto-reset of finished episodes inside the launch).  Workload = BASELINE.json
configs[2]: 65,536 parallel envs per GPU, walking Discrete(18), random 20-block targets (rt20, full
maximal_intersection reward), uniform random actions that are already resident in HBM when the timed
region starts.  Weak scaling: every rank owns its own 65,536 envs; no data-path collective.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- HBM roofline of the dominant kernel from ALGORITHMIC bytes per env-step
                  (1274 + 1106 * p bytes, p = measured fraction of env-steps
                  whose block count changed) over the kernel's average duration (HIP events on the
                  launch stream);
  cpu_baseline -- the CPU oracle (plain-C port of the reference algorithm) timed on the host cores on
                  a bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

ENVS_PER_GPU = 65536
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
BYTES_BASE, BYTES_CHANGED = 1274, 1106


def cpu_baseline(seed):
    """Oracle (plain-C port of the reference algorithm) on the host cores: a bounded sample of the same
    workload (rt20 targets, counter-RNG uniform actions, resets included), sized for ~10 s of wall time."""
    from gridworld_amd import workloads
    from oracle import oracle as O
    cores = len(os.sched_getaffinity(0))
    kw = dict(size_reward=False)
    n = int(min(16384, max(256, 64 * cores)))
    tg = workloads.rt20(n, seed).numpy()
    b = O.OracleBatch(n, **kw)
    b.set_tasks(tg)
    b.reset()
    t = time.perf_counter()
    b.rollout_walking(10, seed, autoreset=True, nthreads=cores)  # calibration (also warms the threads)
    rate0 = n * 10 / (time.perf_counter() - t)
    T = int(min(1500, max(250, round(rate0 * 8 / n / 250) * 250)))
    b.reset()
    t = time.perf_counter()
    steps, changed = b.rollout_walking(T, seed, autoreset=True, nthreads=cores)
    dt = time.perf_counter() - t
    # one core, smaller sample
    n1 = max(256, min(1024, n // 8))
    b1 = O.OracleBatch(n1, **kw)
    b1.set_tasks(tg[:n1])
    b1.reset()
    t = time.perf_counter()
    s1, _ = b1.rollout_walking(250, seed, autoreset=True, nthreads=1)
    dt1 = time.perf_counter() - t
    return {'value': steps / dt, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} envs x {T} steps, rt20 targets, counter-RNG uniform actions, resets included '
                      f'({dt:.1f} s on {cores} threads)',
            'value_1core': s1 / dt1, 'sample_1core': f'{n1} envs x 250 steps ({dt1:.1f} s)',
            'p_changed': changed / max(steps, 1)}


def load_traffic():
    """HBM bytes per launch from the committed PMC profile (profiles/*traffic*.json), or None."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*traffic.json'))):
        try:
            with open(p) as f:
                best = json.load(f)
        except Exception:
            pass
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--lanes-per-env', type=int, default=0)
    ap.add_argument('--seed', type=int, default=2024)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fused', action='store_true')
    ap.add_argument('--debug-flags', type=int, default=0, help='timing-only ablations (results invalid)')
    ap.add_argument('--mode', choices=['walking', 'flying'], default='walking',
                    help='walking = BASELINE configs[2] (headline); flying = configs[3]')
    args = ap.parse_args()

    from gridworld_amd import VecGridWorld, dist as gdist, workloads
    rank, local_rank, world = gdist.init()
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # one process per GPU; IGW_SHARE_GPU=1 lets several ranks share one device (CI on a 1-GPU box, gloo)
    device = torch.device('cuda', 0 if os.environ.get('IGW_SHARE_GPU') else local_rank)
    torch.cuda.set_device(device)
    N, K, W = args.envs_per_gpu, args.steps, args.warmup
    env_offset = rank * N  # rank-offset RNG streams / task seeds

    flying = args.mode == 'flying'
    env = VecGridWorld(N, device=device, action_space=args.mode, size_reward=False, max_steps=250,
                       autoreset=True, lanes_per_env=args.lanes_per_env, debug_flags=args.debug_flags)
    env.set_tasks(workloads.rt20(N, seed=args.seed + rank, device=device))
    env.reset()
    # actions for warmup + timed steps are generated on the device before the clock starts
    if flying:
        if args.no_fused is False:
            args.no_fused = True  # the fused rollout kernel is walking-only
        g = torch.Generator(device=device)
        g.manual_seed(args.seed + 7919 * rank)
        mv = torch.rand((W + K, N, 3), generator=g, device=device) * 2 - 1      # movement ~ U(-1, 1)^3
        cam = torch.rand((W + K, N, 2), generator=g, device=device) * 10 - 5    # camera ~ U(-5, 5)^2
        inv = torch.randint(0, 7, (W + K, N), generator=g, device=device, dtype=torch.int32)
        plc = torch.randint(0, 3, (W + K, N), generator=g, device=device, dtype=torch.int32)
        import ctypes as C
        from gridworld_amd import _lib as L

        def step(t):
            L.check(env.lib.igw_step_flying(env.ctx, mv[t].data_ptr(), cam[t].data_ptr(), inv[t].data_ptr(),
                                            plc[t].data_ptr(), env._stream()), 'igw_step_flying')
    else:
        chunk = 256
        actions = [env.fill_actions(min(chunk, W + K - t0), seed=args.seed, t0=t0, env_offset=env_offset)
                   for t0 in range(0, W + K, chunk)]

        def step(t):
            env.step_walking_ptr(actions[t // chunk][t % chunk])

    for t in range(W):
        step(t)
    torch.cuda.synchronize(device)
    st0 = env.stats()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gdist.barrier(device)
    torch.cuda.synchronize(device)
    t_start = time.perf_counter()
    ev0.record()
    for t in range(W, W + K):
        step(t)
    ev1.record()
    torch.cuda.synchronize(device)
    gdist.barrier(device)
    elapsed = time.perf_counter() - t_start
    st1 = env.stats()
    kernel_ms = ev0.elapsed_time(ev1) / K
    total_steps, max_elapsed = gdist.reduce_window(N * K, elapsed, device)

    # secondary: fused T-step rollout (state resident in LDS/registers, in-kernel RNG)
    fused = None
    if not args.no_fused:
        Tf = 250
        env.rollout(Tf, seed=args.seed + 1, t0=0, env_offset=env_offset)  # warm
        torch.cuda.synchronize(device)
        gdist.barrier(device)
        t0 = time.perf_counter()
        env.rollout(Tf, seed=args.seed + 2, t0=0, env_offset=env_offset)
        torch.cuda.synchronize(device)
        gdist.barrier(device)
        f_steps, f_el = gdist.reduce_window(N * Tf, time.perf_counter() - t0, device)
        fused = f_steps / f_el

    if rank != 0:
        return
    lanes = env.cfg.lanes_per_env
    if not lanes:  # the library's automatic choice (include/igw.h: IGW_TARGET_WAVES)
        lanes = 64
        while lanes > 1 and N * lanes // 64 > 4096:
            lanes //= 2
        if lanes == 2:
            lanes = 4
    p = (st1['changed'] - st0['changed']) / float(N * K)
    resets = st1['resets'] - st0['resets']
    bytes_per_step = (BYTES_BASE + (24 if flying else 0)) + BYTES_CHANGED * p  # flying actions are 28 B, not 4
    achieved = N * bytes_per_step / (kernel_ms * 1e-3) / 1e9
    traffic = load_traffic()
    out = {
        'metric': 'env-steps/sec (render=False, vector_state) at N parallel envs, 1/2/4/8 GPU',
        'value': total_steps / max_elapsed,
        'unit': 'env-steps/s',
        'n_gpus': world,
        'steps': K,
        'warmup': W,
        'ms_per_step': max_elapsed / K * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': ('configs[3]: 65,536 parallel envs per GPU, flying action space (continuous movement / '
                                'camera Box), random 20-block targets (rt20), uniform random actions, auto-reset at '
                                'done (max_steps=250)') if flying else
                               ('configs[2]: 65,536 parallel envs per GPU, walking Discrete(18), random 20-block '
                                'targets (rt20), full maximal_intersection reward, uniform random actions, '
                                'auto-reset at done (max_steps=250)'),
                   'envs_per_gpu': N, 'total_envs': N * world, 'lanes_per_env': lanes,
                   'launches_per_step': 1, 'resets_in_window': resets, 'p_changed': p,
                   'fused_rollout_env_steps_per_s': fused},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS,
                     'traffic': None if traffic is None else traffic.get('hbm_bytes_per_launch'),
                     'kernel': 'igw::step_kernel<%d, %d>' % (lanes, 1 if flying else 0), 'kernel_avg_ms': kernel_ms,
                     'algorithmic_bytes_per_env_step': bytes_per_step,
                     'algorithmic_bytes_per_launch': N * bytes_per_step},
    }
    if flying:
        out['roofline']['traffic'] = None  # the committed PMC profile is for the walking kernel
    if world == 1 and not args.no_cpu_baseline and not flying:
        out['cpu_baseline'] = cpu_baseline(args.seed)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
