#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the vectorised env.step() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torchrun environment, bench.py starts its own N ranks (one per GPU) as a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` BEFORE touching
the GPU and relays rank 0's JSON line; launched under torchrun it is a rank.

One "step" = one igw_step_walking launch over every env of the rank (one env.step() per env, reward
and done included, auto-reset of finished episodes inside the launch).  Workload = BASELINE.json
configs[2]: 65,536 parallel envs per GPU, walking Discrete(18), random 20-block targets (rt20, full
maximal_intersection reward), uniform random actions that are already resident in HBM when the timed
region starts.  Weak scaling: every rank owns its own 65,536 envs; no data-path collective.

Steady state before the clock: episodes are de-synchronised (every env starts at a random step of its
episode, then an untimed pre-roll of at least 250 steps; 0.3 s more of it right before the clock, after the
graph capture, so the GPU is at its working clocks), so any timed window -- also a 20-step
one -- sees the steady-state fraction of grid-changing steps and about N/250 auto-resets per launch.  The K
timed launches are 2 eager launches followed by ONE HIP-graph replay of the other K - 2 (captured and
instantiated before the clock), inside the barrier / synchronize bracket, so a short window is
kernel-bound, not host-launch-bound (--no-graph times eager launches only).  The whole W + K sequence is
rehearsed a fixed 3 times, untimed, before the measured (always the last) pass: the first passes through the
host launch paths cost tens of microseconds more than later ones (--rehearsals 0 shows it, every pass is
reported in config.rehearsal_ms_per_step; tools/window_variants.py measures the launch paths).

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline     -- HBM roofline of the dominant kernel from ALGORITHMIC bytes per env-step
                  (1274 + 1106 * p bytes, p = measured fraction of env-steps
                  whose block count changed) over the kernel's average duration (HIP events on the
                  launch stream); `traffic` / `hbm_measured_gbs` = PMC bytes of the committed profile;
  issue        -- the instruction-issue side (the real limiter): VALU instructions per
                  env-step and per wave from the committed SQ counter profile, waves per SIMD;
  cpu_baseline -- the CPU oracle (plain-C port of the reference algorithm) timed on the host cores on
                  a bounded sample of the same workload (N = 1 only).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
MAX_STEPS = 250
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
BYTES_BASE, BYTES_CHANGED = 1274, 1106


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--lanes-per-env', type=int, default=0)
    ap.add_argument('--seed', type=int, default=2024)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fused', action='store_true')
    ap.add_argument('--no-async', action='store_true', help='skip the secondary two-sub-batch measurement')
    ap.add_argument('--no-graph', action='store_true', help='time eager launches instead of one HIP-graph replay')
    ap.add_argument('--rehearsals', type=int, default=3,
                    help='untimed passes through the whole W + K sequence before the measured one (reported)')
    ap.add_argument('--lockstep', action='store_true',
                    help='skip the episode de-synchronisation (round-1 behaviour: all envs at the same episode step)')
    ap.add_argument('--mode', choices=['walking', 'flying'], default='walking',
                    help='walking = BASELINE configs[2] (headline); flying = configs[3]')
    ap.add_argument('--debug-flags', type=int, default=0,
                    help='IGW_DIAG=1 only: timing-only ablation switches of the diagnostic library (results invalid)')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / rendezvous check without a GPU: ranks reduce fake counters over gloo')
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """--gpus N > 1 outside torchrun: become the launcher.  This process never touches the GPU; the N
    ranks are children of torch.distributed.run and rank 0's JSON line is relayed to our stdout."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    rc = 1
    for attempt in range(3):  # the probed port can be taken before the rendezvous binds it
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
        got = False
        for line in proc.stdout:
            if line.startswith('{'):
                sys.stdout.write(line)
                sys.stdout.flush()
                got = True
            else:
                sys.stderr.write(line)
        rc = proc.wait()
        if rc == 0 and got:
            return 0
        if got:
            break
    return rc or 1


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


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
    return {'value': steps / dt, 'unit': 'env-steps/s', 'cores': cores, 'cpu_model': cpu_model(), 'kind': 'port',
            'sample': f'{n} envs x {T} steps, rt20 targets, counter-RNG uniform actions, resets included '
                      f'({dt:.1f} s on {cores} threads)',
            'value_1core': s1 / dt1, 'sample_1core': f'{n1} envs x 250 steps ({dt1:.1f} s)',
            'p_changed': changed / max(steps, 1)}


def load_profile(suffix, flying=False):
    """Latest committed profile summary profiles/r*_<suffix> (walking) or r*_flying_<suffix> (json), or None."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_' + ('flying_' if flying else '') + suffix))):
        if ('flying' in os.path.basename(p)) != flying:
            continue
        try:
            with open(p) as f:
                best = json.load(f)
            best['_file'] = os.path.relpath(p, ROOT)
        except Exception:
            pass
    return best


def auto_lanes(n):
    """The library's automatic group width (include/igw.h: IGW_AUTO_*_MAX)."""
    return 32 if n <= 1024 else 16 if n <= 4096 else 8 if n <= 24576 else 4


def dry_run(args):
    """No GPU: proves the launcher brought up `world` ranks that can rendezvous and reduce."""
    from gridworld_amd import dist as gdist
    rank, local_rank, world = gdist.init(backend='gloo')
    gdist.barrier()
    nb = gdist.NodeBarrier()  # the barrier of the timing bracket
    for _ in range(3):
        nb.wait()
    total, mx = gdist.reduce_window(args.envs_per_gpu * args.steps, 1e-3 * (rank + 1))
    ranks = gdist.gather_counts(rank)
    if rank == 0:
        print(json.dumps({'metric': 'dry-run', 'n_gpus': world, 'ranks': ranks, 'total_steps': total,
                          'max_elapsed': mx, 'steps': args.steps, 'warmup': args.warmup}))


def main():
    args = parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ:
        raise SystemExit(self_launch(args))
    if args.dry_run:
        return dry_run(args)

    import torch
    from gridworld_amd import VecGridWorld, dist as gdist, workloads
    rank, local_rank, world = gdist.init()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # one process per GPU; IGW_SHARE_GPU=1 lets several ranks share one device (CI on a 1-GPU box)
    device = torch.device('cuda', 0 if os.environ.get('IGW_SHARE_GPU') else local_rank)
    torch.cuda.set_device(device)
    N, K, W = args.envs_per_gpu, args.steps, args.warmup
    env_offset = rank * N  # rank-offset RNG streams / task seeds

    flying = args.mode == 'flying'
    env = VecGridWorld(N, device=device, action_space=args.mode, size_reward=False, max_steps=MAX_STEPS,
                       autoreset=True, lanes_per_env=args.lanes_per_env, debug_flags=args.debug_flags)
    env.set_tasks(workloads.rt20(N, seed=args.seed + rank, device=device))
    env.reset()
    g = torch.Generator(device=device)
    g.manual_seed(args.seed + 7919 * rank)
    if not args.lockstep:
        # every env starts at a random step of its episode (GridWorld.step_no, agent record bytes 48..49)
        sn = torch.randint(0, MAX_STEPS, (N,), generator=g, device=device, dtype=torch.int32)
        env.agent_buf[:, 48] = (sn & 0xff).to(torch.uint8)
        env.agent_buf[:, 49] = (sn >> 8).to(torch.uint8)

    # actions for pre-roll, warmup and the timed steps are generated on the device before the clock starts
    if flying:
        fly_fused = not args.no_fused
        args.no_fused = True  # the fused rollout with in-kernel random actions is walking-only
        from gridworld_amd import _lib as L

        def fly_actions(n_steps):
            mv = torch.rand((n_steps, N, 3), generator=g, device=device) * 2 - 1      # movement ~ U(-1, 1)^3
            cam = torch.rand((n_steps, N, 2), generator=g, device=device) * 10 - 5    # camera ~ U(-5, 5)^2
            inv = torch.randint(0, 7, (n_steps, N), generator=g, device=device, dtype=torch.int32)
            plc = torch.randint(0, 3, (n_steps, N), generator=g, device=device, dtype=torch.int32)
            return mv, cam, inv, plc

        def fly_step(a, t):
            L.check(env.lib.igw_step_flying(env.ctx, a[0][t].data_ptr(), a[1][t].data_ptr(), a[2][t].data_ptr(),
                                            a[3][t].data_ptr(), env._stream()), 'igw_step_flying')
        def busy(seconds, min_steps):
            """Untimed stepping with FRESH random actions: the pre-roll to the steady state / the clock ramp."""
            t_ramp, n_pre = time.perf_counter(), 0
            while n_pre < min_steps or time.perf_counter() - t_ramp < seconds:
                pre = fly_actions(50)
                for t in range(50):
                    fly_step(pre, t)
                torch.cuda.synchronize(device)
                n_pre += 50
        if not args.lockstep:
            busy(0.0, MAX_STEPS)  # at least one episode length
        acts = fly_actions(W + K)
        # the timed loop passes pre-computed device pointers: slicing a tensor per step costs more host time
        # than the launch itself
        fptrs = [tuple(a[t].data_ptr() for a in acts) for t in range(W + K)]
        fly_fn, ctx_h = env.lib.igw_step_flying, env.ctx

        def step(t, stream):
            p = fptrs[t]
            rc = fly_fn(ctx_h, p[0], p[1], p[2], p[3], stream)
            if rc:
                L.check(rc, 'igw_step_flying')
    else:
        pre_rolled = [0]

        def busy(seconds, min_steps):
            """Untimed stepping with fresh in-kernel random actions (fused rollout): the pre-roll to the steady
            state / the clock ramp."""
            t_ramp, n_pre = time.perf_counter(), 0
            while n_pre < min_steps or time.perf_counter() - t_ramp < seconds:
                env.rollout(MAX_STEPS, seed=args.seed + 17 + pre_rolled[0], t0=pre_rolled[0], env_offset=env_offset)
                torch.cuda.synchronize(device)
                n_pre += MAX_STEPS
                pre_rolled[0] += MAX_STEPS
        if not args.lockstep:
            busy(0.0, MAX_STEPS)  # at least one episode length
        chunk = 256
        actions = [env.fill_actions(min(chunk, W + K - t0), seed=args.seed, t0=t0, env_offset=env_offset)
                   for t0 in range(0, W + K, chunk)]

        # the timed loop passes pre-computed device pointers: slicing a tensor per step costs more host time
        # than the launch itself (tools/window_variants.py: 3.5 us per launch this way, 9-17 us through the
        # tensor-slicing wrapper)
        wptrs = [actions[t // chunk][t % chunk].data_ptr() for t in range(W + K)]
        walk_fn, ctx_h = env.lib.igw_step_walking, env.ctx
        from gridworld_amd import _lib as L

        def step(t, stream):
            rc = walk_fn(ctx_h, wptrs[t], stream)
            if rc:
                L.check(rc, 'igw_step_walking')

    # The timed launches as a short eager head + ONE HIP graph for the rest.  Capture records launches without
    # running them (the entry points never synchronise or allocate) and instantiation happens here, before the
    # warm-up and the clock.  The eager head starts the GPU within a few microseconds of the clock; the graph is
    # launched while those kernels run, so its launch latency (10-16 us) is off the critical path.
    graph, head = None, K
    if not args.no_graph and K > 2:
        head = 2
        graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device=device)
        cap.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.graph(graph, stream=cap):
            cap_h = ctypes.c_void_p(cap.cuda_stream)
            for t in range(W + head, W + K):
                step(t, cap_h)
        torch.cuda.current_stream(device).wait_stream(cap)
        graph.replay()  # part of the setup: the first launch of a graph also uploads it
        torch.cuda.synchronize(device)
    cur_h = env._stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the barrier of the timing bracket: ranks of one node meet in shared memory (microseconds; an RCCL barrier
    # costs 50-100 us, a quarter of a 20-step window)
    node_barrier = gdist.NodeBarrier()

    def window():
        """W untimed warm-up steps, then the clock around exactly K steps.  Returns (wall seconds, counters before
        the clock as a device tensor, host timeline)."""
        # warm-up right before the clock; the counters are snapshotted on the device, not read, so nothing idles
        # the GPU between warm-up and clock
        for t in range(W):
            step(t, cur_h)
        before = env.stats_buf.sum(0)
        ev0.record()   # torch creates the HIP events lazily at their first record(): not inside the clock
        ev1.record()
        ev1.query()
        node_barrier.wait()
        torch.cuda.synchronize(device)
        if K == 1:
            ev0.record()
        t_start = time.perf_counter()
        step(W, cur_h)
        # The HIP-event window (kernel duration for the roofline) opens behind the first timed launch and spans the
        # other K - 1: an event recorded on the idle stream would be processed at once and the window would then
        # contain the launch latency of the first kernel, not only kernels.  The GPU is busy with that first launch
        # while the host records, so the wall clock does not see it.
        if K > 1:
            ev0.record()
        for t in range(W + 1, W + head):
            step(t, cur_h)
        t_b = time.perf_counter()
        if graph is not None:
            graph.replay()
        t_c = time.perf_counter()
        ev1.record()
        t_d = time.perf_counter()
        while not ev1.query():  # spin: a blocking synchronize sleeps on an interrupt and wakes tens of us late
            pass
        t_e = time.perf_counter()
        torch.cuda.synchronize(device)
        t_f = time.perf_counter()
        node_barrier.wait()
        t_end = time.perf_counter()
        return t_end - t_start, before, (t_b - t_start, t_c - t_b, t_d - t_c, t_e - t_d, t_f - t_e, t_end - t_f)

    # A fixed number of untimed rehearsals of the whole sequence first: the first passes through these host code
    # paths (Python bytecode, ctypes thunks, the HIP runtime's launch and graph-launch paths) cost 30-80 us more
    # than later ones, which is 10-20 % of a 20-step window and nothing to do with the step kernel.  A rehearsal
    # steps the envs like any other warm-up step; the measured pass is always the LAST one, a complete window of
    # its own, and the rehearsals' ms/step are reported next to it (config.rehearsal_ms_per_step).
    # ... and, right before them, 0.3 s of untimed stepping: task upload, graph capture and instantiation leave the
    # GPU idle for tens of milliseconds and its clocks drop; a 20-step window (0.3 ms) would otherwise be measured
    # on the ramp (16.3 us per launch instead of 15.3)
    if not args.lockstep:
        busy(0.3, 0)
    rehearsal_ms = [round(1e3 * window()[0] / K, 5) for _ in range(args.rehearsals)]
    elapsed, st0_dev, host_tl = window()
    if os.environ.get('IGW_BENCH_TRACE'):
        print('host us: head launches %.1f | graph launch %.1f | ev1.record %.1f | spin %.1f | synchronize %.1f | '
              'barrier %.1f | total %.1f' % (tuple(1e6 * x for x in host_tl) + (1e6 * elapsed,)), file=sys.stderr)
    from gridworld_amd import _lib as _L
    s0 = st0_dev.cpu()
    st0 = {'changed': int(s0[_L.STAT_CHANGED]), 'resets': int(s0[_L.STAT_RESETS])}
    st1 = env.stats()
    kernel_ms = ev0.elapsed_time(ev1) / max(K - 1, 1)  # the event window spans the last K - 1 timed launches
    total_steps, max_elapsed = gdist.reduce_window(N * K, elapsed, device)
    n_ranks = len(gdist.gather_counts(rank, device))

    # secondary: fused T-step rollout (state resident in LDS/registers, in-kernel RNG)
    fused = fused_rec = None
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
        # the same fused loop over RECORDED actions (igw_rollout_walking_actions): the first chunk of the timed actions
        Tr = actions[0].shape[0]
        env.rollout_actions(actions[0])  # warm
        torch.cuda.synchronize(device)
        gdist.barrier(device)
        t0 = time.perf_counter()
        env.rollout_actions(actions[0])
        torch.cuda.synchronize(device)
        gdist.barrier(device)
        r_steps, r_el = gdist.reduce_window(N * Tr, time.perf_counter() - t0, device)
        fused_rec = r_steps / r_el

    if flying and fly_fused:   # the fused loop over the recorded flying actions (igw_rollout_flying_actions)
        rec = dict(movement=acts[0], camera=acts[1], inventory=acts[2], placement=acts[3])
        Tr = acts[0].shape[0]
        env.rollout_actions(rec)  # warm
        torch.cuda.synchronize(device)
        gdist.barrier(device)
        t0 = time.perf_counter()
        env.rollout_actions(rec)
        torch.cuda.synchronize(device)
        gdist.barrier(device)
        r_steps, r_el = gdist.reduce_window(N * Tr, time.perf_counter() - t0, device)
        fused_rec = r_steps / r_el

    # secondary: the same kernel as two independent sub-batches on two HIP streams (VecGridWorld.split, the
    # EnvPool-style asynchronous mode): no barrier between the halves, so the start of one half's next step
    # fills the end of the other's -- what one launch per step over the whole batch cannot do
    async2 = None
    if not args.no_async and not flying and N % 2 == 0:
        subs = env.split(2)
        jobs = [(sb.ctx, ctypes.c_void_p(sb.stream.cuda_stream), 4 * sb.lo) for sb in subs]
        Ka = min(K, W + K)
        torch.cuda.synchronize(device)
        for timed in (False, True):
            gdist.barrier(device)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for t in range(Ka):
                for ctx_s, st_s, off in jobs:
                    rc = walk_fn(ctx_s, wptrs[t] + off, st_s)
                    if rc:
                        L.check(rc, 'igw_step_walking')
            torch.cuda.synchronize(device)
            gdist.barrier(device)
            a_el = time.perf_counter() - t0
        a_steps, a_el = gdist.reduce_window(N * Ka, a_el, device)
        async2 = a_steps / a_el
        del subs

    if rank != 0:
        return
    lanes = env.cfg.lanes_per_env or auto_lanes(N)
    p = (st1['changed'] - st0['changed']) / float(N * K)
    resets = st1['resets'] - st0['resets']
    bytes_per_step = (BYTES_BASE + (24 if flying else 0)) + BYTES_CHANGED * p  # flying actions are 28 B, not 4
    achieved = N * bytes_per_step / (kernel_ms * 1e-3) / 1e9
    traffic = load_profile('traffic.json', flying)
    issue = load_profile('issue.json', flying)
    hbm_bytes = None if traffic is None else traffic.get('hbm_bytes_per_launch')
    out = {
        'metric': 'env-steps/sec (render=False, vector_state) at N parallel envs, 1/2/4/8 GPU',
        'value': total_steps / max_elapsed,
        'unit': 'env-steps/s',
        'n_gpus': n_ranks,
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
                   'envs_per_gpu': N, 'total_envs': N * n_ranks, 'lanes_per_env': lanes,
                   'launches_per_step': 1,
                   'timed_as': 'eager launches' if graph is None else f'{head} eager launches + one HIP-graph replay of the other {K - head}',
                   'episodes': 'lock-step' if args.lockstep else 'de-synchronised (random episode phase + pre-roll)',
                   'setup': 'untimed: task upload, pre-roll of >= 250 steps with fresh random actions (steady state), graph capture + one replay, 0.3 s more of such stepping (clock ramp), %d untimed rehearsals of the W + K sequence (host code paths warm), then the W warm-up steps and the clock' % args.rehearsals,
                   'rehearsal_ms_per_step': rehearsal_ms,
                   'resets_in_window': resets, 'p_changed': p,
                   'fused_rollout_env_steps_per_s': fused,
                   'fused_rollout_recorded_actions_env_steps_per_s': fused_rec,
                   'async_2_subbatches_env_steps_per_s': async2},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': hbm_bytes,
                     'kernel': 'igw::step_kernel<%d, %d, false>' % (lanes, 1 if flying else 0), 'kernel_avg_ms': kernel_ms,
                     'algorithmic_bytes_per_env_step': bytes_per_step,
                     'algorithmic_bytes_per_launch': N * bytes_per_step,
                     # what the memory system really moved (committed PMC profile) over this run's kernel time:
                     # the kernel is latency / issue bound, not HBM bound -- see `issue` and DESIGN.md section 5
                     'hbm_measured_gbs': None if hbm_bytes is None else hbm_bytes / (kernel_ms * 1e-3) / 1e9,
                     'hbm_measured_frac': None if hbm_bytes is None else hbm_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     'traffic_profile': None if traffic is None else traffic.get('_file')},
    }
    if issue is not None:
        out['issue'] = {k: v for k, v in issue.items() if not k.startswith('_') and k != 'raw'}
        out['issue']['profile'] = issue.get('_file')
    if world == 1 and not args.no_cpu_baseline and not flying:
        out['cpu_baseline'] = cpu_baseline(args.seed)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
