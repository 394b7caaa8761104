#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the vectorised env.step() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torchrun environment, bench.py starts its own N ranks (one per GPU) as a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` BEFORE touching
the GPU and relays rank 0's JSON line; launched under torchrun it is a rank.

One "step" = one igw_step_walking launch over every env of the rank (one env.step() per env, reward
and done included, auto-reset of finished episodes inside the launch).  Headline workload = BASELINE.json
configs[2]: 65,536 parallel envs per GPU, walking Discrete(18), random 20-block targets (rt20, full
maximal_intersection reward), uniform random actions that are already resident in HBM when the timed
region starts.  Weak scaling: every rank owns its own 65,536 envs; no data-path collective.

Steady state before the clock: episodes are de-synchronised (every env starts at a random step of its
episode, then an untimed pre-roll of at least 250 steps; 0.3 s more of it right before the clock, after the
graph capture, so the GPU is at its working clocks), so any timed window -- also a 20-step one -- sees the
steady-state fraction of grid-changing steps and about N/250 auto-resets per launch.

A WINDOW is the contract's measurement: W untimed warm-up steps, then the clock around EXACTLY K steps,
bracketed by a barrier + torch.cuda.synchronize() on both sides; the K timed launches are 6 eager launches
followed by ONE HIP-graph replay of the other K - 6 (captured and instantiated before the clock; --no-graph
times eager launches only), so a short window is kernel-bound, not host-launch-bound.  The run makes
--rehearsals untimed-in-spirit passes first (host code paths warm; reported) and then --windows complete measured
windows; before EVERY pass the W + K action buffers are refilled on the device with fresh random actions (same
buffers, so the captured graph stays valid; untimed), so no pass replays an action it has seen.  `value` is the
MEDIAN window (max over ranks per window); every window is listed in config.windows_ms_per_step.

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline     -- what bounds the dominant kernel: top level `bound: "issue"`, `achieved` = `frac` = VALU issue
                  utilisation (VALU instructions per wavefront of the committed SQ profile x wavefronts per SIMD x 4
                  cycles over the kernel's average duration measured in THIS run with HIP events on the launch stream),
                  `traffic` = PMC bytes per launch of the committed rocprofv3 profile; `roofline.hbm` = those bytes over
                  the run's kernel time against the 8 TB/s (with `design_bytes_per_env_step`, `wasted_traffic`, the
                  SURVEY 8(d) convention figure and a note on what the counters count at this batch size);
                  `roofline.issue` = the SQ counters behind the top level.  Profiles carry the build id of the library
                  they were taken with: `stale` says whether that is the library being timed;
  cpu_baseline -- the CPU oracle (plain-C port of the reference algorithm) timed on the host cores on bounded
                  samples: the headline workload on all usable cores and on one, BASELINE configs[0] (1 env,
                  DUMMY_TASK-equivalent, 1,000 random steps, 1 core) and configs[3] (flying); on rank 0 at every N;
  config.flying / config.cdm / config.small / config.facade_1env -- secondary measurements of the same window
                  machinery in the default single-GPU run: BASELINE configs[3], the real IGLU targets with partial
                  starting grids, configs[1] (4,096 envs) and configs[0] through the 1-env gym facade;
  config.sweep / config.large -- the headline workload at larger batches per GPU (envs per GPU is a tunable), the
                  largest (2,097,152 envs: beyond the Infinity Cache) with its own PMC profile: the HBM-bound regime.
"""
import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
MAX_STEPS = 250
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
BYTES_BASE, BYTES_CHANGED = 1274, 1106
GPU_CLOCK_GHZ = 2.4
CDM_GOALS = os.path.join(ROOT, 'tests', 'golden', 'cdm_goals.npz')


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--lanes-per-env', type=int, default=0)
    ap.add_argument('--seed', type=int, default=2024)
    ap.add_argument('--windows', type=int, default=15, help='complete measured windows; value = their median')
    ap.add_argument('--rehearsals', type=int, default=3,
                    help='passes through the whole W + K sequence before the measured windows (reported, not counted)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-scale', type=float, default=1.0,
                    help='scales the CPU baseline\'s bounded samples (1.0 = about 25 s of CPU work; tests use 0.05)')
    ap.add_argument('--sweep', default='131072,262144,524288,2097152',
                    help='envs-per-GPU sizes of config.sweep (default single-GPU run); the largest one >= 2,097,152 is also '
                         'config.large, the batch whose per-step working set exceeds the 256 MB Infinity Cache; "" = none')
    ap.add_argument('--no-fused', action='store_true')
    ap.add_argument('--no-async', action='store_true', help='(kept for old command lines; the two-sub-batch figure needs --async now)')
    ap.add_argument('--async', dest='do_async', action='store_true',
                    help='also time two sub-batches launched eagerly on two streams (host-bound at long windows; profiles/r05_chains.txt)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the flying / cdm secondary windows of the default run')
    ap.add_argument('--no-graph', action='store_true', help='time eager launches instead of one HIP-graph replay')
    ap.add_argument('--chains', type=int, default=1,
                    help='capture the timed steps as this many independent chains of sub-batch launches (VecGridWorld.capture_steps(chains=))')
    ap.add_argument('--chain-sweep', default='',
                    help='also measure these chain counts, e.g. 2,4 (config.chain_sweep); measured slower than one chain on MI355X '
                         '(profiles/r05_chains.txt), so not part of the default run')
    ap.add_argument('--no-api', action='store_true', help='skip the public-API loop measurements (config.api_*)')
    ap.add_argument('--lockstep', action='store_true',
                    help='skip the episode de-synchronisation (round-1 behaviour: all envs at the same episode step)')
    ap.add_argument('--mode', choices=['walking', 'flying'], default='walking',
                    help='walking = BASELINE configs[2] (headline); flying = configs[3]')
    ap.add_argument('--workload', choices=['rt20', 'cdm'], default='rt20',
                    help='rt20 = random 20-block targets, empty start (BASELINE); cdm = IGLU CDM structures with partial starting grids')
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


def usable_cpus():
    """CPUs this process can really run on at once: the affinity mask capped by the cgroup's CFS quota (the GPU
    boxes show 256 CPUs in the mask and grant 16)."""
    affinity = len(os.sched_getaffinity(0))
    quota = cgroup_cpu_quota()
    return int(max(1, min(affinity, int(quota + 0.999) if quota else affinity)))


def self_launch(args):
    """--gpus N > 1 outside torchrun: become the launcher.  This process never touches the GPU; the N
    ranks are children of torch.distributed.run and rank 0's JSON line is relayed to our stdout."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, usable_cpus() // args.gpus)))
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
        if got:
            if rc != 0:   # the result line is out, but a rank failed afterwards: the failure stays visible in our exit code
                sys.stderr.write(f'bench.py: torch.distributed.run exited with {rc} after the result line was relayed\n')
            return rc
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


def cgroup_cpu_quota():
    """CPUs this process may use according to its cgroup's CFS quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or
    None when there is no quota.  os.sched_getaffinity does not see such a limit."""
    try:
        with open('/proc/self/cgroup') as f:
            rel = [l.strip().split(':', 2)[2] for l in f if l.startswith('0::')]
        paths = ['/sys/fs/cgroup' + (rel[0] if rel else '') + '/cpu.max', '/sys/fs/cgroup/cpu.max']
        for p in paths:
            if os.path.exists(p):
                q, period = open(p).read().split()[:2]
                if q != 'max':
                    return float(q) / float(period)
                break
    except Exception:
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        period = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            return q / period
    except Exception:
        pass
    return None


def cpu_baseline(seed, scale=1.0):
    """Oracle (plain-C port of the reference algorithm) on the host cores, bounded samples (about 25 s in all):
    the headline workload (rt20 targets, counter-RNG uniform actions, resets included) on every usable core and on
    one; BASELINE configs[3] (flying) likewise; BASELINE configs[0] (1 env, DUMMY_TASK-equivalent, size_reward=True,
    1,000 random steps) on one core."""
    import numpy as np
    from gridworld_amd import workloads
    from oracle import oracle as O
    affinity = len(os.sched_getaffinity(0))
    quota = cgroup_cpu_quota()
    cores = usable_cpus()   # threads actually used: no more than the cgroup lets run at once

    def timed(batch, fn, T, nthreads):
        t = time.perf_counter()
        steps, changed = fn(T, seed, autoreset=True, nthreads=nthreads)
        dt = time.perf_counter() - t
        return steps / dt, dt, changed / max(steps, 1)

    def leg(mode, budget_s):
        kw = dict(size_reward=False) if mode == 'walking' else dict(size_reward=False, action_space='flying')
        n = int(min(16384, max(256, 1024 * cores)))   # (about 10 s of CPU work on the box's 16 granted cores)
        if scale < 0.25:   # (tests: a sample of a second or two)
            n = max(256, n // 8)
        tg = workloads.rt20(n, seed).numpy()
        b = O.OracleBatch(n, **kw)
        b.set_tasks(tg)
        b.reset()
        fn = b.rollout_walking if mode == 'walking' else b.rollout_flying
        rate0, _, _ = timed(b, fn, 10, cores)  # calibration (also warms the threads)
        T = int(min(1500, max(250, round(rate0 * budget_s / n / 250) * 250)))
        b.reset()
        rate, dt, p = timed(b, fn, T, cores)
        n1 = max(128, min(1024, n // 8)) if mode == 'walking' else 256
        b1 = O.OracleBatch(n1, **kw)
        b1.set_tasks(tg[:n1])
        b1.reset()
        fn1 = b1.rollout_walking if mode == 'walking' else b1.rollout_flying
        rate1, dt1, _ = timed(b1, fn1, 250, 1)
        return rate, dt, p, n, T, rate1, dt1, n1

    rate, dt, p, n, T, rate1, dt1, n1 = leg('walking', 8.0 * scale)
    frate, fdt, fp, fn_, fT, frate1, fdt1, fn1 = leg('flying', 6.0 * scale)
    # BASELINE configs[0]: the loop of examples/run_env.py (1 env, DUMMY_TASK-equivalent, size_reward default True)
    dummy = np.zeros((1, 9, 11, 11), np.int8)
    dummy[0, 8, 10, 10] = 1
    b0 = O.OracleBatch(1, size_reward=True)
    b0.set_tasks(dummy, invariant=False)
    b0.reset()
    b0.rollout_walking(1000, seed, autoreset=True, nthreads=1)  # warm
    b0.reset()
    rate0, dt0, _ = timed(b0, b0.rollout_walking, 1000, 1)
    return {'value': rate, 'unit': 'env-steps/s', 'cores': cores, 'cpu_model': cpu_model(), 'kind': 'port',
            'sample': f'{n} envs x {T} steps, rt20 targets, counter-RNG uniform actions, resets included '
                      f'({dt:.1f} s on {cores} threads)',
            'affinity_cpus': affinity, 'cgroup_cpu_quota': quota,
            'effective_cores': round(rate / rate1, 2),   # measured: N-thread rate / 1-thread rate
            'value_1core': rate1, 'sample_1core': f'{n1} envs x 250 steps ({dt1:.1f} s)',
            'p_changed': p,
            'flying': {'value': frate, 'unit': 'env-steps/s', 'cores': cores, 'workload': 'configs[3]',
                       'sample': f'{fn_} envs x {fT} steps, rt20 targets, uniform movement / camera / inventory / '
                                 f'placement from the counter RNG, resets included ({fdt:.1f} s on {cores} threads)',
                       'value_1core': frate1, 'sample_1core': f'{fn1} envs x 250 steps ({fdt1:.1f} s)', 'p_changed': fp},
            'config0': {'value': rate0, 'unit': 'env-steps/s', 'cores': 1, 'workload': 'configs[0]',
                        'sample': f'1 env, DUMMY_TASK-equivalent, size_reward=True, 1000 counter-RNG uniform steps '
                                  f'({dt0 * 1e3:.1f} ms on 1 thread)'}}


def load_profile(suffix, kind='', library_build_id=None, profiles_dir=None):
    """Latest committed profile summary profiles/r*_<suffix> (the headline workload), r*_flying_<suffix> or
    r*_cdm_<suffix> (kind = 'flying' / 'cdm'), or None.
    The summary carries the build id of the library it was taken with (tools/summarize_profile.py); `stale` says
    whether that is another build than the one being timed (`library_build_id` = igw_build_id() of the loaded
    library).  A profile without a build id (rounds 1-4) is stale by definition."""
    import glob
    best = None
    d = profiles_dir or os.path.join(ROOT, 'profiles')
    kind = 'flying' if kind is True else (kind or '')
    for p in sorted(glob.glob(os.path.join(d, 'r[0-9]*_' + (kind + '_' if kind else '') + suffix))):
        if os.path.basename(p).split('_', 1)[1] != (kind + '_' if kind else '') + suffix:
            continue   # (r05_flying_traffic.json is not a walking profile)
        try:
            with open(p) as f:
                best = json.load(f)
            best['_file'] = os.path.relpath(p, ROOT)
        except Exception:
            pass
    if best is not None:
        best['stale'] = library_build_id is None or best.get('build_id') != library_build_id
    return best


def auto_lanes(n):
    """The library's automatic group width (include/igw.h: IGW_AUTO_*_MAX; one host-side copy: gridworld_amd/_lib.py)."""
    from gridworld_amd import _lib
    return _lib.auto_lanes(n)


def dry_run(args):
    """No GPU: proves the launcher brought up `world` ranks that can rendezvous, meet in the shared-memory barrier of
    the timing bracket, reduce the windows and gather the per-rank values -- everything of a multi-rank run but the
    kernels."""
    from gridworld_amd import dist as gdist
    rank, local_rank, world = gdist.init(backend='gloo')
    try:
        gdist.barrier()
        nb = gdist.NodeBarrier()  # the barrier of the timing bracket
        for _ in range(50):
            nb.wait()
        total, mx = gdist.reduce_window(args.envs_per_gpu * args.steps, 1e-3 * (rank + 1))
        wins = gdist.reduce_windows([1e-3 * (rank + 1), 2e-3 * (world - rank)])
        ranks = gdist.gather_counts(rank)
        kernel_us = gdist.gather_floats(10.0 + rank)
        if rank == 0:
            print(json.dumps({'metric': 'dry-run', 'n_gpus': world, 'ranks': ranks, 'total_steps': total,
                              'max_elapsed': mx, 'windows_max': wins, 'steps': args.steps, 'warmup': args.warmup,
                              'config': {'ranks_seen': ranks, 'kernel_us_per_rank': kernel_us,
                                         'usable_cpus': usable_cpus(), 'torch_threads': _torch_threads()}}), flush=True)
    except BaseException:
        gdist.shutdown(barrier=False)
        raise
    gdist.shutdown()


def _torch_threads():
    import torch
    return torch.get_num_threads()


def spin_until(pred, spins=2000):
    """Busy-wait on `pred` for a few microseconds, then yield the CPU between polls: with 8 ranks on a 16-CPU quota
    next to RCCL proxy and watchdog threads an unbounded spin can starve the thread it is waiting for."""
    n = 0
    while not pred():
        n += 1
        if n > spins:
            time.sleep(0) if n < 20 * spins else time.sleep(50e-6)


class Runner:
    """One workload on this rank's GPU through the PUBLIC API (gridworld_amd.VecGridWorld): env, tasks, the W + K
    action buffers (refilled in place before every pass), the captured step graph (VecGridWorld.capture_steps) and
    the contract's window."""

    def __init__(self, args, mode, workload, device, rank, world, node_barrier, N=None):
        import torch
        from gridworld_amd import VecGridWorld, _lib as L, workloads
        self.torch, self.L = torch, L
        self.args, self.mode, self.workload, self.device, self.rank, self.world = args, mode, workload, device, rank, world
        self.flying = mode == 'flying'
        self.N = N = int(N or args.envs_per_gpu)
        self.K, self.W = args.steps, args.warmup
        self.env_offset = rank * N  # rank-offset RNG streams / task seeds
        self.node_barrier = node_barrier
        # 'dummy' = BASELINE configs[1]: the DUMMY_TASK-equivalent target with gym.make's defaults (SizeReward on)
        self.env = env = VecGridWorld(N, device=device, action_space=mode, size_reward=(workload == 'dummy'), max_steps=MAX_STEPS,
                                      autoreset=True, lanes_per_env=args.lanes_per_env, debug_flags=args.debug_flags,
                                      env_index_base=self.env_offset)
        if workload == 'dummy':   # tasks/task_set.py:160: one blue block at dense [8, 10, 10], invariant=False
            dummy = torch.zeros((1, 9, 11, 11), dtype=torch.int8)
            dummy[0, 8, 10, 10] = 1
            env.set_tasks(dummy, invariant=False)
        elif workload == 'cdm':
            import numpy as np
            tg, st = workloads.cdm(N, args.seed + rank, np.load(CDM_GOALS)['dense'])
            env.set_tasks(tg.to(device), st.to(device))
        else:
            env.set_tasks(workloads.rt20(N, seed=args.seed + rank, device=device))
        env.reset()
        self.g = g = torch.Generator(device=device)
        g.manual_seed(args.seed + 7919 * rank + (1 if self.flying else 0))
        if not args.lockstep:
            # every env starts at a random step of its episode (GridWorld.step_no)
            env.set_step_no(torch.randint(0, MAX_STEPS, (N,), generator=g, device=device, dtype=torch.int32))
        self.pre_rolled = 0
        self.passes = 0
        W, K = self.W, self.K
        if self.flying:
            self.acts = dict(movement=torch.empty((W + K, N, 3), device=device), camera=torch.empty((W + K, N, 2), device=device),
                             inventory=torch.empty((W + K, N), device=device, dtype=torch.int32),
                             placement=torch.empty((W + K, N), device=device, dtype=torch.int32))
            # what env.step() is handed per step: the slicing is done once (a user loop would write acts[t] inline)
            self.act_t = [{k: v[t] for k, v in self.acts.items()} for t in range(W + K)]
        else:
            self.acts = torch.empty((W + K, N), dtype=torch.int32, device=device)
            self.act_t = list(self.acts.unbind(0))
        self.refill()
        if not args.lockstep:
            self.busy(0.0, MAX_STEPS)  # at least one episode length
        self.graph, self.head, self.timed_as = None, K, 'eager env.step() calls'
        self.chains = max(1, int(getattr(args, 'chains', 1)))
        if not args.no_graph and K > 8:
            self._capture()
        self.ev0, self.ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    # -- actions ----------------------------------------------------------------------------------------------
    def refill(self):
        """Fresh random actions into the SAME W + K device buffers (so a captured graph stays valid); untimed."""
        env = self.env
        self.passes += 1
        if self.flying:
            self.acts['movement'].uniform_(-1, 1, generator=self.g)     # movement ~ U(-1, 1)^3
            self.acts['camera'].uniform_(-5, 5, generator=self.g)       # camera ~ U(-5, 5)^2
            self.acts['inventory'].random_(0, 7, generator=self.g)
            self.acts['placement'].random_(0, 3, generator=self.g)
        else:
            t0 = self.passes * (self.W + self.K)
            self.L.check(env.lib.igw_fill_actions_walking(env.ctx, self.acts.data_ptr(), self.acts.shape[0], t0, self.args.seed,
                                                          self.env_offset, env._stream()), 'igw_fill_actions_walking')

    def busy(self, seconds, min_steps, fresh=True):
        """Untimed stepping with FRESH random actions: the pre-roll to the steady state / the clock ramp."""
        torch, env = self.torch, self.env
        t_ramp, n_pre = time.perf_counter(), 0
        while n_pre < min_steps or time.perf_counter() - t_ramp < seconds:
            if self.flying:   # (the first W + K actions of the buffers)
                if fresh:
                    self.refill()
                for t in range(min(self.W + self.K, 128)):
                    env.step(self.act_t[t])
                n_pre += min(self.W + self.K, 128)
            else:  # fused rollout with in-kernel random actions
                env.rollout(MAX_STEPS, seed=self.args.seed + 17 + self.pre_rolled, t0=self.pre_rolled, env_offset=self.env_offset)
                n_pre += MAX_STEPS
                self.pre_rolled += MAX_STEPS
            torch.cuda.synchronize(self.device)

    # -- the timed steps: a short eager head of env.step() calls + ONE replay of a captured step graph ------------
    def _capture(self):
        """VecGridWorld.capture_steps records the launches of steps W + head .. W + K - 1 without running them (the entry
        points never synchronise or allocate); instantiation happens here, before the warm-up and the clock.  The eager
        head starts the GPU within a few microseconds of the clock; the graph is launched while those kernels run, so
        its launch latency (10-16 us) is off the critical path.  If capture or instantiation fails the window is timed as
        eager env.step() calls, in this process."""
        torch, W, K = self.torch, self.W, self.K
        # eager launches while the host is still busy launching the graph (the first launch after a synchronize takes
        # ~15 us of host time, an event record ~7, a graph launch ~20; a kernel ~12): with a head of 2 the GPU ran out
        # of work for ~15 us before the graph arrived
        head = min(int(os.environ.get('IGW_BENCH_HEAD', 6)), K - 2)   # (IGW_BENCH_HEAD: experiments only)
        try:
            sub = {k: v[W + head:] for k, v in self.acts.items()} if self.flying else self.acts[W + head:]
            graph = self.env.capture_steps(sub, chains=self.chains)
            graph.replay()  # part of the setup: the first launch of a graph also uploads it
            torch.cuda.synchronize(self.device)
            self.graph, self.head = graph, head
            self.timed_as = f'{head} eager env.step() calls + one replay of VecGridWorld.capture_steps over the other {K - head}' + (
                f' (as {self.chains} independent chains of {self.N // self.chains}-env launches)' if self.chains > 1 else '')
        except Exception as e:  # noqa: BLE001 -- any capture / instantiate failure: eager launches instead
            try:
                torch.cuda.synchronize(self.device)
            except Exception:  # noqa: BLE001
                pass
            self.graph, self.head = None, K
            self.timed_as = 'eager env.step() calls (graph capture failed: %s)' % (str(e).splitlines()[0][:160] if str(e) else type(e).__name__)

    def window(self):
        """Refill (untimed), W untimed warm-up steps, then the clock around exactly K steps.  Returns (wall seconds,
        counters before the clock as a device tensor, kernel ms per launch from HIP events, host timeline)."""
        torch, env, W, K = self.torch, self.env, self.W, self.K
        ev0, ev1, act_t, step = self.ev0, self.ev1, self.act_t, self.env.step
        # Between two windows the host reads counters and refills buffers: the GPU idles for hundreds of microseconds
        # and its clocks drop; W = 5 warm-up steps (65 us) do not bring them back, and the first kernels of the window
        # would run 2-3 % slower than in a long run.  So: about 2 ms of untimed stepping with fresh random actions
        # first (setup, like the pre-roll), then the contract's W warm-up steps and the clock.
        if not self.args.lockstep:
            self.busy(0.002, 0, fresh=False)   # (the window refills the buffers right after)
        self.refill()
        # warm-up right before the clock; the counters are snapshotted on the device, not read, so nothing idles
        # the GPU between warm-up and clock
        for t in range(W):
            step(act_t[t])
        before = env.stats_tensor()
        ev0.record()   # torch creates the HIP events lazily at their first record(): not inside the clock
        ev1.record()
        ev1.query()
        self.node_barrier.wait()
        torch.cuda.synchronize(self.device)
        if K == 1:
            ev0.record()
        t_start = time.perf_counter()
        step(act_t[W])
        # The HIP-event window (kernel duration for the roofline) opens behind the first timed launch and spans the
        # other K - 1: an event recorded on the idle stream would be processed at once and the window would then
        # contain the launch latency of the first kernel, not only kernels.  The GPU is busy with that first launch
        # while the host records, so the wall clock does not see it.
        if K > 1:
            ev0.record()
        for t in range(W + 1, W + self.head):
            step(act_t[t])
        t_b = time.perf_counter()
        if self.graph is not None:
            self.graph.replay()
        t_c = time.perf_counter()
        ev1.record()
        t_d = time.perf_counter()
        spin_until(ev1.query)  # spin (with back-off): a blocking synchronize sleeps on an interrupt and wakes tens of us late
        t_e = time.perf_counter()
        torch.cuda.synchronize(self.device)
        t_f = time.perf_counter()
        self.node_barrier.wait()
        t_end = time.perf_counter()
        kernel_ms = ev0.elapsed_time(ev1) / max(K - 1, 1)  # the event window spans the last K - 1 timed launches
        return t_end - t_start, before, kernel_ms, (t_b - t_start, t_c - t_b, t_d - t_c, t_e - t_d, t_f - t_e, t_end - t_f)

    def api_loops(self, reps=5):
        """What a user of the public API gets, measured the way examples/run_env.py:18-26 loops: a plain
        `for t in range(K): obs, reward, done, info = env.step(actions[t])` (the slice included) between two
        synchronizes -- eager, for K = 20 and K = the run's K -- and the same K steps as ONE replay of
        env.capture_steps(actions[:K]).  Median of `reps` passes each (actions refilled before every pass; 2 ms of
        untimed stepping first, as before every window)."""
        torch, env, N = self.torch, self.env, self.N
        acts = self.acts
        out = {'eager': {}, 'graph': {}}

        def one(K, fn):
            ts = []
            for _ in range(reps):
                if not self.args.lockstep:
                    self.busy(0.002, 0, fresh=False)
                self.refill()
                for t in range(self.W):
                    env.step(self.act_t[t])
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                fn(K)
                torch.cuda.synchronize(self.device)
                ts.append(time.perf_counter() - t0)
            return N * K / statistics.median(ts)

        def eager(K):
            if self.flying:
                for t in range(K):
                    env.step({'movement': acts['movement'][t], 'camera': acts['camera'][t],
                              'inventory': acts['inventory'][t], 'placement': acts['placement'][t]})
            else:
                for t in range(K):
                    env.step(acts[t])
        for K in sorted({min(20, self.K), self.K}):
            out['eager'][str(K)] = one(K, eager)
            try:
                g = env.capture_steps({k: v[:K] for k, v in acts.items()} if self.flying else acts[:K])
                g.replay()
                torch.cuda.synchronize(self.device)
                out['graph'][str(K)] = one(K, lambda K: g.replay())
                del g
            except Exception as e:  # noqa: BLE001
                out['graph'][str(K)] = None
                out['graph_error'] = str(e).splitlines()[0][:160] if str(e) else type(e).__name__
        return out

    def measure(self, windows, rehearsals):
        """`rehearsals` passes that are reported but not counted, then `windows` measured ones.  The first passes
        through the host code paths (Python bytecode, ctypes thunks, the HIP runtime's launch and graph-launch paths)
        cost 30-80 us more than later ones -- 10-20 % of a 20-step window and nothing to do with the step kernel.
        Right before them 0.3 s of untimed stepping: task upload, graph capture and instantiation leave the GPU idle
        for tens of milliseconds and its clocks drop."""
        from gridworld_amd import dist as gdist
        L, env, N, K = self.L, self.env, self.N, self.K
        if not self.args.lockstep:
            self.busy(0.3, 0)
        rehearsal_ms = [round(1e3 * self.window()[0] / K, 5) for _ in range(rehearsals)]
        walls, kernels, ps, resets, cells, counted, timelines = [], [], [], [], [], [], []
        import gc
        for _ in range(max(1, windows)):
            # (a cyclic-GC pass of the interpreter inside a 250-us window is a 20-50 % outlier: collect between windows,
            # not in them)
            gc.collect()
            gc.disable()
            try:
                el, st0_dev, kms, host_tl = self.window()
            finally:
                gc.enable()
            timelines.append(host_tl)
            s0, s1 = st0_dev.cpu(), env.stats_tensor().cpu()
            walls.append(el)
            kernels.append(kms)
            ps.append(float(s1[L.STAT_CHANGED] - s0[L.STAT_CHANGED]) / (N * K))
            cells.append(float(s1[L.STAT_RESCANS] - s0[L.STAT_RESCANS]) / (N * K))
            resets.append(int(s1[L.STAT_RESETS] - s0[L.STAT_RESETS]))
            counted.append(int(s1[L.STAT_STEPS] - s0[L.STAT_STEPS]))   # env-steps the kernels counted on the device
            if os.environ.get('IGW_BENCH_TRACE'):
                print('host us: head launches %.1f | graph launch %.1f | ev1.record %.1f | spin %.1f | synchronize %.1f | '
                      'barrier %.1f | total %.1f | kernel %.2f us' % (tuple(1e6 * x for x in host_tl) + (1e6 * el, 1e3 * kms)),
                      file=sys.stderr)
        # per window the MAX over ranks (one collective for all windows), then the median window
        walls_max = gdist.reduce_windows(walls, self.device)
        med = statistics.median(walls_max)
        i_med = min(range(len(walls_max)), key=lambda i: abs(walls_max[i] - med))
        srt = sorted(walls_max)
        pct = lambda q: srt[min(len(srt) - 1, max(0, int(round(q * (len(srt) - 1)))))]   # noqa: E731
        i_max = max(range(len(walls_max)), key=lambda i: walls_max[i])
        names = ('head_launches', 'graph_launch', 'event_record', 'spin', 'synchronize', 'barrier')
        return {'elapsed': med, 'windows_ms_per_step': [round(1e3 * w / K, 5) for w in walls_max],
                'windows_kernel_us': [round(1e3 * k, 3) for k in kernels],
                'window_spread': (max(walls_max) - min(walls_max)) / med,
                'window_p10_ms_per_step': 1e3 * pct(0.1) / K, 'window_p90_ms_per_step': 1e3 * pct(0.9) / K,
                'window_spread_p10_p90': (pct(0.9) - pct(0.1)) / med,
                # where the host spent the slowest and the median window of THIS rank (us): an outlier window shows
                # up either in its kernels (windows_kernel_us) or in one of these host segments
                'slowest_window': {'index': i_max, 'ms_per_step': round(1e3 * walls_max[i_max] / K, 5),
                                   'kernel_us': round(1e3 * kernels[i_max], 3),
                                   'host_us': {n: round(1e6 * x, 1) for n, x in zip(names, timelines[i_max])}},
                'median_window': {'index': i_med, 'kernel_us': round(1e3 * kernels[i_med], 3),
                                  'host_us': {n: round(1e6 * x, 1) for n, x in zip(names, timelines[i_med])}},
                'rehearsal_ms_per_step': rehearsal_ms,
                'kernel_ms': statistics.median(kernels), 'p_changed': ps[i_med], 'p_cell_changed': cells[i_med],
                'resets_in_window': resets[i_med], 'steps_counted': counted[i_med], 'steps_counted_all': counted}


HBM_ACHIEVABLE_GBS = 6300.0   # what a streaming kernel reaches on this part (MI355X_MICROARCH.md)


def roofline_of(r, m, lanes, has_start_frac=0.0, profiles_dir=None, kind=None):
    """The roofline object of a measured workload.

    `achieved` / `frac` are PHYSICAL: the HBM bytes one launch of the dominant kernel moves -- FETCH_SIZE / WRITE_SIZE
    of the committed rocprofv3 PMC profile, corrected as MI355X_MICROARCH.md prescribes (tools/summarize_profile.py),
    scaled to this run's env count -- over the kernel's average duration measured in THIS run with HIP events on the
    launch stream, against the 8 TB/s peak.  The profile must be of the same kernel build: it carries the library's
    build id (hash of csrc/ + include/igw.h + flags), and `stale` is true when that differs from igw_build_id() of the
    library being timed.  Without any profile the design's byte count stands in (`basis` says so).

    Beside it: `frac_of_achievable` (against the ~6.3 TB/s a streaming kernel reaches), `design_bytes_per_env_step`
    (what this data layout has to move, counted below) and `wasted_traffic` = measured / design bytes, and the
    SURVEY 8(d) convention figure (`convention`: the int8 grid priced as if streamed every step -- this design reads
    a 192-byte bitmap of it instead, so that figure is not a physical fraction and can exceed 1)."""
    from gridworld_amd import _lib as L
    N, flying = r.N, r.flying
    kernel_ms, p, pc = m['kernel_ms'], m['p_changed'], m['p_cell_changed']
    kernel_s = kernel_ms * 1e-3
    resets_per_step = m['resets_in_window'] / float(N * r.K)
    # SURVEY 8(d) convention
    bytes_per_step = (BYTES_BASE + (24 if flying else 0)) + BYTES_CHANGED * p  # flying actions are 28 B, not 4
    conv_gbs = N * bytes_per_step / kernel_s / 1e9
    # What this design has to move per env-step: occupancy bitmap 192 r, agent record 64 r +
    # 64 w, aux record 16 r, action 4 (28 flying), output record 64 w; per changed cell: histogram row 1024 r + 1024 w,
    # colour-index block 160 r, start byte + break colour (one 32-byte sector each) 64 r, grid byte + bitmap word
    # (one sector each) 64 w, aux record 16 w; per auto-reset: metadata 128 r, grid row 1104 w, bitmap 192 w,
    # histogram row 1024 w, aux 16 w (+ starting row 1104 r and its bitmap 192 r with a starting grid).
    base = 192 + 64 + 64 + 16 + (28 if flying else 4) + 64
    per_cell = 1024 + 1024 + 160 + 64 + 64 + 16
    per_reset = 128 + 1104 + 192 + 1024 + 16 + has_start_frac * (1104 + 192)
    design = base + pc * per_cell + resets_per_step * per_reset
    lib_id = L.build_id()
    if kind is None:
        kind = 'flying' if flying else ('cdm' if r.workload == 'cdm' else '')
    traffic = load_profile('traffic.json', kind, lib_id, profiles_dir)
    issue = load_profile('issue.json', kind, lib_id, profiles_dir)
    hbm_bytes = None if traffic is None else traffic.get('hbm_bytes_per_launch')
    prof_envs = None if traffic is None else (traffic.get('envs') or ENVS_PER_GPU)
    if hbm_bytes is not None and prof_envs and prof_envs != N:
        hbm_bytes = hbm_bytes * N / prof_envs   # the committed profile is of the 65,536-env launch
    basis = 'pmc' if hbm_bytes is not None else 'design'
    moved = hbm_bytes if hbm_bytes is not None else N * design
    achieved = moved / kernel_s / 1e9
    stale = None if traffic is None else bool(traffic['stale'])
    roof = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': hbm_bytes,
            'basis': {'pmc': 'HBM bytes per launch from the committed rocprofv3 PMC profile (FETCH_SIZE x 2 + WRITE_SIZE, KiB) '
                             '/ kernel_avg_ms of this run', 'design': 'no PMC profile found: the design byte count / kernel_avg_ms'}[basis],
            'stale': stale,
            'frac_of_achievable': achieved / HBM_ACHIEVABLE_GBS, 'achievable_gbs': HBM_ACHIEVABLE_GBS,
            # <lanes per env, action space, extras, flying's whole-blocks variant> (csrc/igw_kernels.hip)
            'kernel': 'igw::step_kernel<%d, %d, false, %s>' % (lanes, 1 if flying else 0,
                                                              'true' if flying and lanes == 4 and N % 64 == 0 else 'false'),
            'kernel_avg_ms': kernel_ms,
            'design_bytes_per_env_step': design, 'design_bytes_per_launch': N * design,
            'frac_design': N * design / kernel_s / 1e9 / HBM_PEAK_GBS,
            'wasted_traffic': None if hbm_bytes is None else hbm_bytes / (N * design),
            'profile': None if traffic is None else {
                'source': traffic.get('_file'), 'build_id': traffic.get('build_id'), 'git_commit': traffic.get('git_commit'),
                'library_build_id': lib_id, 'stale': stale, 'envs': prof_envs,
                'kernel_avg_ns_under_rocprof': traffic.get('kernel_avg_ns'),
                # reproducible from profiles/ alone: the profile's bytes over the profile's own kernel time
                'frac_in_profile_run': (None if not traffic.get('kernel_avg_ns') or traffic.get('hbm_bytes_per_launch') is None else
                                        traffic['hbm_bytes_per_launch'] / (traffic['kernel_avg_ns'] * 1e-9) / 1e9 / HBM_PEAK_GBS)},
            'convention': {'what': 'SURVEY.md 8(d): 1274 + 1106 p bytes per env-step (1298 flying), the int8 grid priced as streamed '
                                   'every step; this design keeps a 192-byte occupancy bitmap of it, so the figure is not a physical '
                                   'fraction (it exceeds 1 at large launches) -- kept for comparison across rounds',
                           'algorithmic_bytes_per_env_step': bytes_per_step, 'algorithmic_bytes_per_launch': N * bytes_per_step,
                           'achieved': conv_gbs, 'frac': conv_gbs / HBM_PEAK_GBS},
            'frac_convention': conv_gbs / HBM_PEAK_GBS,
            'limiter': 'issue',   # see `issue`: the kernel is bound by instruction issue of the waves sharing a SIMD, not by HBM
            'measured_in_this_run': ['kernel_avg_ms', 'design_bytes_per_env_step (p_changed, resets counted on the device)',
                                     'convention'],
            'from_profile': ['traffic', 'issue.*_per_wave']}
    iss = None
    if issue is not None:
        valu = issue.get('valu_insts_per_wave')
        waves_per_simd = (N * lanes / 64.0) / 1024.0   # 256 CUs x 4 SIMDs
        iss = {'bound': 'issue', 'source': issue.get('_file'), 'build_id': issue.get('build_id'), 'stale': bool(issue['stale']),
               'from_profile': {k: issue.get(k) for k in ('valu_insts_per_wave', 'salu_insts_per_wave', 'lds_insts_per_wave',
                                                          'vmem_rd_insts_per_wave', 'vmem_wr_insts_per_wave',
                                                          'frac_wave_time_parked_on_waitcnt', 'frac_wave_time_issue_stalled',
                                                          'frac_wave_time_issuing', 'kernel_avg_ns', 'dispatches_averaged')},
               'waves_per_simd': waves_per_simd,
               # one VALU instruction occupies its SIMD for 4 cycles: share of THIS run's kernel time the SIMDs spend
               # issuing VALU instructions (instruction count of the profile, kernel time of this run)
               'achieved': None if not valu else valu * waves_per_simd * 4.0 / (kernel_s * GPU_CLOCK_GHZ * 1e9),
               'peak': 1.0, 'unit': 'VALU issue cycles per SIMD cycle',
               'valu_issue_util_per_simd': None if not valu else valu * waves_per_simd * 4.0 / (kernel_s * GPU_CLOCK_GHZ * 1e9),
               'valu_insts_per_env_step': None if not valu else valu * lanes / 64.0}
    util = None if iss is None else iss['valu_issue_util_per_simd']
    mall = N * (192 + 64 + 64 + 16 + 64) <= 256e6   # the per-step records + bitmaps of the batch fit the 256 MB Infinity Cache
    roof['note'] = ('FETCH_SIZE / WRITE_SIZE count requests that leave the XCD L2s for the fabric; Infinity-Cache (MALL) hits are '
                    'included (MI355X_MICROARCH.md).  At %s envs the per-step working set (%.0f MB of bitmaps and records + the '
                    'touched histogram rows) %s the 256 MB Infinity Cache, so this is %s.'
                    % (f'{N:,}', N * (192 + 64 + 64 + 16 + 64) / 1e6, 'fits' if mall else 'exceeds',
                       'fabric traffic served mostly by the Infinity Cache, not DRAM streaming: see config.large for the batch size '
                       'at which 8 TB/s is the ceiling' if mall else 'DRAM traffic: the regime in which the 8 TB/s of HBM3E is the ceiling'))
    if util is None:   # no SQ profile of this build: the HBM object stands alone (as in rounds 1-5)
        roof['issue'] = iss
        roof['issue_util'] = None
        return roof, iss
    # The top-level object names what bounds the kernel: VALU instruction issue.  `achieved` = the share
    # of the launch in which a SIMD's vector ALU is issuing (VALU instructions per wavefront of the committed SQ profile x
    # wavefronts per SIMD x 4 cycles, over THIS run's kernel time); `traffic` = the PMC bytes per launch as the contract asks;
    # the HBM-side fraction stays beside it as `hbm`.
    top = {'bound': 'issue', 'achieved': util, 'peak': 1.0, 'unit': 'VALU issue cycles per SIMD cycle', 'frac': util,
           'traffic': roof['traffic'], 'kernel': roof['kernel'], 'kernel_avg_ms': kernel_ms,
           'stale': bool(iss['stale']) or bool(stale),
           'what': 'the step kernel is bound by vector-ALU instruction issue of the four wavefronts that share a SIMD and by the '
                   'launch structure around it (launch ramp, input burst, the wait for the slowest wavefront), not by HBM: '
                   'frac = VALU issue utilisation; hbm = the PMC traffic fraction (rounds 1-5 had this at the top level)',
           'valu_insts_per_wave': iss['from_profile']['valu_insts_per_wave'], 'waves_per_simd': iss['waves_per_simd'],
           'hbm': roof, 'issue': iss, 'issue_util': util, 'limiter': 'issue',
           'measured_in_this_run': ['kernel_avg_ms', 'hbm.design_bytes_per_env_step (p_changed, resets counted on the device)'],
           'from_profile': ['traffic', 'valu_insts_per_wave', 'hbm.traffic', 'issue.from_profile.*']}
    return top, iss


HBM_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'basis', 'stale', 'kernel', 'kernel_avg_ms',
            'design_bytes_per_env_step', 'wasted_traffic', 'frac_of_achievable', 'frac_convention', 'profile', 'limiter', 'note')


def slim_roofline(roof):
    """The roofline object of a secondary workload: the top level (what bounds the kernel) and the HBM side without the
    long explanatory fields."""
    if 'hbm' not in roof:
        return {k: roof[k] for k in HBM_KEYS if k in roof}
    out = {k: roof[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_avg_ms', 'stale',
                                'valu_insts_per_wave', 'waves_per_simd', 'limiter')}
    out['hbm'] = {k: roof['hbm'][k] for k in HBM_KEYS if k in roof['hbm']}
    return out


def facade_rate(seed, steps=3000):
    """The latency end of the API -- BASELINE configs[0] through the reference's own calling sequence on the HIP path:
    gridworld_amd.make('IGLUGridworldVector-v0') (create_env defaults: Discrete(18), SizeReward, select_and_place), the
    DUMMY_TASK-equivalent task, `obs, reward, done, info = env.step(int)` with numpy observations, reset on done.
    Every step is one kernel launch + one stream synchronisation (records, grid and action in pinned host memory).
    Timed as examples/run_env.py:18-26 does (step() only) and including the resets."""
    import numpy as np
    import gridworld_amd as G
    env = G.make('IGLUGridworldVector-v0')
    dummy = np.zeros((9, 11, 11), np.int32)
    dummy[8, 10, 10] = 1
    env.set_task(G.Task('', dummy, starting_grid=[], invariant=False))
    env.reset()
    acts = [int(a) for a in np.random.RandomState(seed).randint(18, size=steps + 300)]
    for a in acts[:300]:   # warm: graph capture, code paths
        if env.step(a)[2]:
            env.reset()
    t_step, n_reset = 0.0, 0
    t0 = time.perf_counter()
    for a in acts[300:]:
        ts = time.perf_counter()
        obs, reward, done, info = env.step(a)
        t_step += time.perf_counter() - ts
        if done:
            env.reset()
            n_reset += 1
    wall = time.perf_counter() - t0
    return {'workload': 'configs[0]: 1 env, walking Discrete(18), DUMMY_TASK-equivalent task, gym.make defaults, numpy '
                        'observations, reset on done',
            'steps': steps, 'resets': n_reset, 'steps_per_s_step_only': steps / t_step, 'steps_per_s_incl_resets': steps / wall,
            'us_per_step': 1e6 * t_step / steps,
            'how': 'one launch of step_kernel<32,...> + one stream synchronisation per step; the env\'s records, grid and '
                   'action live in pinned host memory the kernel reads and writes across PCIe (no copies)',
            'reference_python_here': 'BASELINE.md: 3.8-4.2 k steps/s step only, 2.9-3.2 k incl. resets (build container, 1 core)'}


def main():
    args = parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ:
        raise SystemExit(self_launch(args))
    if args.dry_run:
        return dry_run(args)
    from gridworld_amd import dist as gdist
    try:
        run(args)
    except BaseException:
        # A rank that failed must not wait for peers that are still inside a collective or the shared-memory barrier:
        # no closing barrier, tear our side down and leave non-zero at once -- torchrun then ends the other ranks.
        gdist.shutdown(barrier=False)
        raise
    gdist.shutdown()   # success: closing barrier (gloo) + destroy_process_group, so no rank falls off the end early


def run(args):

    import torch
    from gridworld_amd import _lib as L, dist as gdist
    rank, local_rank, world = gdist.init()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # one process per GPU; IGW_SHARE_GPU=1 lets several ranks share one device (CI on a 1-GPU box)
    device = torch.device('cuda', 0 if os.environ.get('IGW_SHARE_GPU') else local_rank)
    torch.cuda.set_device(device)
    N, K, W = args.envs_per_gpu, args.steps, args.warmup
    flying = args.mode == 'flying'
    # the barrier of the timing bracket: ranks of one node meet in shared memory (microseconds; an RCCL barrier
    # costs 50-100 us, a quarter of a 20-step window)
    node_barrier = gdist.NodeBarrier()

    torch.set_num_threads(max(1, usable_cpus() // world))   # host-side torch work of a rank stays within its share of the quota
    r = Runner(args, args.mode, args.workload, device, rank, world, node_barrier)
    env = r.env
    m = r.measure(args.windows, args.rehearsals)
    max_elapsed = m['elapsed']
    # The one RCCL collective of the run (north star): every rank's DEVICE-SIDE step count of the median window --
    # the delta of IGW_STAT_STEPS, which every step launch adds its env count to (one atomic per launch) -- gathered over RCCL / xGMI; `value`
    # is computed from their sum, and the sum must be what the launches were asked to do.
    steps_per_rank, gather_via = gdist.gather_counts_rccl(m['steps_counted'], device)
    total_steps = sum(steps_per_rank)
    if args.debug_flags or os.environ.get('IGW_AB_NO_STEP_COUNTER'):   # (diagnostic build, timing-only ablations: some switches end the
        total_steps = N * K * world                                     # kernel before it counts; tools/ab_variants.sh: the -DIGW_STEPS_MODE=0 variant)
    elif any(c != N * K for c in m['steps_counted_all']) or total_steps != N * K * world:
        raise SystemExit(f'device step counters disagree with the launches: windows {m["steps_counted_all"]}, '
                         f'per rank {steps_per_rank}, expected {N * K} per rank and window')
    ranks_seen = gdist.gather_counts(rank)   # (control plane)
    n_ranks = len(ranks_seen)
    kernel_us_per_rank = [round(x, 3) for x in gdist.gather_floats(1e3 * m['kernel_ms'])]   # (control plane)
    api = r.api_loops() if not args.no_api else None
    # the same window with the captured steps as P independent chains of sub-batch launches (parallel graph branches)
    chain_sweep = {}
    if world == 1 and args.chain_sweep and not args.no_graph and K > 8:
        for P in [int(x) for x in args.chain_sweep.split(',') if x.strip()]:
            if P == r.chains or N % P:
                continue
            keep = (r.graph, r.head, r.timed_as, r.chains)
            r.chains = P
            r._capture()
            if r.graph is not None and r.graph.chains == P:
                mP = r.measure(max(3, args.windows // 2), 1)
                chain_sweep[str(P)] = {'value': N * K / mP['elapsed'], 'ms_per_step': 1e3 * mP['elapsed'] / K,
                                       'event_span_us_per_step': 1e3 * mP['kernel_ms'], 'windows_ms_per_step': mP['windows_ms_per_step']}
            else:
                chain_sweep[str(P)] = {'error': r.timed_as}
            r.graph, r.head, r.timed_as, r.chains = keep
    chains_used = r.chains

    # secondary: fused T-step rollout (state resident in LDS/registers, in-kernel RNG)
    fused = fused_rec = None
    if not args.no_fused and not flying:
        Tf = 250
        env.rollout(Tf, seed=args.seed + 1, t0=0, env_offset=r.env_offset)  # warm
        torch.cuda.synchronize(device)
        gdist.barrier()
        t0 = time.perf_counter()
        env.rollout(Tf, seed=args.seed + 2, t0=0, env_offset=r.env_offset)
        torch.cuda.synchronize(device)
        gdist.barrier()
        f_steps, f_el = gdist.reduce_window(N * Tf, time.perf_counter() - t0)
        fused = f_steps / f_el
        # the same fused loop over RECORDED actions (igw_rollout_walking_actions): the first 256 of the timed actions
        rec_acts = r.acts[:min(256, r.acts.shape[0])]
        Tr = rec_acts.shape[0]
        env.rollout_actions(rec_acts)  # warm
        torch.cuda.synchronize(device)
        gdist.barrier()
        t0 = time.perf_counter()
        env.rollout_actions(rec_acts)
        torch.cuda.synchronize(device)
        gdist.barrier()
        r_steps, r_el = gdist.reduce_window(N * Tr, time.perf_counter() - t0)
        fused_rec = r_steps / r_el

    if flying and not args.no_fused:   # the fused loop over the recorded flying actions (igw_rollout_flying_actions)
        rec = r.acts
        Tr = r.acts['movement'].shape[0]
        env.rollout_actions(rec)  # warm
        torch.cuda.synchronize(device)
        gdist.barrier()
        t0 = time.perf_counter()
        env.rollout_actions(rec)
        torch.cuda.synchronize(device)
        gdist.barrier()
        r_steps, r_el = gdist.reduce_window(N * Tr, time.perf_counter() - t0)
        fused_rec = r_steps / r_el

    # secondary: the same kernel as two independent sub-batches on two HIP streams (VecGridWorld.split, the
    # EnvPool-style asynchronous mode): no barrier between the halves, so the start of one half's next step
    # fills the end of the other's -- what one launch per step over the whole batch cannot do
    async2 = None
    if args.do_async and not flying and N % 2 == 0:
        subs = env.split(2)
        jobs = [(sb.ctx, ctypes.c_void_p(sb.stream.cuda_stream), 4 * sb.lo) for sb in subs]
        Ka = min(K, W + K)
        act_ptrs = [a.data_ptr() for a in r.act_t]
        walk_fn = env.lib.igw_step_walking
        torch.cuda.synchronize(device)
        for timed in (False, True):
            gdist.barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for t in range(Ka):
                for ctx_s, st_s, off in jobs:
                    rc = walk_fn(ctx_s, act_ptrs[t] + off, st_s)
                    if rc:
                        L.check(rc, 'igw_step_walking')
            torch.cuda.synchronize(device)
            gdist.barrier()
            a_el = time.perf_counter() - t0
        a_steps, a_el = gdist.reduce_window(N * Ka, a_el)
        async2 = a_steps / a_el
        del subs

    lanes = env.cfg.lanes_per_env or auto_lanes(N)
    roof, iss = roofline_of(r, m, lanes, has_start_frac=0.96 if args.workload == 'cdm' else 0.0)
    timed_as = r.timed_as

    # secondaries of the default single-GPU run: BASELINE configs[3] (flying) and the real IGLU targets with partial
    # starting grids, through the same window machinery (fewer windows)
    secondary = {}
    if world == 1 and not args.no_secondary and not flying and args.workload == 'rt20' and not args.debug_flags:
        del r, env
        torch.cuda.empty_cache()
        for key, mode, wl in (('flying', 'flying', 'rt20'), ('cdm', 'walking', 'cdm')):
            r2 = Runner(args, mode, wl, device, rank, world, node_barrier)
            m2 = r2.measure(max(3, args.windows // 2), 2)
            roof2, iss2 = roofline_of(r2, m2, lanes, has_start_frac=0.96 if wl == 'cdm' else 0.0)
            secondary[key] = {
                'workload': ('configs[3]: 65,536 parallel envs, flying action space (continuous movement / camera Box), '
                             'random 20-block targets (rt20), uniform random actions, auto-reset at done (max_steps=250)')
                if key == 'flying' else
                ('65,536 parallel envs, walking Discrete(18), the 156 IGLU CDM target structures tiled over the batch, '
                 'each with a random partial starting grid (96 % of the envs; at least one block left to build; a third of the envs '
                 'with blocks that are not in the target: negative synthetic ids), uniform random actions, auto-reset at '
                 'done (max_steps=250)'),
                'value': N * K / m2['elapsed'], 'unit': 'env-steps/s', 'ms_per_step': 1e3 * m2['elapsed'] / K,
                'windows_ms_per_step': m2['windows_ms_per_step'], 'kernel_us': 1e3 * m2['kernel_ms'],
                'p_changed': m2['p_changed'], 'p_cell_changed': m2['p_cell_changed'],
                'resets_in_window': m2['resets_in_window'], 'timed_as': r2.timed_as,
                'roofline': slim_roofline(roof2),
                'valu_issue_util_per_simd': None if iss2 is None else iss2['valu_issue_util_per_simd']}
            del r2
            torch.cuda.empty_cache()

    # the latency end: BASELINE configs[1] (4,096 envs, DUMMY_TASK-equivalent, gym.make defaults) through the same
    # window machinery, and configs[0] through the 1-env gym facade
    small = facade = None
    if world == 1 and not args.no_secondary and not flying and args.workload == 'rt20' and not args.debug_flags:
        Ns = 4096
        r3 = Runner(args, 'walking', 'dummy', device, rank, world, node_barrier, N=Ns)
        m3 = r3.measure(max(3, args.windows // 2), 2)
        lanes3 = r3.env.cfg.lanes_per_env or auto_lanes(Ns)
        waves3 = Ns * lanes3 // 64
        small = {'workload': 'configs[1]: 4,096 parallel envs, walking Discrete(18), DUMMY_TASK-equivalent task (one blue block at '
                             'dense [8,10,10], invariant=False), gym.make defaults (SizeReward, select_and_place), uniform random '
                             'actions, auto-reset at done (max_steps=250)',
                 'value': Ns * K / m3['elapsed'], 'unit': 'env-steps/s', 'ms_per_step': 1e3 * m3['elapsed'] / K,
                 'windows_ms_per_step': m3['windows_ms_per_step'], 'kernel_us': 1e3 * m3['kernel_ms'],
                 'kernel': 'igw::step_kernel<%d, 0, false, false>' % lanes3, 'lanes_per_env': lanes3, 'wavefronts': waves3,
                 'p_changed': m3['p_changed'], 'resets_in_window': m3['resets_in_window'], 'timed_as': r3.timed_as,
                 'bound': 'latency: %d wavefronts for 1,024 SIMDs run as ONE round, so a launch lasts the launch ramp + the '
                          'dependent instruction chain of one wavefront (wider groups shorten the chain: %d lanes per env); '
                          'env-steps/s per env is highest here, per GPU lowest' % (waves3, lanes3)}
        del r3
        torch.cuda.empty_cache()
        facade = facade_rate(args.seed)

    # config.sweep: the same workload and window at larger batches per GPU -- envs_per_gpu is a tunable of the API, and the
    # launch structure (launch ramp, input burst, the wait for the slowest wavefront: about 30 % of a 65,536-env launch)
    # amortises with it; the headline stays BASELINE's 65,536.  The largest size (>= 2,097,152 envs: 0.8 GB of bitmaps and
    # records per step, beyond the 256 MB Infinity Cache) is also config.large, with its own PMC profile
    # (profiles/r*_large_traffic.json): the one regime in which the HBM roofline is the roofline.
    sweep = large = None
    sizes = []
    if world == 1 and not args.no_secondary and not flying and args.workload == 'rt20' and not args.debug_flags and args.sweep:
        sizes = sorted({int(x) for x in args.sweep.split(',') if x.strip()} - {N})
    if sizes:
        sweep = {str(N): {'value': N * K / max_elapsed, 'ms_per_step': 1e3 * max_elapsed / K, 'kernel_us': 1e3 * m['kernel_ms'],
                          'kernel_us_per_65536_env_steps': 1e3 * m['kernel_ms'] * 65536.0 / N, 'lanes_per_env': lanes, 'headline': True}}
        for Ns in sizes:
            try:
                r4 = Runner(args, 'walking', 'rt20', device, rank, world, node_barrier, N=Ns)
                m4 = r4.measure(3, 1)
                lanes4 = r4.env.cfg.lanes_per_env or auto_lanes(Ns)
                sweep[str(Ns)] = {'value': Ns * K / m4['elapsed'], 'ms_per_step': 1e3 * m4['elapsed'] / K, 'kernel_us': 1e3 * m4['kernel_ms'],
                                  'kernel_us_per_65536_env_steps': 1e3 * m4['kernel_ms'] * 65536.0 / Ns, 'lanes_per_env': lanes4,
                                  'windows_ms_per_step': m4['windows_ms_per_step'], 'p_changed': m4['p_changed']}
                if Ns == sizes[-1] and Ns >= 2097152:
                    roof4, iss4 = roofline_of(r4, m4, lanes4, kind='large')
                    large = {'workload': 'configs[2] at %s envs on ONE GPU: walking Discrete(18), rt20 targets (one private task row per '
                                         'env), uniform random actions, auto-reset at done; per step %.2f GB of bitmaps, records and '
                                         'histogram rows -- beyond the 256 MB Infinity Cache' % (f'{Ns:,}', Ns * 614e-9),
                             'envs': Ns, 'value': Ns * K / m4['elapsed'], 'unit': 'env-steps/s', 'ms_per_step': 1e3 * m4['elapsed'] / K,
                             'kernel_us': 1e3 * m4['kernel_ms'], 'kernel_us_per_65536_env_steps': 1e3 * m4['kernel_ms'] * 65536.0 / Ns,
                             'p_changed': m4['p_changed'], 'resets_in_window': m4['resets_in_window'], 'timed_as': r4.timed_as,
                             'device_memory_gb': torch.cuda.max_memory_allocated(device) / 1e9,
                             'roofline': slim_roofline(roof4)}
                del r4
            except Exception as e:  # noqa: BLE001 -- a size that does not fit (another process on the GPU) must not lose the headline
                sweep[str(Ns)] = {'error': (str(e).splitlines() or [type(e).__name__])[0][:200]}
            torch.cuda.empty_cache()

    if rank != 0:
        return None
    wl_text = {('walking', 'rt20'): 'configs[2]: 65,536 parallel envs per GPU, walking Discrete(18), random 20-block '
                                    'targets (rt20), full maximal_intersection reward, uniform random actions, '
                                    'auto-reset at done (max_steps=250)',
               ('flying', 'rt20'): 'configs[3]: 65,536 parallel envs per GPU, flying action space (continuous movement / '
                                   'camera Box), random 20-block targets (rt20), uniform random actions, auto-reset at '
                                   'done (max_steps=250)',
               ('walking', 'cdm'): 'IGLU CDM target structures with random partial starting grids, walking Discrete(18), '
                                   'uniform random actions, auto-reset at done (max_steps=250)',
               ('flying', 'cdm'): 'IGLU CDM target structures with random partial starting grids, flying action space, '
                                  'uniform random actions, auto-reset at done (max_steps=250)'}[(args.mode, args.workload)]
    if N != ENVS_PER_GPU:
        wl_text = wl_text.replace('65,536', f'{N:,}')
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
        'config': {'workload': wl_text,
                   'envs_per_gpu': N, 'total_envs': N * n_ranks, 'lanes_per_env': lanes,
                   'launches_per_step': 1,
                   'timed_as': timed_as,
                   'value_is': 'median of %d complete windows (W warm-up steps, barrier + synchronize, K timed steps, '
                               'synchronize + barrier; max over ranks per window)' % len(m['windows_ms_per_step']),
                   'windows_ms_per_step': m['windows_ms_per_step'], 'window_spread': m['window_spread'],
                   'window_p10_ms_per_step': m['window_p10_ms_per_step'], 'window_p90_ms_per_step': m['window_p90_ms_per_step'],
                   'window_spread_p10_p90': m['window_spread_p10_p90'],
                   'slowest_window': m['slowest_window'], 'median_window': m['median_window'],
                   'windows_kernel_us': m['windows_kernel_us'],
                   'episodes': 'lock-step' if args.lockstep else 'de-synchronised (random episode phase + pre-roll)',
                   'setup': 'untimed: task upload, pre-roll of >= 250 steps with fresh random actions (steady state), graph '
                            'capture + one replay, 0.3 s more of such stepping (clock ramp), %d rehearsal passes (host code '
                            'paths warm; reported, not counted); before every pass 2 ms of such stepping (the clocks drop while the host '
                            'reads counters between passes) and a refill of the W + K action buffers on the device with fresh random '
                            'actions' % args.rehearsals,
                   'rehearsal_ms_per_step': m['rehearsal_ms_per_step'],
                   'resets_in_window': m['resets_in_window'], 'p_changed': m['p_changed'],
                   'p_cell_changed': m['p_cell_changed'],
                   'step_count_gather': gather_via, 'device_step_counts_per_rank': steps_per_rank,
                   'ranks_seen': ranks_seen, 'kernel_us_per_rank': kernel_us_per_rank,
                   'host': {'usable_cpus': usable_cpus(), 'torch_threads': torch.get_num_threads()},
                   'fused_rollout_env_steps_per_s': fused,
                   'fused_rollout_recorded_actions_env_steps_per_s': fused_rec,
                   'chains': chains_used,
                   'overlap_note': 'concurrent sub-batch launches (two streams, parallel graph branches, one graph per stream) are '
                                   'all slower than one whole-batch launch per step on this part: profiles/r05_chains.txt'},
        'roofline': roof,
    }
    if api is not None:
        head = out['value'] / n_ranks   # this rank's share: the API loops are per rank, not reduced
        out['config']['api_step_env_steps_per_s'] = api['eager']
        out['config']['api_graph_env_steps_per_s'] = api['graph']
        out['config']['api_note'] = ('rank 0, walking: `for t in range(K): env.step(actions[t])` through VecGridWorld.step '
                                     '(eager, slice included) / ONE replay of VecGridWorld.capture_steps(actions[:K]); '
                                     'keys = K; between two synchronizes, median of 5 passes; `value` itself is timed through '
                                     'the same public calls (config.timed_as)')
        out['config']['api_vs_value'] = {'eager': {k: (v / head if v else None) for k, v in api['eager'].items()},
                                         'graph': {k: (v / head if v else None) for k, v in api['graph'].items()}}
        if 'graph_error' in api:
            out['config']['api_graph_error'] = api['graph_error']
    if async2 is not None:
        out['config']['async_2_subbatches_env_steps_per_s'] = async2
    if chain_sweep:
        out['config']['chain_sweep'] = chain_sweep
    out['config'].update(secondary)
    if small is not None:
        out['config']['small'] = small
    if sweep is not None:
        out['config']['sweep'] = sweep
        out['config']['sweep_note'] = ('envs_per_gpu is a tunable (VecGridWorld(num_envs)): same workload, same window, 3 windows each; '
                                       'kernel_us_per_65536_env_steps = the launch time scaled to the headline batch')
    if large is not None:
        out['config']['large'] = large
    if facade is not None:
        out['config']['facade_1env'] = facade
    if iss is not None:
        out['issue'] = iss
    if not args.no_cpu_baseline:
        # on rank 0 at every N (north star: "next to the reference CPU path timed on the node's own host cores ... in the
        # same run").  The windows are over; with N > 1 the other ranks have returned and wait in the closing gloo barrier
        # (blocked in a socket read, not spinning), so rank 0 has the node's granted cores for the sample.
        out['cpu_baseline'] = cpu_baseline(args.seed, scale=args.cpu_sample_scale)
        out['cpu_baseline']['timed_on'] = 'rank 0, after the timed windows' + ('' if world == 1 else f' (the other {world - 1} ranks idle in the closing barrier)')
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
