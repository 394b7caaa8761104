"""Multi-GPU launch helpers: one process per GPU, envs sharded with no data-path collective.

Envs are fully independent, so ranks never exchange state.  The only
collectives are control-plane: a barrier around the timing window and one all-reduce of the
per-rank env-step counters / elapsed time (RCCL over xGMI on GPUs, gloo on CPU in tests).
"""
import os

import torch
import torch.distributed as dist


def env_info():
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))


def init(backend=None):
    """Initialises torch.distributed from the torchrun environment (no-op for world size 1).

    On GPUs the group is MIXED: gloo for CPU tensors, nccl (= RCCL over xGMI) for CUDA tensors.  The control plane
    of a run -- rendezvous of the shared-memory barrier, the reduction of the timing windows -- goes over gloo, so a
    bench line is printed whatever the state of the RCCL stack; the per-rank step counts are gathered over RCCL
    (gather_counts_rccl), falling back to gloo if that collective raises."""
    rank, local_rank, world = env_info()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('IGW_DIST_BACKEND') or ('cpu:gloo,cuda:nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'nccl' in backend and not os.environ.get('IGW_SHARE_GPU'):
            torch.cuda.set_device(local_rank)
        # (no silent re-initialisation when the mixed group cannot be built: a rank that fell back to gloo alone while
        # its peers stayed in the mixed group would hang them; IGW_DIST_BACKEND=gloo selects the control plane only)
        dist.init_process_group(backend)
    return rank, local_rank, world


def shard_envs(total_envs, rank, world):
    """Contiguous env range [lo, hi) of `rank` (weak scaling passes total = per_gpu * world)."""
    lo = total_envs * rank // world
    hi = total_envs * (rank + 1) // world
    return lo, hi


def barrier(device=None):
    """Control-plane barrier: explicitly over gloo when the group has a CPU backend (ctl_barrier); `device` is
    accepted for old callers."""
    ctl_barrier()


class NodeBarrier:
    """Barrier of the ranks of ONE node through a page of shared memory (/dev/shm): every rank publishes the
    number of the barrier it has reached and spins until all the others have.  A few microseconds, against
    50-100 us for an RCCL / gloo barrier -- which matters when the barrier is part of a timing bracket around a
    window of a few hundred microseconds (bench.py).  Control plane only; the env data never crosses ranks."""

    def __init__(self, tag=None):
        import numpy as np
        self.rank, _, self.world = env_info()
        self.epoch = 0
        self.arr = None
        if self.world == 1:
            return
        tag = tag or 'igw_barrier_%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', '0'))
        self.path = os.path.join('/dev/shm' if os.path.isdir('/dev/shm') else '/tmp', tag)
        if self.rank == 0:  # created (zeroed) by rank 0 before anybody maps it
            tmp = self.path + '.tmp%d' % os.getpid()
            np.zeros(64 * self.world, np.int64).tofile(tmp)
            os.replace(tmp, self.path)
        barrier()  # the collective one, once
        self.arr = np.memmap(self.path, dtype=np.int64, mode='r+', shape=(64 * self.world,))  # one 512-B stripe per rank
        barrier()
        if self.rank == 0:
            os.unlink(self.path)  # the mappings keep it alive

    def wait(self, timeout=120.0, spins=2000):
        """Spin for a few microseconds, then yield the CPU between polls (sched_yield, later short sleeps): with 8
        ranks on a 16-CPU quota next to RCCL proxy / watchdog threads, an unbounded busy-wait can starve the very
        rank it waits for."""
        if self.arr is None:
            return
        import time
        self.epoch += 1
        self.arr[64 * self.rank] = self.epoch
        others = [64 * r for r in range(self.world) if r != self.rank]
        t0 = time.perf_counter()
        n = 0
        for o in others:
            while self.arr[o] < self.epoch:
                n += 1
                if n > spins:
                    if n < 20 * spins:
                        os.sched_yield()
                    else:
                        time.sleep(50e-6)
                    if time.perf_counter() - t0 > timeout:
                        raise RuntimeError('NodeBarrier: rank %d timed out waiting for the others' % self.rank)


def ctl_barrier():
    """Barrier over the CONTROL plane only: an all-reduce of a CPU tensor (gloo).  dist.barrier() on a mixed
    cpu:gloo,cuda:nccl group may pick the NCCL side and then needs every rank's GPU stream to be alive."""
    if not dist.is_initialized():
        return
    if _cpu_ok():
        dist.all_reduce(torch.zeros(1, dtype=torch.int64))
    else:
        dist.barrier()


def shutdown(barrier=True):
    """destroy_process_group, after a closing control-plane barrier on the SUCCESS path (every rank leaves together: a
    rank that falls off the end of main while others are still inside a collective aborts them at teardown).  A rank
    that failed passes barrier=False: its peers are still in a collective or in the NodeBarrier spin and would never
    reach a closing barrier -- it must go down at once so that torchrun ends the group."""
    if dist.is_initialized():
        if barrier:
            try:
                ctl_barrier()
            except Exception:  # noqa: BLE001 -- a peer already gone: still tear our side down
                pass
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


def _cpu_ok():
    """The group can reduce CPU tensors (gloo present)."""
    return 'gloo' in str(dist.get_backend()) or 'cpu' in str(dist.get_backend_config())


def _ctl_device(device):
    return torch.device('cpu') if _cpu_ok() or device is None else device


def reduce_window(steps, seconds, device=None):
    """(total env-steps over all ranks, max elapsed seconds over ranks)."""
    if not dist.is_initialized():
        return int(steps), float(seconds)
    dev = _ctl_device(device)
    s = torch.tensor([int(steps)], dtype=torch.int64, device=dev)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(s.item()), float(t.item())


def reduce_windows(seconds, device=None):
    """Per timing window the MAX elapsed seconds over the ranks: one all-reduce for all windows of a run."""
    seconds = [float(x) for x in seconds]
    if not dist.is_initialized():
        return seconds
    t = torch.tensor(seconds, dtype=torch.float64, device=_ctl_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.cpu()]


def gather_counts(value, device=None):
    """all_gather of one int64 per rank (e.g. per-rank step counters) over the control plane."""
    if not dist.is_initialized():
        return [int(value)]
    mine = torch.tensor([int(value)], dtype=torch.int64, device=_ctl_device(device))
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [int(o.item()) for o in out]


def gather_floats(value, device=None):
    """all_gather of one float64 per rank over the control plane (e.g. per-rank kernel time)."""
    if not dist.is_initialized():
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=_ctl_device(device))
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(o.item()) for o in out]


def gather_counts_rccl(value, device):
    """The same gather over RCCL (CUDA tensors; xGMI between the GPUs of a node) -- the one place the data plane's
    fabric is used at all.  Returns (values, how).  Whether the RCCL collective worked is AGREED over the control plane
    (an all-reduce(MIN) of an ok flag over gloo) before anybody falls back: a rank whose collective raised (a timeout on
    one rank, say) must not enter the gloo gather alone while the others have already returned."""
    if not dist.is_initialized():
        return [int(value)], 'single process'
    if device is not None and device.type == 'cuda' and 'nccl' in str(dist.get_backend_config()):
        vals, why = None, ''
        try:
            import datetime
            mine = torch.tensor([int(value)], dtype=torch.int64, device=device)
            out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            work = dist.all_gather(out, mine, async_op=True)
            work.wait(timeout=datetime.timedelta(seconds=float(os.environ.get('IGW_RCCL_TIMEOUT_S', '120'))))
            torch.cuda.synchronize(device)
            vals = [int(o.item()) for o in out]
        except Exception as e:  # noqa: BLE001
            why = (str(e).splitlines() or [type(e).__name__])[0][:160]
            if not _cpu_ok():
                raise
        if not _cpu_ok():
            return vals, 'rccl all_gather of one int64 per rank'
        ok = torch.tensor([1 if vals is not None else 0], dtype=torch.int64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # gloo
        if int(ok.item()) == 1:
            return vals, 'rccl all_gather of one int64 per rank'
        return gather_counts(value), 'gloo (rccl all_gather failed on at least one rank%s)' % (': ' + why if why else '')
    return gather_counts(value), 'gloo'
