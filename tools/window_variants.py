"""Runs on the GPU box: host-side cost of different ways to launch and time a 20-step window (the size the
round-end driver uses).  Prints, per variant, the wall time of the window over several repetitions."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
from gridworld_amd import VecGridWorld, workloads, _lib as L  # noqa: E402

N, K, W = 65536, 20, 5
dev = torch.device('cuda', 0)
env = VecGridWorld(N, device=dev, action_space='walking', size_reward=False, max_steps=250, autoreset=True)
env.set_tasks(workloads.rt20(N, seed=0, device=dev))
env.reset()
g = torch.Generator(device=dev)
g.manual_seed(0)
sn = torch.randint(0, 250, (N,), generator=g, device=dev, dtype=torch.int32)
env.agent_buf[:, 48] = (sn & 0xff).to(torch.uint8)
env.agent_buf[:, 49] = (sn >> 8).to(torch.uint8)
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 0.3:
    env.rollout(250, seed=17 + n, t0=n)
    torch.cuda.synchronize()
    n += 250
acts = env.fill_actions(W + K, seed=0)
ptrs = [acts[t].data_ptr() for t in range(W + K)]
fn, ctx = env.lib.igw_step_walking, env.ctx
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def launch(t):
    fn(ctx, ptrs[t], stream)


def make_graph(lo, hi):
    gr = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    cap.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.graph(gr, stream=cap):
        cs = C.c_void_p(cap.cuda_stream)
        for t in range(lo, hi):
            fn(ctx, ptrs[t], cs)
    torch.cuda.current_stream(dev).wait_stream(cap)
    gr.replay()
    torch.cuda.synchronize()
    return gr


g18, g19, g20 = make_graph(W + 2, W + K), make_graph(W + 1, W + K), make_graph(W, W + K)
g17, g16, g14 = make_graph(W + 3, W + K), make_graph(W + 4, W + K), make_graph(W + 6, W + K)


def window(head, graph, ev_inside, ev_timing=True, stamps=None):
    for t in range(W):
        launch(t)
    ev0, ev1 = torch.cuda.Event(enable_timing=ev_timing), torch.cuda.Event(enable_timing=ev_timing)
    ev0.record(); ev1.record(); ev1.query()
    torch.cuda.synchronize()
    if not ev_inside:
        ev0.record()
    a = time.perf_counter()
    if ev_inside:
        ev0.record()
    b = time.perf_counter()
    for t in range(W, W + head):
        launch(t)
    c = time.perf_counter()
    if graph is not None:
        graph.replay()
    d = time.perf_counter()
    ev1.record()
    e = time.perf_counter()
    while not ev1.query():
        pass
    f = time.perf_counter()
    torch.cuda.synchronize()
    z = time.perf_counter()
    evt = ev0.elapsed_time(ev1) * 1e3 if ev_timing else float('nan')
    return [1e6 * x for x in (z - a, b - a, c - b, d - c, e - d, f - e, z - f)] + [evt]


variants = [
    ('A  ev0 inside, 2 eager + graph18', dict(head=2, graph=g18, ev_inside=True)),
    ('B  ev0 before, 2 eager + graph18', dict(head=2, graph=g18, ev_inside=False)),
    ('C  ev0 before, 1 eager + graph19', dict(head=1, graph=g19, ev_inside=False)),
    ('D  ev0 before, graph20', dict(head=0, graph=g20, ev_inside=False)),
    ('E  ev0 before, 20 eager', dict(head=K, graph=None, ev_inside=False)),
    ('H  ev0 before, 3 eager + graph17', dict(head=3, graph=g17, ev_inside=False)),
    ('I  ev0 before, 4 eager + graph16', dict(head=4, graph=g16, ev_inside=False)),
    ('J  ev0 before, 6 eager + graph14', dict(head=6, graph=g14, ev_inside=False)),
    ('F  untimed events, 2 eager + graph18', dict(head=2, graph=g18, ev_inside=False, ev_timing=False)),
    ('G  untimed events, 20 eager', dict(head=K, graph=None, ev_inside=False, ev_timing=False)),
]
print('%-40s %8s | %6s %6s %6s %6s %6s %6s | %8s' % ('variant (median of 9)', 'wall us', 'ev0', 'head', 'graph', 'ev1', 'spin', 'sync', 'events us'))
for rep in range(2):
    for name, kw in variants:
        rows = [window(**kw) for _ in range(9)]
        rows.sort(key=lambda r: r[0])
        m = rows[len(rows) // 2]
        print('%-40s %8.1f | %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f | %8.1f   (min %.1f max %.1f)' % (
            name, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], rows[0][0], rows[-1][0]))

# cost of individual eager launches right after a synchronize
torch.cuda.synchronize()
for rep in range(3):
    ts = [time.perf_counter()]
    for t in range(W, W + 8):
        launch(t)
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    print('eager launch host cost after a sync, us:', ' '.join('%.1f' % (1e6 * (ts[i + 1] - ts[i])) for i in range(8)))
