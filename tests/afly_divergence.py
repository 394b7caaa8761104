#!/usr/bin/env python3
"""A-fly residual against GLIBC, at scale (GPU box; test infrastructure: it drives the CPU oracle).

Two action spaces send arbitrary float angles through the general trig (csrc/igw_trig.h): `flying` (the text below) and
`walking` with discretize=False (`--action-space walking_dict`: continuous camera deltas + button combinations whose
diagonal strafes go through atan2; core/world.py:396-414, 163-201) -- same comparison, same contract.

BASELINE configs[3] -- 65,536 flying envs, rt20 targets, uniform random actions, full 250-step episodes -- on the HIP
path (its own correctly rounded sincos / atan2, csrc/igw_trig.h) against the CPU oracle computing with GLIBC's
sin / cos / atan2, i.e. what the Python reference calls (CPython's math module), for EVERY env of the batch and EVERY
step.  This is the independent check of the flying path: nothing of the product's trig is on the oracle's side.

glibc's results are within 1 ulp but not always correctly rounded, so the two trajectories may part in the last bit
of a float64 internal; such a difference reaches an OUTPUT only when it flips a rounding -- the float32 cast of an
observation, or, far rarer, normalize() of a ray sample / a collision test, which then changes grid, inventory or
reward.  Per env the first step at which an integer output (grid, inventory, reward, done) or a float32 observation
(agentPos, compass) differs is recorded; an env that has diverged is out of the exposure from then on.

    python tests/afly_divergence.py [--passes 7] [--envs 65536] [--steps 250] [--out gpurun_out/afly_divergence.json]

tests/test_gpu_flying.py runs one pass under the driver; the long run is logged in profiles/r05_afly_divergence.json."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402


SPACES = ('flying', 'walking_dict')


def draw_actions(space, N, g, dev):
    """One step's uniform random actions on the device + their host copies (what the oracle is stepped with)."""
    import torch
    cam = torch.rand((N, 2), generator=g, device=dev) * 10 - 5          # continuous camera deltas: off-lattice yaw / pitch
    if space == 'flying':
        a = dict(movement=torch.rand((N, 3), generator=g, device=dev) * 2 - 1, camera=cam,
                 inventory=torch.randint(0, 7, (N,), generator=g, device=dev, dtype=torch.int32),
                 placement=torch.randint(0, 3, (N,), generator=g, device=dev, dtype=torch.int32))
        return a, (a['movement'].cpu().numpy(), cam.cpu().numpy(), a['inventory'].cpu().numpy(), a['placement'].cpu().numpy())
    # walking Dict (discretize=False, core/world.py:396-414): forward, back, left, right, jump, attack, use each pressed with
    # probability 0.2 (so diagonal strafes -- atan2 of two non-zero components -- and opposite keys occur), a hotbar slot on
    # 30 % of the steps
    keys = (torch.rand((N, 7), generator=g, device=dev) < 0.2).to(torch.uint8)
    hot = torch.where(torch.rand((N,), generator=g, device=dev) < 0.3,
                      torch.randint(0, 7, (N,), generator=g, device=dev), torch.zeros((N,), dtype=torch.int64, device=dev)).to(torch.uint8)
    b = torch.cat([keys, hot[:, None]], dim=1).contiguous()
    return dict(buttons=b, camera=cam), (b.cpu().numpy(), cam.cpu().numpy())


def one_pass(N, T, seed, ob=None, grid_every=10, cores=None, space='flying'):
    """One batch of N flying envs for T steps (max_steps = T: every episode runs its full length, no resets).
    Returns (result dict, the OracleBatch).  Both sides start from FRESH agents: an Agent's active_block, dy and
    time_int_steps survive reset() in the reference, so an OracleBatch that has been stepped before is not
    in the state a new VecGridWorld is in (the first 7-pass run of this script reused one: 3 'divergences' at steps
    2-3, all of them placements with the colour left over from the previous pass -- profiles/r05_afly_divergence.json
    keeps that run's cases as a record of what the comparison catches)."""
    import torch
    from gridworld_amd import VecGridWorld, workloads
    from oracle import oracle as O
    cores = cores or len(os.sched_getaffinity(0))
    assert space in SPACES
    kw = dict(size_reward=False, action_space='flying' if space == 'flying' else 'walking', max_steps=T)
    tg = workloads.rt20(N, seed=seed)
    env = VecGridWorld(N, autoreset=False, **kw, **({} if space == 'flying' else dict(discretize=False)))
    env.set_tasks(tg.to(env.device))
    env.reset()
    ob = O.OracleBatch(N, **kw)              # fresh agents; default trig mode: libm (glibc)
    O.use_device_trig(False)
    ob.set_tasks(tg.numpy())
    ob.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(seed * 7919 + 4)
    first_int = np.full(N, -1)           # first step at which grid / inventory / reward / done differ
    first_f32 = np.full(N, -1)           # first step at which agentPos / compass (float32) differ
    dev = env.device
    cases, hist = [], []       # details of the first few integer divergences; the last steps' actions (host copies)
    for t in range(T):
        a, ah = draw_actions(space, N, g, dev)
        env.step(a)
        hist = (hist + [ah])[-8:]
        if space == 'flying':
            ob.step_flying(*ah, nthreads=cores)
        else:
            ob.step_walking_dict(*ah, nthreads=cores)
        out = env.out_buf.cpu().numpy()                  # one copy: the 64-byte output records (include/igw.h)
        f = out[:, :52].copy().view(np.float32)
        d_int = (out[:, 52] != ob.done) | (f[:, 12] != ob.reward) | (f[:, 5:11] != ob.inventory).any(-1)
        if t % grid_every == grid_every - 1 or t == T - 1:
            d_int |= (env.grid_buf.cpu().numpy()[:, :1089] != ob.grid).any(-1)
        d_f32 = (f[:, :5].view(np.uint32) != ob.agentPos.view(np.uint32)).any(-1) | \
            (f[:, 11].view(np.uint32) != ob.compass.view(np.uint32))
        new_int = np.nonzero((first_int < 0) & d_int)[0]
        if len(new_int) and len(cases) < 16:     # rare: what the two sides hold right after the diverging step
            di = env.internals()
            for e in new_int[:4]:
                k0 = t + 1 - len(hist)
                cases.append(dict(env=int(e), step=t, device=dict(done=int(out[e, 52]), reward=float(f[e, 12]), inventory=f[e, 5:11].tolist(),
                                                                  agentPos=f[e, :5].tolist(), internal=[float(v).hex() for v in di[e]]),
                                  oracle_glibc=dict(done=int(ob.done[e]), reward=float(ob.reward[e]), inventory=ob.inventory[e].tolist(),
                                                    agentPos=ob.agentPos[e].tolist(), internal=[float(v).hex() for v in ob.envs[e].internal()]),
                                  grid_cells_that_differ=np.nonzero(env.grid_buf[e].cpu().numpy()[:1089] != ob.grid[e])[0].tolist(),
                                  target_cells=np.nonzero(tg[e].numpy().reshape(-1))[0].tolist(),
                                  actions_from_step=k0,
                                  actions=[(dict(movement=[float(v) for v in h[0][e]], camera=[float(v) for v in h[1][e]],
                                                 inventory=int(h[2][e]), placement=int(h[3][e])) if space == 'flying' else
                                            dict(buttons=[int(v) for v in h[0][e]], camera=[float(v) for v in h[1][e]])) for h in hist]))
        first_int[(first_int < 0) & d_int] = t
        first_f32[(first_f32 < 0) & d_f32] = t
    fin = env.internals()
    ref = ob.internals()
    diff64 = (fin.view(np.uint64) != ref.view(np.uint64)).any(-1)
    clean = (first_int < 0) & (first_f32 < 0)
    res = dict(action_space=space, envs=N, steps=T, seed=seed, env_steps=N * T,
               envs_with_integer_divergence=int((first_int >= 0).sum()),
               envs_with_float32_obs_divergence=int((first_f32 >= 0).sum()),
               first_integer_divergence_step=None if (first_int < 0).all() else int(first_int[first_int >= 0].min()),
               first_float32_divergence_step=None if (first_f32 < 0).all() else int(first_f32[first_f32 >= 0].min()),
               # exposure: env-steps compared while the env had not diverged yet (in that class)
               exposure_integer=int(np.where(first_int >= 0, first_int + 1, T).sum()),
               exposure_float32=int(np.where(first_f32 >= 0, first_f32 + 1, T).sum()),
               envs_ending_with_a_float64_difference=int(diff64.sum()),
               max_abs_float64_deviation_of_clean_envs=float(np.abs(fin[clean, :6] - ref[clean, :6]).max()) if clean.any() else None,
               all_done=bool(env.done.all()), min_inventory=int(env.inventory.min()), integer_divergence_cases=cases)
    return res, ob


def summarize(passes, wall):
    n_int = sum(p['envs_with_integer_divergence'] for p in passes)
    n_f32 = sum(p['envs_with_float32_obs_divergence'] for p in passes)
    e_int = sum(p['exposure_integer'] for p in passes)
    e_f32 = sum(p['exposure_float32'] for p in passes)

    def bound(k, n):   # one-sided 95 % upper bound of a Poisson rate: 3 / n for k = 0, else (k + 2 sqrt(k) + 2) / n (conservative)
        return (3.0 if k == 0 else k + 2.0 * k ** 0.5 + 2.0) / n
    return dict(
        what='HIP %s path (own correctly rounded trig) vs the CPU oracle with GLIBC trig, every env, every step' %
             '/'.join(sorted({p.get('action_space', 'flying') for p in passes})),
        env_steps_compared=sum(p['env_steps'] for p in passes), passes=len(passes),
        integer_divergences=n_int, float32_obs_divergences=n_f32,
        integer_divergence_cases=[c for p in passes for c in p.get('integer_divergence_cases', [])],
        envs_ending_with_a_float64_difference=sum(p['envs_ending_with_a_float64_difference'] for p in passes),
        float64_difference_rate_per_episode=sum(p['envs_ending_with_a_float64_difference'] for p in passes) /
        float(sum(p['envs'] for p in passes)),
        integer_divergence_rate_per_env_step=n_int / e_int, float32_divergence_rate_per_env_step=n_f32 / e_f32,
        integer_divergence_rate_upper_bound_95=bound(n_int, e_int),
        float32_divergence_rate_upper_bound_95=bound(n_f32, e_f32),
        first_integer_divergence_step=min([p['first_integer_divergence_step'] for p in passes
                                           if p['first_integer_divergence_step'] is not None], default=None),
        first_float32_divergence_step=min([p['first_float32_divergence_step'] for p in passes
                                           if p['first_float32_divergence_step'] is not None], default=None),
        wall_s=round(wall, 1), per_pass=passes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--passes', type=int, default=7)
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=250)
    ap.add_argument('--seed0', type=int, default=9000)
    ap.add_argument('--seeds', default='', help='explicit comma-separated seeds instead of seed0 .. seed0 + passes - 1')
    ap.add_argument('--action-space', choices=SPACES, default='flying',
                    help="flying (BASELINE configs[3]) or walking_dict (walking with discretize=False: continuous camera deltas)")
    ap.add_argument('--out', default='gpurun_out/afly_divergence.json')
    a = ap.parse_args()
    t0 = time.time()
    passes, ob = [], None
    seeds = [int(x) for x in a.seeds.split(',') if x] or [a.seed0 + k for k in range(a.passes)]
    for sd in seeds:
        r, ob = one_pass(a.envs, a.steps, sd, ob, space=a.action_space)
        passes.append(r)
        print(json.dumps(r), flush=True)
    s = summarize(passes, time.time() - t0)
    s['command'] = ' '.join(sys.argv)
    os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(s, f, indent=1)
    print(json.dumps({k: v for k, v in s.items() if k != 'per_pass'}, indent=1))


if __name__ == '__main__':
    main()
