"""Seeded synthetic task generators for the BASELINE configs (device-side, torch).

rt20  : the shape of RandomTasks(max_blocks=20, height_levels=1, max_dist=2, num_colors=6)
        (gridworld/tasks/task_set.py:135-157): 20 blocks on level 0 inside the 5x5 window centred
        on a first block drawn from [2,8]^2 (the restriction avoids the reference's infinite
        rejection loop near corners) -> 49 admissible translations per rotation.
uniform20 : 20 distinct cells uniform over the 1089, colours U{1..6}.
dummy : DUMMY_TASK (task_set.py:160): one blue block at dense [8,10,10], invariant=False.
cdm   : real IGLU targets (the CDM structures the reference ships in skills/goals.pkl, passed in as dense grids)
        tiled over the batch, each with a random PARTIAL starting grid: a random subset of the target's blocks
        already built (GridWorld.reset places them, env.py:234-238; inventory < 20) plus, for a third of the envs, a
        few blocks that are not part of the target (negative ids in the synthetic target `target - start`,
        env.py:227-231) -- the Subtasks-shaped workload: has_start, task_start reads, the stale-cache path.
"""
import torch


def rt20(n, seed, device='cpu'):
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    bx = torch.randint(2, 9, (n,), generator=g)
    bz = torch.randint(2, 9, (n,), generator=g)
    pick = torch.rand((n, 25), generator=g).argsort(dim=1)[:, :20]        # 20 of the 25 window cells
    colour = torch.randint(1, 7, (n, 20), generator=g).to(torch.int8)
    x = bx[:, None] + pick // 5 - 2
    z = bz[:, None] + pick % 5 - 2
    out = torch.zeros((n, 9, 11, 11), dtype=torch.int8)
    out[torch.arange(n)[:, None], 0, x, z] = colour
    return out.to(device)


def uniform20(n, seed, device='cpu'):
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    pick = torch.rand((n, 1089), generator=g).argsort(dim=1)[:, :20]
    colour = torch.randint(1, 7, (n, 20), generator=g).to(torch.int8)
    out = torch.zeros((n, 1089), dtype=torch.int8)
    out[torch.arange(n)[:, None], pick] = colour
    return out.reshape(n, 9, 11, 11).to(device)


def dummy(device='cpu'):
    out = torch.zeros((1, 9, 11, 11), dtype=torch.int8)
    out[0, 8, 10, 10] = 1
    return out.to(device)


def cdm(n, seed, goals, device='cpu'):
    """goals: dense int8 [G, 9, 11, 11] (e.g. tests/golden/cdm_goals.npz['dense']).  Returns (targets, starts),
    int8 [n, 9, 11, 11] each."""
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    goals = torch.as_tensor(goals).to(torch.int8).reshape(-1, 1089)
    pick = torch.randint(0, goals.shape[0], (n,), generator=g)
    targets = goals[pick]
    # a random fraction of every target's blocks is already there at reset
    frac = torch.rand((n, 1), generator=g)
    keep = (torch.rand((n, 1089), generator=g) < frac) & (targets != 0)
    # ... but never all of them: a task whose target is already built is `done` at every step (tasks/task.py:107)
    # and would reset every step; one random block of every target is always left to build
    score = torch.rand((n, 1089), generator=g).masked_fill(targets == 0, -1.0)
    keep[torch.arange(n), score.argmax(dim=1)] = False
    starts = torch.where(keep, targets, torch.zeros_like(targets))
    # a third of the envs: up to three foreign blocks on the lower levels, on cells the target leaves empty
    extra_env = torch.rand((n,), generator=g) < (1.0 / 3.0)
    for _ in range(3):
        cell = torch.randint(0, 3 * 121, (n,), generator=g)
        colour = torch.randint(1, 7, (n,), generator=g).to(torch.int8)
        free = extra_env & (targets[torch.arange(n), cell] == 0) & (torch.rand((n,), generator=g) < 0.7)
        idx = torch.nonzero(free)[:, 0]
        starts[idx, cell[idx]] = colour[idx]
    return targets.reshape(n, 9, 11, 11).to(device), starts.reshape(n, 9, 11, 11).to(device)
