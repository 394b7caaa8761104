"""Seeded synthetic task generators for the BASELINE configs (device-side, torch).

rt20  : the shape of RandomTasks(max_blocks=20, height_levels=1, max_dist=2, num_colors=6)
        (gridworld/tasks/task_set.py:135-157): 20 blocks on level 0 inside the 5x5 window centred
        on a first block drawn from [2,8]^2 (the restriction avoids the reference's infinite
        rejection loop near corners) -> 49 admissible translations per rotation.
uniform20 : 20 distinct cells uniform over the 1089, colours U{1..6}.
dummy : DUMMY_TASK (task_set.py:160): one blue block at dense [8,10,10], invariant=False.
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
