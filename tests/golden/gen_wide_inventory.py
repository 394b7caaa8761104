#!/usr/bin/env python3
"""Golden vectors for inventories far below zero: starting grids with hundreds of blocks of one colour.

Run in the build container only:   python tests/golden/gen_wide_inventory.py   (needs /root/reference)

GridWorld.reset (gridworld/env.py:243-246) starts every colour at 20 and takes one off per starting block of that
colour -- a Python int, unbounded: two full floors of blue leave inventory[0] = -222, a full 9 x 11 x 11 zone -1069.
A placement needs inventory > 0 (core/world.py:317), a break refunds one (env.py:146-153 via on_remove), so such an
agent cannot place that colour until it has broken enough of it.  Scenarios (walking, Discrete(18)):

  0  two full floors of blue (242)            -> -222      5  148 blue (the last count an int8 holds: -128)
  1  three full floors of orange (363)        -> -343      6  149 blue (-129)
  2  the whole zone purple (1089)             -> -1069     7  two floors blue + 100 red + 30 yellow (-222, -80, -10)
  3  two floors yellow + one of green         -> -222/-101 8  one floor of each of the six colours (-101 x 6)
  4  the whole zone, colours striped by level -> 5 x -222 (one colour -101 ... see the code)

Agents start on top of their structure (initialize_world pose), look down and alternate breaks, placements of every
colour (hotbar ids with select_and_place: the place is attempted at once) and random actions; episodes run to
max_steps and are reset by the harness, so the negative reset inventory is recorded several times per env.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as GG  # noqa: E402
import ref_harness as H  # noqa: E402


def floors(levels, colour):
    g = np.zeros((9, 11, 11), np.int8)
    for y in levels:
        g[y] = colour
    return g


def scattered(rng, count, colour, into):
    free = np.flatnonzero(into.reshape(-1) == 0)
    into.reshape(-1)[rng.choice(free, count, replace=False)] = colour
    return into


def scenarios(rng):
    s = [floors([0, 1], 1), floors([0, 1, 2], 2), floors(range(9), 3), floors([0, 1], 6) + floors([2], 4)]
    striped = np.zeros((9, 11, 11), np.int8)
    for y in range(9):
        striped[y] = 1 + (y % 6)
    s.append(striped)
    s.append(scattered(rng, 148, 1, np.zeros((9, 11, 11), np.int8)))
    s.append(scattered(rng, 149, 1, np.zeros((9, 11, 11), np.int8)))
    s.append(scattered(rng, 30, 6, scattered(rng, 100, 5, floors([0, 1], 1))))
    s.append(sum(floors([y], y + 1) for y in range(6)))
    return np.stack(s)


def top_pose(start):
    """Standing on the highest block of the centre column (head = top of the block + 1.25), or on the ground."""
    col = np.flatnonzero(start[:, 5, 5])
    y = (col.max() - 1 + 0.5 + 1.25) if len(col) else -0.25
    return [0.0, float(y), 0.0, 0.0, 0.0]


def actions_for(rng, T):
    a = rng.randint(18, size=T)
    heavy = rng.rand(T) < 0.55   # mostly: break, place, hotbar (= place with select_and_place)
    a[heavy] = rng.choice([16, 16, 16, 17, 6, 7, 8, 9, 10, 11], size=int(heavy.sum()))
    a[:12] = [14] * 12           # look down first (pitch -60)
    a[12:20] = [16, 6, 16, 7, 16, 17, 8, 16]
    return a


def main():
    rng = np.random.RandomState(1212)
    starts = scenarios(rng)
    E, T = len(starts), 420
    # targets: the starting structure with a handful of cells changed (to remove / to recolour) plus a few more to
    # build, so that the synthetic target has negative ids and the reward path is exercised
    targets = starts.copy()
    for e in range(E):
        flat = targets[e].reshape(-1)
        occ, free = np.flatnonzero(flat), np.flatnonzero(flat == 0)
        flat[rng.choice(occ, 6, replace=False)] = 0
        flat[rng.choice(occ, 4, replace=False)] = rng.randint(1, 7, size=4)
        if len(free):
            flat[rng.choice(free, min(5, len(free)), replace=False)] = rng.randint(1, 7, size=min(5, len(free)))
    acts = np.stack([actions_for(rng, T) for _ in range(E)])
    pose = np.array([top_pose(s) for s in starts])
    sparse = [H.dense_to_sparse(s) for s in starts]
    for name, kw in (('s12_wide_inventory', dict(size_reward=False, max_steps=150)),
                     ('s12_wide_inventory_sizereward', dict(size_reward=True, max_steps=150))):
        n = E if not kw['size_reward'] else 4
        ref = H.run_batch(kw, targets[:n], sparse[:n], acts[:n], init_pose=pose[:n])
        GG.save(name, kw, targets[:n], sparse[:n], acts[:n], ref, init_pose=pose[:n])
        print(name, 'reset inventory min per env:', ref['reset_inventory'].min(1), 'step inventory range',
              ref['inventory'].min(), ref['inventory'].max(), 'changes', int((ref['inventory'][:, 1:] != ref['inventory'][:, :-1]).any(-1).sum()))


if __name__ == '__main__':
    main()
