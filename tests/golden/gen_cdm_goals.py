#!/usr/bin/env python3
"""The IGLU CDM target structures as plain data for the GPU box: the reference keeps them in skills/goals.pkl
({name: sparse [(x, y, z, colour)]}, 156 structures; loaded by its examples through Tasks.to_dense).  Writes
tests/golden/cdm_goals.npz: names [156] and dense int8 [156, 9, 11, 11] grids ([y+1][x+5][z+5], the layout of
gridworld/tasks/task.py:168-175).  Build container only (needs /root/reference)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402


def main():
    goals = H.load_cdm_goals()
    names = sorted(goals, key=lambda n: int(n[1:]))
    dense = np.stack([goals[n] for n in names]).astype(np.int8)
    path = os.path.join(HERE, 'cdm_goals.npz')
    np.savez_compressed(path, names=np.array(names), dense=dense)
    nz = (dense != 0).reshape(len(names), -1).sum(1)
    print('cdm_goals: %d structures, %d..%d blocks (mean %.1f), top level %d -> %.0f KiB' % (
        len(names), nz.min(), nz.max(), nz.mean(), np.nonzero(dense)[1].max(), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
