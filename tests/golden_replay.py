"""Replays a golden fixture (tests/golden/*.npz, recorded from the Python reference)
through a batched driver and checks every output bit-for-bit.

A driver exposes (E = number of envs in the fixture):
    set_tasks(targets[E,9,11,11] i8, starts[E,9,11,11] i8, invariant=bool)
    set_initial_pose(poses[E,5])                     (only if the fixture has one)
    reset(mask or None) -> fills outputs
    step_walking(actions[E]) / step_flying(mv[E,3], cam[E,2], inv[E], place[E])
    outputs(): dict(agentPos f32[E,5], inventory f32[E,6], compass f32[E], reward f32[E],
                    done u8[E], grid i8[E,1089], internal f64[E,8] or None)
Both the CPU oracle (oracle.OracleBatch) and the HIP path (gridworld_amd.VecGridWorld)
are wrapped this way, so their tests read the same.
"""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

WALK_FIXTURES = ['s1_walk_dummy', 's1_walk_dummy_sizereward', 's2_walk_cdm', 's2_walk_cdm_sizereward',
                 's3_walk_rt20', 's5_scripted', 's5_scripted_scales', 's5_scripted_leak', 's5_init_pose',
                 's9_walk_no_select_and_place',
                 # starting grids with hundreds of blocks of one colour: inventories down to -1069 (env.py:243-246)
                 's12_wide_inventory', 's12_wide_inventory_sizereward']
FLY_FIXTURES = ['s4_fly_rt20', 's4_fly_cdm', 's9_fly_no_select_and_place']
DICT_FIXTURES = ['s8_walk_dict']
# recorded with the reference's sin/cos/atan2 replaced by correctly rounded ones (ref_harness.cr_libm)
CRLIBM_FIXTURES = ['s4_fly_crlibm', 's8_walk_dict_crlibm']


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    fx = {k: z[k] for k in z.files}
    fx['kwargs'] = json.loads(str(fx['kwargs']))
    fx['task_kwargs'] = json.loads(str(fx['task_kwargs']))
    fx['name'] = name
    fx['flying'] = 'act_movement' in fx
    fx['walkdict'] = 'act_buttons' in fx
    return fx


def bits32(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def bits64(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def replay(fx, driver, check_internal=True, max_steps=None, float_bits=True):
    """Returns the number of env-steps checked; raises AssertionError on the first mismatch."""
    E, T = fx['done'].shape
    if max_steps is not None:
        T = min(T, max_steps)
    starts = fx['starts']
    driver.set_tasks(fx['targets'], starts, invariant=fx['task_kwargs'].get('invariant', True))
    if 'init_pose' in fx:
        driver.set_initial_pose(fx['init_pose'])
    driver.reset(None)
    out = driver.outputs()
    name = fx['name']
    assert np.array_equal(bits32(out['agentPos']), bits32(fx['reset_agentPos'])), f'{name}: reset agentPos'
    assert np.array_equal(out['inventory'], fx['reset_inventory']), f'{name}: reset inventory'
    assert np.array_equal(bits32(out['compass']), bits32(fx['reset_compass'])), f'{name}: reset compass'
    assert np.array_equal(out['grid'].reshape(E, -1), fx['reset_grid'].reshape(E, -1)), f'{name}: reset grid'
    grid = starts.reshape(E, -1).copy()
    ar = np.arange(E)
    for t in range(T):
        rb = fx['reset_before'][:, t].astype(bool)
        if rb.any():
            driver.reset(rb)
            grid[rb] = starts.reshape(E, -1)[rb]
        if fx['walkdict']:
            driver.step_walking_dict(fx['act_buttons'][:, t], fx['act_camera'][:, t])
        elif fx['flying']:
            driver.step_flying(fx['act_movement'][:, t], fx['act_camera'][:, t], fx['act_inventory'][:, t],
                               fx['act_placement'][:, t])
        else:
            driver.step_walking(fx['actions'][:, t])
        out = driver.outputs()
        idx = fx['grid_change_idx'][:, t].astype(np.int64)
        ch = idx >= 0
        grid[ar[ch], idx[ch]] = fx['grid_change_val'][:, t][ch]

        def bad(mask):
            e = int(np.nonzero(mask)[0][0])
            return f'{name}: env {e} step {t}'
        m = out['done'].astype(bool) != fx['done'][:, t].astype(bool)
        assert not m.any(), bad(m) + f' done {out["done"][m][:1]} vs {fx["done"][:, t][m][:1]}'
        m = np.asarray(out['reward'], np.float32) != fx['reward'][:, t].astype(np.float32)
        assert not m.any(), bad(m) + f' reward {out["reward"][m][:1]} vs {fx["reward"][:, t][m][:1]}'
        m = (out['grid'].reshape(E, -1) != grid).any(-1)
        assert not m.any(), bad(m) + ' grid'
        m = (out['inventory'] != fx['inventory'][:, t]).any(-1)
        assert not m.any(), bad(m) + f' inventory {out["inventory"][m][:1]} vs {fx["inventory"][:, t][m][:1]}'
        if float_bits:
            m = (bits32(out['agentPos']) != bits32(fx['agentPos'][:, t])).any(-1)
            assert not m.any(), bad(m) + f' agentPos {out["agentPos"][m][:1]} vs {fx["agentPos"][:, t][m][:1]}'
            m = bits32(out['compass']) != bits32(fx['compass'][:, t])
            assert not m.any(), bad(m) + f' compass {out["compass"][m][:1]} vs {fx["compass"][:, t][m][:1]}'
        else:
            assert np.array_equal(out['agentPos'], fx['agentPos'][:, t]), f'{name}: step {t} agentPos'
            assert np.array_equal(out['compass'], fx['compass'][:, t]), f'{name}: step {t} compass'
        if check_internal and out.get('internal') is not None:
            m = (bits64(out['internal']) != bits64(fx['internal'][:, t])).any(-1)
            assert not m.any(), bad(m) + f' internal {out["internal"][m][:1]} vs {fx["internal"][:, t][m][:1]}'
    assert np.array_equal(grid, fx['grid_final'].reshape(E, -1)) or T != fx['done'].shape[1]
    return E * T


SUBTASK_FIXTURES = ['s11_subtasks_env', 's11_subtasks_env_nosize']


def load_subtasks_fixture(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    fx = {k: z[k] for k in z.files}
    fx['kwargs'] = json.loads(str(fx['kwargs']))
    fx['name'] = name
    return fx


def subtasks_table(fx):
    """Task table of a Subtasks fixture: row e * R + r = the task env e was given for its episode r (target of
    turn k, start = structure of turn k - 1, full_grid = the final structure; tasks/task.py:260-286)."""
    E, R = fx['ep_targets'].shape[:2]
    tg = fx['ep_targets'].reshape(E * R, 9, 11, 11)
    st = fx['ep_starts'].reshape(E * R, 9, 11, 11)
    fg = np.repeat(fx['full_grids'], R, axis=0)
    return tg, st, fg


def replay_subtasks(fx, driver, check_internal=True):
    """Replays an env-level set_task_generator(Subtasks) fixture (tests/golden/gen_subtasks_env.py): every reset
    gives the env the task row the reference's generator produced for that episode.  The driver additionally
    exposes  set_task_table(targets, starts, full_grids)  and  assign_tasks(mask[E], rows[E]).
    Returns the number of env-steps checked."""
    E, T = fx['done'].shape
    R = fx['ep_targets'].shape[1]
    name = fx['name']
    tg, st, fg = subtasks_table(fx)
    driver.set_task_table(tg, st, fg)
    ar = np.arange(E)
    grid = np.zeros((E, 1089), np.int8)
    for t in range(T):
        rb = fx['reset_before'][:, t].astype(bool) if t else np.ones(E, bool)
        if rb.any():
            ep = fx['episode'][:, t]
            driver.assign_tasks(rb, ar * R + ep)
            driver.reset(rb)
            out = driver.outputs()
            grid[rb] = fx['ep_starts'][ar[rb], ep[rb]].reshape(-1, 1089)
            assert np.array_equal(out['grid'].reshape(E, -1)[rb], grid[rb]), f'{name}: step {t} reset grid'
            assert np.array_equal(out['inventory'][rb], fx['ep_reset_inventory'][ar[rb], ep[rb]]), f'{name}: step {t} reset inventory'
            assert not np.asarray(out['agentPos'])[rb].any(), f'{name}: step {t} reset agentPos'
            if 'env_max_int' in out:
                assert np.array_equal(out['env_max_int'][rb], fx['ep_env_max_int'][ar[rb], ep[rb]]), \
                    f'{name}: step {t} GridWorld.max_int at reset {out["env_max_int"][rb]} vs {fx["ep_env_max_int"][ar[rb], ep[rb]]}'
        driver.step_walking(fx['actions'][:, t])
        out = driver.outputs()
        idx = fx['grid_change_idx'][:, t].astype(np.int64)
        ch = idx >= 0
        grid[ar[ch], idx[ch]] = fx['grid_change_val'][:, t][ch]

        def bad(mask):
            e = int(np.nonzero(mask)[0][0])
            return f'{name}: env {e} step {t}'
        m = out['done'].astype(bool) != fx['done'][:, t].astype(bool)
        assert not m.any(), bad(m) + f' done {out["done"][m][:1]} vs {fx["done"][:, t][m][:1]}'
        m = np.asarray(out['reward'], np.float32) != fx['reward'][:, t].astype(np.float32)
        assert not m.any(), bad(m) + f' reward {out["reward"][m][:1]} vs {fx["reward"][:, t][m][:1]}'
        m = (out['grid'].reshape(E, -1) != grid).any(-1)
        assert not m.any(), bad(m) + ' grid'
        m = (out['inventory'] != fx['inventory'][:, t]).any(-1)
        assert not m.any(), bad(m) + ' inventory'
        m = (bits32(out['agentPos']) != bits32(fx['agentPos'][:, t])).any(-1)
        assert not m.any(), bad(m) + ' agentPos'
        m = bits32(out['compass']) != bits32(fx['compass'][:, t])
        assert not m.any(), bad(m) + ' compass'
        if 'syn_max_int' in out:
            m = out['syn_max_int'] != fx['syn_max_int'][:, t]
            assert not m.any(), bad(m) + f' synthetic max_int {out["syn_max_int"][m][:1]} vs {fx["syn_max_int"][:, t][m][:1]}'
        if check_internal and out.get('internal') is not None:
            m = (bits64(out['internal']) != bits64(fx['internal'][:, t])).any(-1)
            assert not m.any(), bad(m) + ' internal'
    assert np.array_equal(grid, fx['grid_final'].reshape(E, -1))
    return E * T


class OracleDriver:
    """oracle.OracleBatch behind the replay interface (CPU)."""

    def __init__(self, fx):
        from oracle import oracle as O
        self.b = O.OracleBatch(len(fx['targets']), **fx['kwargs'])

    def set_tasks(self, targets, starts, invariant=True):
        self.b.set_tasks(targets, starts, invariant=invariant)

    def set_initial_pose(self, poses):
        self.b.set_initial_pose(poses)

    def reset(self, mask):
        self.b.reset(mask)

    def step_walking(self, actions):
        self.b.step_walking(actions)

    def step_flying(self, mv, cam, inv, place):
        self.b.step_flying(mv, cam, inv, place)

    def step_walking_dict(self, buttons, cam):
        self.b.step_walking_dict(buttons, cam)

    def set_task_table(self, targets, starts, full_grids):
        self._table = (targets, starts, full_grids)

    def assign_tasks(self, mask, rows):
        tg, st, fg = self._table
        for e in np.nonzero(mask)[0]:
            self.b.envs[e].set_task(tg[rows[e]], st[rows[e]], fg[rows[e]])

    def outputs(self):
        b = self.b
        ts = [e.task_state() for e in b.envs]
        return dict(agentPos=b.agentPos, inventory=b.inventory, compass=b.compass, reward=b.reward,
                    done=b.done, grid=b.grid, internal=b.internals(),
                    env_max_int=np.array([s['env_max_int'] for s in ts]),
                    syn_max_int=np.array([s['syn_max_int'] for s in ts]))
