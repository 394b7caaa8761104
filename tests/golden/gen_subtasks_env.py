#!/usr/bin/env python3
"""Golden vectors for the env-level Subtasks / full_grid path.

Records, from the imported Python reference, episodes of
    env = gym.make('IGLUGridworld-v0', vector_state=True, render=False, size_reward=..., max_steps=...)
    env.set_task_generator(Subtasks(dialog, structure_seq))
where every reset draws a new turn of the structure sequence: start = structure of turn k-1, target = turn k,
full_grid = the final structure (gridworld/tasks/task.py:208-286).  The user task's admissible translations then
come from full_grid (tasks/task.py:63-72) and GridWorld.max_int -- the user task evaluated on the starting grid
(env.py:241) -- reaches the agent through SizeReward at the first step of the episode (env.py:325-331).

Per env e and episode r the file stores the task the generator produced (target, start as dense grids, the
turn, GridWorld.max_int at reset, the reset observation) and per step the usual outputs; `episode[e, t]` names
the episode step t belongs to.  Build container only (needs /root/reference); writes s11_subtasks_env*.npz.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402


def structure_seq(rng, goal_dense, n_turns):
    """A CDM structure cut into n_turns cumulative building turns (a random build order, bottom level first)."""
    blocks = H.dense_to_sparse(goal_dense)
    order = sorted(range(len(blocks)), key=lambda i: (blocks[i][1], rng.rand()))
    blocks = [blocks[i] for i in order]
    n_turns = max(1, min(n_turns, len(blocks)))
    cuts = sorted(set(int(round(len(blocks) * (k + 1) / n_turns)) for k in range(n_turns)))
    return [blocks[:c] for c in cuts if c > 0]


def builder_actions(rng, T):
    """Walking Discrete(18) ids biased towards looking down and placing / breaking, so that episodes change the
    grid often (and complete turns now and then)."""
    p = np.array([2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 2, 2, 4, 1, 3, 8], np.float64)
    return rng.choice(18, size=T, p=p / p.sum()).astype(np.int32)


def record(kwargs, specs, T, max_eps, seed, scripted=None):
    """specs: list of (dialog, structure_seq); scripted: {env: action script, cycled}.  Returns the fixture dict."""
    gym, Task, Tasks = H.load_reference()
    from gridworld.tasks.task import Subtasks
    E = len(specs)
    out = dict(
        agentPos=np.zeros((E, T, 5), np.float32), inventory=np.zeros((E, T, 6), np.float32),
        compass=np.zeros((E, T), np.float32), reward=np.zeros((E, T), np.float64),
        done=np.zeros((E, T), np.uint8), internal=np.zeros((E, T, 8), np.float64),
        reset_before=np.zeros((E, T), np.uint8), episode=np.zeros((E, T), np.int32),
        syn_max_int=np.zeros((E, T), np.int32),
        ep_targets=np.zeros((E, max_eps, 9, 11, 11), np.int8), ep_starts=np.zeros((E, max_eps, 9, 11, 11), np.int8),
        ep_turn=np.full((E, max_eps, 2), -9, np.int32), ep_env_max_int=np.zeros((E, max_eps), np.int32),
        ep_reset_inventory=np.zeros((E, max_eps, 6), np.float32), ep_target_size=np.zeros((E, max_eps), np.int32),
        n_episodes=np.zeros((E,), np.int32), full_grids=np.zeros((E, 9, 11, 11), np.int8),
    )
    grids = np.zeros((E, T, 9, 11, 11), np.int8)
    actions = np.zeros((E, T), np.int32)
    rng = np.random.RandomState(seed)
    for e, (dialog, seq) in enumerate(specs):
        np.random.seed(seed * 1000 + e)          # Subtasks.reset draws the turn from the global numpy stream
        env = gym.make('IGLUGridworld-v0', vector_state=True, render=False, **kwargs)
        st = Subtasks(dialog, seq)
        env.set_task_generator(st)
        out['full_grids'][e] = np.asarray(st.full_structure)
        acts = builder_actions(rng, T)
        if scripted and e in scripted:
            acts = np.resize(np.asarray(scripted[e], np.int32), T)
        actions[e] = acts
        ep = -1

        def new_episode():
            nonlocal ep
            obs = env.reset()
            ep += 1
            assert ep < max_eps, 'raise max_eps'
            task = env.unwrapped._task
            out['ep_targets'][e, ep] = np.asarray(task.target_grid)
            out['ep_starts'][e, ep] = np.asarray(Tasks.to_dense(list(task.starting_grid)))
            out['ep_turn'][e, ep] = (st.task_start, st.task_goal)
            out['ep_env_max_int'][e, ep] = env.unwrapped.max_int
            out['ep_reset_inventory'][e, ep] = obs['inventory']
            out['ep_target_size'][e, ep] = task.target_size
            assert np.array_equal(obs['grid'], out['ep_starts'][e, ep])
            assert np.array_equal(np.asarray(task.full_grid), out['full_grids'][e])

        new_episode()
        done = False
        for t in range(T):
            if done:
                new_episode()
                out['reset_before'][e, t] = 1
            out['episode'][e, t] = ep
            obs, reward, done, _ = env.step(int(acts[t]))
            out['agentPos'][e, t] = obs['agentPos']
            out['inventory'][e, t] = obs['inventory']
            out['compass'][e, t] = obs['compass'][0]
            out['reward'][e, t] = float(reward)
            out['done'][e, t] = bool(done)
            grids[e, t] = obs['grid']
            out['internal'][e, t] = H._internals(env)
            out['syn_max_int'][e, t] = env.unwrapped._synthetic_task.max_int
        out['n_episodes'][e] = ep + 1
    # grid as a per-step change log relative to the previous step (or to the episode's starting grid)
    g = grids.reshape(E, T, -1)
    prev = np.concatenate([np.zeros((E, 1, g.shape[2]), np.int8), g[:, :-1]], axis=1)
    ar = np.arange(E)[:, None]
    ep_start = out['ep_starts'].reshape(E, max_eps, -1)[ar, out['episode']]      # [E, T, 1089]
    first = out['reset_before'].astype(bool)
    first[:, 0] = True
    prev = np.where(first[:, :, None], ep_start, prev)
    diff = g != prev
    assert diff.sum(-1).max() <= 1
    idx = np.where(diff.any(-1), diff.argmax(-1), -1).astype(np.int16)
    val = np.take_along_axis(g, np.maximum(idx, 0)[:, :, None].astype(np.int64), axis=2)[:, :, 0]
    out.update(actions=actions, grid_change_idx=idx, grid_change_val=np.where(idx >= 0, val, 0).astype(np.int8),
               grid_final=g[:, -1].reshape(E, 9, 11, 11), kwargs=json.dumps(kwargs),
               # the generator's inputs, so that a port's own Subtasks can be driven through the same episodes:
               # np.random.seed(np_seed[e]) right before Subtasks(dialog, seq) is constructed
               specs=json.dumps([dict(dialog=d, seq=[[list(map(int, b)) for b in turn] for turn in q]) for d, q in specs]),
               np_seed=np.array([seed * 1000 + e for e in range(E)], np.int64))
    return out


def main():
    goals = H.load_cdm_goals()
    names = sorted(goals, key=lambda n: int(n[1:]))
    rng = np.random.RandomState(1111)
    picks = [names[i] for i in rng.permutation(len(names))[:10]]
    specs = []
    for k, n in enumerate(picks):
        seq = structure_seq(rng, goals[n], 2 + k % 4)
        dialog = [['<Architect> turn %d of %s' % (i, n), '<Builder> ok'] for i in range(len(seq))]
        specs.append((dialog, seq))
    # a one-turn sequence (turn = -1: empty start, target = full structure)
    specs.append(([['<Architect> all of it']], [H.dense_to_sparse(goals[picks[0]])]))
    specs.append(([['<Architect> a'], ['<Architect> b']], structure_seq(rng, goals[picks[1]], 2)))
    # scripted completion: the first turn is ONE block right in front of the spawn point, the script (fall, step
    # aside, look down 45 degrees, place) builds it -> done by completion, the next reset draws another turn
    specs.append(([['<Architect> one'], ['<Architect> two']], [[(0, -1, -1, 1)], [(0, -1, -1, 1), (1, -1, -1, 1)]]))
    script = {len(specs) - 1: [0, 0, 0, 0, 1, 4] + [14] * 9 + [17] + [0] * 4}
    for tag, kw, T in (('', dict(size_reward=True, max_steps=40), 320),
                       ('_nosize', dict(size_reward=False, max_steps=50), 300)):
        fx = record(kw, specs, T, max_eps=40, seed=11 + len(tag), scripted=script)
        path = os.path.join(HERE, 's11_subtasks_env%s.npz' % tag)
        np.savez_compressed(path, **fx)
        E = len(specs)
        turns = sorted(set(tuple(v) for e in range(E) for v in fx['ep_turn'][e, :fx['n_episodes'][e]].tolist()))
        print('s11_subtasks_env%s: E=%d T=%d episodes=%d turns=%s env_max_int>0 in %d episodes, changed=%d '
              'reward_nz=%d early_done=%d -> %.0f KiB' % (
                  tag, E, T, int(fx['n_episodes'].sum()), turns, int((fx['ep_env_max_int'] > 0).sum()),
                  int((fx['grid_change_idx'] >= 0).sum()), int((fx['reward'] != 0).sum()),
                  int(fx['done'].sum()), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
