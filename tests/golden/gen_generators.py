#!/usr/bin/env python3
"""Golden vectors for task GENERATORS driving env.reset(): the
reference's Subtasks / RandomTasks / CustomTasks under a fixed np.random seed, stepped through
gym.make(...).set_task_generator().  Records, per reset, the sampled task (target, start, full grid)
and the whole trajectory.  Build container only (needs /root/reference)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402


def sparse_sorted(dense):
    return H.dense_to_sparse(dense)


def make_generators(which, goals):
    gym, Task, Tasks = H.load_reference()
    from gridworld.tasks.task import Subtasks
    from gridworld.tasks.task_set import CustomTasks, RandomTasks
    if which == 'subtasks':
        blocks = sparse_sorted(goals['C3'])
        cuts = [len(blocks) * k // 4 for k in range(1, 5)]
        seq = [blocks[:c] for c in cuts]
        dialog = [['<Architect> step %d' % k, '<Builder> ok'] for k in range(4)]
        return dict(kind='subtasks', dialog=dialog, seq=seq), lambda: Subtasks(dialog, seq)
    if which == 'random':
        kw = dict(max_blocks=6, height_levels=2, max_dist=2, num_colors=3)
        return dict(kind='random', **kw), lambda: RandomTasks(**kw)
    gl = [('goal %d' % k, sparse_sorted(goals[n])) for k, n in enumerate(('C8', 'C12', 'C17'))]
    # starting_grid=[] : with the default None the reference crashes in step() (env.py:290)
    return dict(kind='custom', goals=gl), lambda: CustomTasks(gl, task_kwargs={'starting_grid': []})


def run(which, goals, seed, T, max_steps):
    gym, Task, Tasks = H.load_reference()
    spec, ctor = make_generators(which, goals)
    np.random.seed(seed)
    gen = ctor()
    env = gym.make('IGLUGridworld-v0', vector_state=True, render=False, size_reward=False, max_steps=max_steps)
    env.set_task_generator(gen)
    rng = np.random.RandomState(seed + 1)  # actions come from a private stream, tasks from the global one
    acts = rng.choice(18, size=T, p=np.array([1] * 14 + [5] + [1] * 3) / 22.0)  # looks down more often
    out = dict(actions=acts.astype(np.int32), agentPos=np.zeros((T, 5), np.float32), reward=np.zeros(T),
               done=np.zeros(T, np.uint8), inventory=np.zeros((T, 6), np.float32),
               grid_final=None, reset_before=np.zeros(T, np.uint8), task_targets=[], task_starts=[], task_fulls=[],
               task_chats=[])

    def note_task():
        t = env.unwrapped._task
        cur = getattr(t, 'current', t)
        out['task_targets'].append(np.asarray(cur.target_grid, np.int8))
        st = np.zeros((9, 11, 11), np.int8)
        for x, y, z, c in (cur.starting_grid or []):
            st[y + 1, x + 5, z + 5] = c
        out['task_starts'].append(st)
        out['task_fulls'].append(np.zeros((9, 11, 11), np.int8) if cur.full_grid is None
                                 else np.asarray(cur.full_grid, np.int8))
        out['task_chats'].append(cur.chat)

    env.reset()
    note_task()
    done = False
    for t in range(T):
        if done:
            env.reset()
            note_task()
            out['reset_before'][t] = 1
        obs, r, done, _ = env.step(int(acts[t]))
        out['agentPos'][t], out['reward'][t], out['done'][t] = obs['agentPos'], float(r), done
        out['inventory'][t] = obs['inventory']
    out['grid_final'] = obs['grid'].astype(np.int8)
    for k in ('task_targets', 'task_starts', 'task_fulls'):
        out[k] = np.stack(out[k])
    out['task_chats'] = np.array(out['task_chats'])
    return spec, out


def main():
    import json
    goals = H.load_cdm_goals()
    H.load_reference()
    d = {}
    # RandomTasks cannot be stepped in the reference (its tasks have starting_grid=None -> TypeError in
    # step()), so only its sampling stream is recorded: targets of 40 consecutive reset()s
    from gridworld.tasks.task_set import RandomTasks
    for tag, kw in (('random_a', dict(max_blocks=6, height_levels=2, max_dist=2, num_colors=3)),
                    ('random_b', dict(max_blocks=20, height_levels=1, max_dist=2, num_colors=6, max_cache=5))):
        np.random.seed(12)
        ok = False
        while not ok:  # the reference loops forever when the window has too few free cells
            st = np.random.get_state()
            try:
                import signal
                signal.signal(signal.SIGALRM, lambda *a: (_ for _ in ()).throw(TimeoutError()))
                signal.alarm(5)
                gen = RandomTasks(**kw)
                tg = [np.asarray(gen.reset().target_grid, np.int8) for _ in range(40)]
                signal.alarm(0)
                ok = True
            except TimeoutError:
                np.random.set_state(st)
                np.random.randint(10)  # skip ahead and retry
        d[tag + '_spec'] = json.dumps(kw)
        d[tag + '_state0'] = np.array(st[1], np.uint32)
        d[tag + '_pos0'] = st[2]
        d[tag + '_targets'] = np.stack(tg)
        print(tag, 'distinct', len({t.tobytes() for t in tg}))
    for which, seed in (('subtasks', 11), ('custom', 13)):
        spec, out = run(which, goals, seed, T=500, max_steps=100)
        d[which + '_spec'] = json.dumps(spec)
        d[which + '_seed'] = seed
        for k, v in out.items():
            d[which + '_' + k] = v
        print(which, 'resets', int(out['reset_before'].sum()), 'distinct targets',
              len({t.tobytes() for t in out['task_targets']}), 'reward_nz', int((out['reward'] != 0).sum()))
    path = os.path.join(HERE, 's7_generators.npz')
    np.savez_compressed(path, **d)
    print('->', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
