"""Inputs of the glue fixtures (tests/golden/make_glue_golden.py) - one definition shared by the generator, which feeds them to
the REFERENCE's code in the build container, and by the tests, which feed them to the oracle / the HIP path.  Everything here
is synthetic and regenerable (pvr_habitat_amd.synth + a fixed numpy seed); nothing is read from /root/reference."""
import os
import pickle

import numpy as np
import torch

from pvr_habitat_amd import synth

LENS = (5, 3, 4)                      # ragged trajectory lengths of the synthetic scene


def frames():
    return dict(f64=synth.smooth_frames(101, 3, 64, 64), f96=synth.smooth_frames(102, 2, 96, 128),
                f256=synth.smooth_frames(103, 2, 256, 256))


# name -> tags of frames() the fixture holds outputs for (first two frames of f64 except for resnet50)
EMBED_CASES = (('resnet50', ('f64', 'f96', 'f256')), ('resnet34', ('f64',)), ('moco_aug', ('f64',)),
               ('moco_aug_uber_345', ('f64', 'f96')), ('resnet50_places_l3', ('f64',)), ('resnet50_l4', ('f64',)))


def case_frames(name, tag):
    fr = frames()[tag]
    return fr if (name == 'resnet50' or tag != 'f64') else fr[:2]


def scene():
    """per-scene pickle of save_opt_trajectories.py:100-106: lists of per-trajectory arrays; obs (L,64,64,6) = frame | goal"""
    trajs, goals = [], []
    for t, L in enumerate(LENS):
        fr = synth.smooth_frames(200 + t, L, 64, 64)
        g = synth.smooth_frames(300 + t, 1, 64, 64)[0]
        trajs.append(np.concatenate([fr, np.broadcast_to(g, fr.shape)], axis=-1))
        goals.append(g)
    rng = np.random.RandomState(5)
    raw = dict(obs=trajs, action=[rng.randint(0, 3, L) for L in LENS], reward=[rng.rand(L) for L in LENS],
               done=[np.arange(L) == L - 1 for L in LENS], true_state=[rng.rand(L, 12) for L in LENS])
    return raw, trajs, goals


def write_scene(data_dir, env='scene'):
    """<data_dir>/<env>.pickle and the PNG tree <data_dir>/<env>/{<t>_<s>.png, <t>_goal.png, <t>.pickle}
    (save_opt_trajectories_png.py:44-58; cv2.imwrite(array) stores array[..., ::-1] as the file's RGB)."""
    from PIL import Image
    raw, trajs, goals = scene()
    os.makedirs(os.path.join(data_dir, env), exist_ok=True)
    with open(os.path.join(data_dir, env + '.pickle'), 'wb') as f:
        pickle.dump(raw, f)
    for t, L in enumerate(LENS):
        for s_ in range(L):
            Image.fromarray(np.ascontiguousarray(trajs[t][s_][..., :3][..., ::-1])).save(os.path.join(data_dir, env, '%d_%d.png' % (t, s_)))
        Image.fromarray(np.ascontiguousarray(goals[t][..., ::-1])).save(os.path.join(data_dir, env, '%d_goal.png' % t))
        with open(os.path.join(data_dir, env, '%d.pickle' % t), 'wb') as f:
            pickle.dump({k: raw[k][t] for k in ('action', 'reward', 'done', 'true_state')}, f)
    return raw, trajs, goals


class ScriptedEnv:
    """torchbeast-style environment (env.initial() / env.step(action) -> dict of (1,1) tensors) whose episode e lasts 3 + e
    steps; records every call."""

    def __init__(self, calls):
        self.t, self.ep, self.calls = 0, 0, calls

    def _out(self, done):
        return dict(done=torch.tensor([[done]]), episode_return=torch.tensor([[float(10 * self.ep + self.t)]]),
                    episode_step=torch.tensor([[self.t]]), episode_success=torch.tensor([[float(self.ep % 2)]]))

    def initial(self):
        self.calls.append('initial')
        return self._out(False)

    def step(self, action):
        self.t += 1
        done = self.t >= 3 + self.ep
        self.calls.append('step a=%d t=%d done=%d' % (int(action), self.t, done))
        o = self._out(done)
        if done:
            self.ep += 1
            self.t = 0
        return o


class ScriptedModel:
    """policy stand-in: the action is a function of the CARRIED recurrent state, so the call log shows whether the state
    is carried across episodes (it is, src/test_model.py:5-13)"""
    device = torch.device('cpu')

    def __init__(self, calls):
        self.calls = calls

    def initial_state(self, batch_size):
        self.calls.append('initial_state %d' % batch_size)
        return (torch.zeros(2, batch_size, 4), torch.zeros(2, batch_size, 4))

    def __call__(self, env_output, state):
        self.calls.append('forward state=%g' % float(state[0].sum()))
        return dict(action=torch.tensor([[int(state[0].sum()) % 3]])), (state[0] + 1, state[1])


STAT_KEYS = ['episode_return', 'episode_step', 'episode_success']
