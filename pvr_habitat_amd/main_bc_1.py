"""Behavioural cloning with the embedding computed IN-PROCESS: same `run(flags)` contract as reference main_bc_1.py:25-262.

Differs from main_bc_2 only in where the observations come from: the raw per-scene pickle `<data_path>/<env>.pickle`
(utils_bc.read_habitat_data, main_bc_1.py:113) is pushed through the embedding here, scene by scene (:124-138), instead of being
read pre-embedded.  This is the only consumer of the seed-dependent 'random' PVR: "The random embedding is not pre-trained, but
randomly initialized, so its weights depend on the seed, and we need to pass the obs through it every time" (:121-122) - the
EmbeddingNet is built right after torch.manual_seed(run_id) (:27,68-72), so a run_id reproduces the reference's random network
(tests/test_gpu_glue.py::test_random_pvr_...).  Frames go through the HIP encoder in large overlapped batches
(embeddings.stream_embed) rather than batch_size x n_frames per forward (:126-132): same rows, same order."""
import os

import numpy as np
import torch

from .arguments import make_parser
from .bc_loop import train
from .embeddings import EmbeddingNet, stream_embed
from .main_bc_2 import prepare
from .save_embedded_obs import embed_rows
from .utils_bc import read_habitat_data


def embed_scene(embedding_model, obs_u8, batch_size):
    """main_bc_1.py:124-134: (N,H,W,3n) uint8 -> (N, n*O) fp32; grayscale (Atari) frames are repeated to 3 channels (:128-129)"""
    if obs_u8.shape[-1] == 1:
        obs_u8 = np.repeat(obs_u8, 3, -1)
    n_frames = max(obs_u8.shape[3] // 3, 1)
    if hasattr(getattr(embedding_model, 'embedding', None), 'forward_into'):
        return np.concatenate([stream_embed(embedding_model, obs_u8[..., 3 * f:3 * f + 3], 256)
                               for f in range(n_frames)], axis=-1)
    return embed_rows(embedding_model, obs_u8, n_frames, batch_size)


def run(flags, make_env=None):
    from_env, to_env = flags.env, flags.to_env
    save_path, stats, finished = prepare(flags)                 # seeds first: the 'random' PVR below depends on them
    if finished:
        return stats
    embedding_model = EmbeddingNet(flags.embedding_name, in_channels=3, pretrained=True, train=False, disable_cuda=flags.disable_cuda,
                                   compute_dtype=getattr(flags, 'compute_dtype', None))
    env = None
    if make_env is not None:
        flags.env = to_env
        env = make_env(flags, embedding_model)
    print('=== Loading trajectories ===')
    obs = action = reward = done = None
    for env_id in from_env.split(','):
        data = read_habitat_data(os.path.join(flags.data_path, env_id + '.pickle'))
        n_scene = flags.batch_size * flags.unroll_length if flags.debug else data['obs'].shape[0]
        print('  ', 'passing observations through embedding model')
        obs_scene = embed_scene(embedding_model, data['obs'][:n_scene], flags.batch_size)
        if obs is None:
            obs, action, reward, done = np.array(obs_scene), data['action'][:n_scene], data['reward'][:n_scene], data['done'][:n_scene]
        else:
            obs = np.concatenate((obs, obs_scene)); action = np.concatenate((action, data['action'][:n_scene]))
            reward = np.concatenate((reward, data['reward'][:n_scene])); done = np.concatenate((done, data['done'][:n_scene]))
        del data                                                # frames are not kept in memory (:116-118)
    assert len(obs) == len(action) == len(reward) == len(done), 'data length does not match'
    assert len(reward) > 0, 'no data found'
    print('  ', 'total number of samples', len(reward))
    if env is not None:
        obs_shape, n_actions = env.gym_env.observation_space.shape, env.gym_env.action_space.n
    else:
        obs_shape, n_actions = (obs.shape[1],), int(getattr(flags, 'num_actions', 3))
    return train(flags, obs, action, reward, done, save_path, to_env, stats=stats, env=env, embedding_model=embedding_model,
                 obs_shape=obs_shape, n_actions=n_actions)


if __name__ == '__main__':
    run(make_parser().parse_args())
