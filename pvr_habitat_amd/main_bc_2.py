"""Behavioural cloning on pre-embedded observations: same `run(flags)` contract, flags, batch sampling,
update rule, stats dict and checkpoint files as reference main_bc_2.py:26-262; the training loop itself is bc_loop.train
(shared with main_bc_1, as the reference's two scripts share it line for line).

Evaluation in Habitat (main_bc_2.py:171-172, 230-238) needs the simulator stack (habitat-sim, gym), which is
outside this package: pass `make_env=<callable(flags, embedding_model) -> env>` to enable it; without it the
evaluation entries of the stats dict are NaN (the schema and lengths stay the reference's)."""
import os
import pickle
import random

import numpy as np
import torch

from .arguments import make_parser
from .bc_loop import train
from .test_model import test  # noqa: F401  (re-exported: main_bc_2.test is the reference's import path, main_bc_2.py:22)


def prepare(flags):
    """seeds, save path, completed-run / resume check, device (main_bc_2.py:27-66).  Returns (save_path, stats or None,
    done flag)."""
    torch.manual_seed(flags.run_id)
    np.random.seed(flags.run_id)
    random.seed(flags.run_id)
    if flags.debug:
        flags.n_episodes_test = int(np.minimum(2, flags.n_episodes_test))
    os.makedirs(flags.save_path, exist_ok=True)
    save_path = os.path.join(flags.save_path, flags.env + '_em' + flags.embedding_name + '_s' + str(flags.run_id) + '_' + flags.to_env)
    stats, finished = None, False
    if os.path.isfile(save_path + '.pickle'):
        stats = pickle.load(open(save_path + '.pickle', 'rb'))
        if stats[flags.to_env]['frames'][-1] >= flags.max_frames:
            print('   WARNING! This run was already completed. Stopping now.')
            finished = True
    flags.device = torch.device('cuda') if torch.cuda.is_available() and not flags.disable_cuda else torch.device('cpu')
    return save_path, stats, finished


def run(flags, make_env=None):
    from_env, to_env = flags.env, flags.to_env
    save_path, stats, finished = prepare(flags)
    if finished:
        return stats
    # data (main_bc_2.py:113-147)
    print('=== Loading trajectories ===')
    obs = action = reward = done = None
    for env_id in from_env.split(','):
        name = env_id + ('_resnet50' if flags.embedding_name == 'true_state' else '_' + flags.embedding_name) + '.pickle'
        data = pickle.load(open(os.path.join(flags.data_path, name), 'rb'))
        n_scene = flags.batch_size * flags.unroll_length if flags.debug else data['obs'].shape[0]
        obs_scene = (data['true_state'] if flags.embedding_name == 'true_state' else data['obs'])[:n_scene]
        if obs is None:
            obs, action, reward, done = np.array(obs_scene), data['action'][:n_scene], data['reward'][:n_scene], data['done'][:n_scene]
        else:
            obs = np.concatenate((obs, obs_scene)); action = np.concatenate((action, data['action'][:n_scene]))
            reward = np.concatenate((reward, data['reward'][:n_scene])); done = np.concatenate((done, data['done'][:n_scene]))
    assert len(obs) == len(action) == len(reward) == len(done), 'data length does not match'
    assert len(reward) > 0, 'no data found'
    print('  ', 'total number of samples', len(reward))

    env, embedding_model = None, None
    if make_env is not None:
        from .embeddings import EmbeddingNet
        embedding_model = EmbeddingNet(flags.embedding_name, in_channels=3, pretrained=True, train=False, disable_cuda=flags.disable_cuda)
        flags.env = to_env
        env = make_env(flags, embedding_model)
        obs_shape, n_actions = env.gym_env.observation_space.shape, env.gym_env.action_space.n
    else:
        obs_shape, n_actions = (obs.shape[1],), int(getattr(flags, 'num_actions', 3))      # never derived from the data
    return train(flags, obs, action, reward, done, save_path, to_env, stats=stats, env=env, embedding_model=embedding_model,
                 obs_shape=obs_shape, n_actions=n_actions)


if __name__ == '__main__':
    run(make_parser().parse_args())
