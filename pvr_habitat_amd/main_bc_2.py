"""Behavioural cloning on pre-embedded observations: same `run(flags)` contract, flags, batch sampling,
update rule, stats dict and checkpoint files as reference main_bc_2.py:26-262; the policy forward/backward/
RMSprop iteration is one fused HIP plan (models.HipRMSprop.step).

Evaluation in Habitat (main_bc_2.py:171-172, 230-238) needs the simulator stack (habitat-sim, gym), which is
outside this package: pass `make_env=<callable(flags, embedding_model) -> env>` to enable it; without it the
evaluation entries of the stats dict are NaN (the schema and lengths stay the reference's)."""
import os
import pickle
import random

import numpy as np
import torch

from .arguments import make_parser
from .models import PolicyNet, HipRMSprop
from .test_model import test
from .utils_bc import is_essential_save, sample_with_minimum_distance, gather_unrolls


def run(flags, make_env=None):
    torch.manual_seed(flags.run_id)
    np.random.seed(flags.run_id)
    random.seed(flags.run_id)
    if flags.debug:
        flags.n_episodes_test = int(np.minimum(2, flags.n_episodes_test))
    from_env, to_env = flags.env, flags.to_env
    os.makedirs(flags.save_path, exist_ok=True)
    save_path = os.path.join(flags.save_path, from_env + '_em' + flags.embedding_name + '_s' + str(flags.run_id) + '_' + to_env)

    resume = False
    if os.path.isfile(save_path + '.pickle'):
        stats = pickle.load(open(save_path + '.pickle', 'rb'))
        if stats[to_env]['frames'][-1] >= flags.max_frames:
            print('   WARNING! This run was already completed. Stopping now.')
            return stats
        resume = True
    flags.device = torch.device('cuda') if torch.cuda.is_available() and not flags.disable_cuda else torch.device('cpu')

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
    n_samples = len(reward)
    assert n_samples > 0, 'no data found'
    print('  ', 'total number of samples', n_samples)

    env, embedding_model = None, None
    if make_env is not None:
        from .embeddings import EmbeddingNet
        embedding_model = EmbeddingNet(flags.embedding_name, in_channels=3, pretrained=True, train=False, disable_cuda=flags.disable_cuda)
        flags.env = to_env
        env = make_env(flags, embedding_model)
        obs_shape, n_actions = env.gym_env.observation_space.shape, env.gym_env.action_space.n
    else:
        obs_shape, n_actions = (obs.shape[1],), int(getattr(flags, 'num_actions', 3))      # never derived from the data

    actor_model = PolicyNet(obs_shape, n_actions, flags.batch_norm, max_unroll=flags.unroll_length,
                            max_batch=flags.batch_size).to(device=flags.device)
    max_epochs = flags.max_frames // (flags.unroll_length * flags.batch_size) + 1
    optimizer = HipRMSprop(actor_model, lr=flags.learning_rate, momentum=flags.momentum, eps=flags.epsilon, alpha=flags.alpha,
                           max_grad_norm=flags.max_grad_norm, max_epochs=max_epochs)
    if resume:
        checkpoint = torch.load(save_path + '.tar', weights_only=False)
        actor_model.load_state_dict(checkpoint['actor_model_state_dict'])
        optimizer.load_state_dict(checkpoint['actor_model_optimizer_state_dict'])
        optimizer.last_epoch = checkpoint['scheduler_state_dict']['last_epoch']
    test_model = PolicyNet(obs_shape, n_actions, flags.batch_norm, max_unroll=1, max_batch=1).to(device=flags.device)
    test_model.eval()

    stat_keys = ['episode_return', 'episode_success']

    def evaluate():
        if env is None:
            return {k: np.nan for k in stat_keys}
        test_model.load_state_dict(actor_model.state_dict())
        ep = test(test_model, env, stat_keys, flags.n_episodes_test)
        return {k: float(np.mean(ep[k])) for k in stat_keys}

    if resume:
        print('=== Resuming previous run ===')
        init_frames = stats[to_env]['frames'][-1]
    else:
        print('=== Initial evaluation ===')
        stats = {to_env: {**{k: [] for k in stat_keys}, 'frames': [], 'training_loss': [], 'gradient_norm': []}}
        for k, v in evaluate().items():
            stats[to_env][k].append(v)
        stats[to_env]['frames'].append(0)
        stats[to_env]['training_loss'].append(np.nan)
        stats[to_env]['gradient_norm'].append(np.nan)
        init_frames = 0

    print('=== Training policy ===')
    actor_model.train()
    # torch's nll_loss raises for a target outside [0, A); the fused loss kernel only turns it into a NaN loss: check the data once
    assert int(np.min(action)) >= 0 and int(np.max(action)) < n_actions, \
        'actions in the data (%d..%d) do not fit num_actions=%d' % (int(np.min(action)), int(np.max(action)), n_actions)
    from .bc_data import DeviceDataset
    dataset = DeviceDataset(obs, action, done, flags.device)    # resident in HBM; every (T,B) batch is gathered there (pvr_bc_gather)
    for frames in range(init_frames, flags.max_frames, flags.batch_size * flags.unroll_length):
        epoch = frames // (flags.batch_size * flags.unroll_length)
        starting_i = sample_with_minimum_distance(n=n_samples, k=flags.batch_size, d=flags.unroll_length)
        o, a, d = dataset.gather(starting_i, flags.unroll_length)   # (T,B,obs) == np.stack(..., axis=1) of main_bc_2.py:194-201
        optimizer.scheduler_step()                                # precedes the update (main_bc_2.py:216)
        loss, gradient_norm = optimizer.step(o, d, a)
        if (epoch + 1) % flags.eval_frequency == 0:
            if (flags.essential_save_only and is_essential_save(epoch, max_epochs, flags.eval_frequency)) or not flags.essential_save_only:
                ev = evaluate()
            else:
                ev = {k: np.nan for k in stat_keys}
            for k in stat_keys:
                stats[to_env][k].append(ev[k])
            stats[to_env]['frames'].append(frames)
            stats[to_env]['training_loss'].append(float(loss))
            stats[to_env]['gradient_norm'].append(float(gradient_norm))
            print('  ', 'frames', frames, 'training loss', float(loss), 'gradient norm', float(gradient_norm))
            if not flags.disable_save:
                pickle.dump(stats, open(save_path + '.pickle', 'wb'), protocol=pickle.HIGHEST_PROTOCOL)
                torch.save({'embedding_model_state_dict': embedding_model.state_dict() if embedding_model is not None else {},
                            'actor_model_state_dict': actor_model.state_dict(),
                            'actor_model_optimizer_state_dict': optimizer.state_dict(),
                            'scheduler_state_dict': {'last_epoch': optimizer.last_epoch},
                            'flags': {k: v for k, v in vars(flags).items() if k != 'device'}}, save_path + '.tar')
    if env is not None:
        env.close()
    return stats


if __name__ == '__main__':
    run(make_parser().parse_args())
