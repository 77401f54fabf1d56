"""End-to-end BC of `PolicyNetWithConv` on raw uint8 frames: same `run(flags)` contract, flags, data files and
update rule as reference main_bc_finetune.py:25-242 (model :70, data :102-128, loop :167-208).

Data parallel (BASELINE config 4 / SURVEY 8e; the reference `main` is single-GPU): under an initialised
torch.distributed group every rank draws the SAME `sample_with_minimum_distance` list (same `random.seed(run_id)`),
takes its contiguous slice of the B start indices (the LSTM keeps T whole), runs forward/backward on it, and the
flat 18.1 M-float gradient is averaged in four buckets (RCCL over xGMI with backend "nccl") that leave while backward is still
running (HipRMSprop.step_data_parallel / pvr_policy_set_data_parallel) before the clipped RMSprop update, which is then
identical on every rank.  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m pvr_habitat_amd.main_bc_finetune ...
(`__main__` reads RANK / LOCAL_RANK / WORLD_SIZE, picks the GPU and creates the process group before the first GPU call).
Interrupted runs resume, completed runs return early (main_bc_finetune.py:47-56,84-89,135-143).  Habitat evaluation needs the
simulator: pass `make_env` as in main_bc_2.run."""
import os
import pickle
import random

import numpy as np
import torch

from .arguments import make_parser
from .models import PolicyNetWithConv, HipRMSprop
from .utils_bc import is_essential_save, sample_with_minimum_distance, shard_bounds
from .test_model import test
from .dist_utils import rank_world, init_distributed, finalize_distributed


def load_raw(flags, from_env):
    """main_bc_finetune.py:102-128: per-scene `<env>.pickle` with lists of per-trajectory arrays."""
    obs = action = reward = done = None
    for env_id in from_env.split(','):
        data = pickle.load(open(os.path.join(flags.data_path, env_id + '.pickle'), 'rb'))
        n = flags.batch_size * flags.unroll_length if flags.debug else len(data['obs'])
        parts = [np.concatenate(data[k][:n]) for k in ('obs', 'action', 'reward', 'done')]
        if obs is None:
            obs, action, reward, done = parts
        else:
            obs, action = np.concatenate((obs, parts[0])), np.concatenate((action, parts[1]))
            reward, done = np.concatenate((reward, parts[2])), np.concatenate((done, parts[3]))
    assert len(obs) == len(action) == len(reward) == len(done), 'data length does not match'
    assert len(reward) > 0, 'no data found'
    return obs, action, reward, done


def run(flags, make_env=None):
    rank, world = rank_world()
    torch.manual_seed(flags.run_id)
    np.random.seed(flags.run_id)
    random.seed(flags.run_id)                                   # identical sampler stream on every rank
    from_env, to_env = flags.env, flags.to_env
    os.makedirs(flags.save_path, exist_ok=True)
    save_path = os.path.join(flags.save_path, from_env + '_emrandom_finetuned_s' + str(flags.run_id) + '_' + to_env)
    # a finished run returns early, an interrupted one resumes (main_bc_finetune.py:47-56); every rank reads the same files
    resume = False
    if os.path.isfile(save_path + '.pickle'):
        stats = pickle.load(open(save_path + '.pickle', 'rb'))
        if stats[to_env]['frames'][-1] >= flags.max_frames:
            print('   WARNING! This run was already completed. Stopping now.')
            return stats
        resume = os.path.isfile(save_path + '.tar')
    flags.device = torch.device('cuda') if torch.cuda.is_available() and not flags.disable_cuda else torch.device('cpu')
    print('=== Loading trajectories ===')
    obs, action, reward, done = load_raw(flags, from_env)
    n_samples = len(reward)
    print('  ', 'total number of samples', n_samples)
    env = None
    if make_env is not None:
        flags.env = to_env
        env = make_env(flags, None)
        obs_shape, n_actions = env.gym_env.observation_space.shape, env.gym_env.action_space.n
    else:
        obs_shape, n_actions = obs.shape[1:], int(getattr(flags, 'num_actions', 3))           # never derived from the data
    # torch's nll_loss raises for a target outside [0, A); the fused loss kernel would only turn it into a NaN loss: check once here
    assert int(np.min(action)) >= 0 and int(np.max(action)) < n_actions, \
        'actions in the data (%d..%d) do not fit num_actions=%d' % (int(np.min(action)), int(np.max(action)), n_actions)
    assert flags.batch_size % world == 0, 'batch_size must divide evenly over the ranks'
    b_lo, b_hi = shard_bounds(flags.batch_size, rank, world)
    actor_model = PolicyNetWithConv(obs_shape, n_actions, flags.batch_norm, max_unroll=flags.unroll_length,
                                    max_batch=b_hi - b_lo).to(device=flags.device)
    max_epochs = flags.max_frames // (flags.unroll_length * flags.batch_size) + 1
    optimizer = HipRMSprop(actor_model, lr=flags.learning_rate, momentum=flags.momentum, eps=flags.epsilon, alpha=flags.alpha,
                           max_grad_norm=flags.max_grad_norm, max_epochs=max_epochs)
    test_model = PolicyNetWithConv(obs_shape, n_actions, flags.batch_norm, max_unroll=1, max_batch=1).to(device=flags.device)
    test_model.eval()
    stat_keys = ['episode_return', 'episode_success']
    init_frames = 0
    if resume:                                                  # (:84-89,135-143) weights, optimizer state and schedule position
        print('=== Resuming previous run ===')
        checkpoint = torch.load(save_path + '.tar', weights_only=False, map_location='cpu')
        actor_model.load_state_dict(checkpoint['actor_model_state_dict'])
        optimizer.load_state_dict(checkpoint['actor_model_optimizer_state_dict'])
        optimizer.last_epoch = checkpoint['scheduler_state_dict']['last_epoch']
        init_frames = stats[to_env]['frames'][-1]                # (the sampler stream restarts from the seed, as in the reference)
    else:
        stats = {to_env: {**{k: [np.nan] for k in stat_keys}, 'frames': [0], 'training_loss': [np.nan], 'gradient_norm': [np.nan]}}
    print('=== Training policy ===')
    actor_model.train()
    from .bc_data import DeviceDataset
    dataset = DeviceDataset(obs, action, done, flags.device)    # raw uint8 frames resident in HBM (24.6 KB per sample)
    for frames in range(init_frames, flags.max_frames, flags.batch_size * flags.unroll_length):
        epoch = frames // (flags.batch_size * flags.unroll_length)
        starting_i = sample_with_minimum_distance(n=n_samples, k=flags.batch_size, d=flags.unroll_length)
        o, a, d = dataset.gather(starting_i[b_lo:b_hi], flags.unroll_length)     # (T, B/world, 64, 64, 6) uint8, this rank's sequences
        optimizer.scheduler_step()
        loss, gradient_norm = optimizer.step_data_parallel(o, d, a)
        if (epoch + 1) % flags.eval_frequency == 0:
            ev = {k: np.nan for k in stat_keys}
            if env is not None and ((flags.essential_save_only and is_essential_save(epoch, max_epochs, flags.eval_frequency))
                                    or not flags.essential_save_only):
                test_model.load_state_dict(actor_model.state_dict())
                ep = test(test_model, env, stat_keys, flags.n_episodes_test)
                ev = {k: float(np.mean(ep[k])) for k in stat_keys}
            for k in stat_keys:
                stats[to_env][k].append(ev[k])
            stats[to_env]['frames'].append(frames)
            stats[to_env]['training_loss'].append(float(loss))              # (synchronises)
            stats[to_env]['gradient_norm'].append(float(gradient_norm))
            actor_model.check_status()
            if rank == 0:
                print('  ', 'frames', frames, 'training loss', float(loss), 'gradient norm', float(gradient_norm))
                if not flags.disable_save:
                    pickle.dump(stats, open(save_path + '.pickle', 'wb'), protocol=pickle.HIGHEST_PROTOCOL)
                    torch.save({'actor_model_state_dict': actor_model.state_dict(),
                                'actor_model_optimizer_state_dict': optimizer.state_dict(),
                                'scheduler_state_dict': {'last_epoch': optimizer.last_epoch},
                                'flags': {k: v for k, v in vars(flags).items() if k != 'device'}}, save_path + '.tar')
    if flags.device.type == 'cuda':
        torch.cuda.synchronize()
    actor_model.check_status()
    actor_model.close()                                         # library handles are freed here, not at garbage-collection time
    test_model.close()
    if env is not None:
        env.close()
    return stats


def main(argv=None):
    flags = make_parser().parse_args(argv)
    init_distributed()                                          # device + process group first, before any GPU call
    try:
        return run(flags)
    finally:
        finalize_distributed()


if __name__ == '__main__':
    main()
