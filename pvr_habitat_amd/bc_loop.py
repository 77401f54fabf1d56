"""The BC training loop shared by main_bc_1.run and main_bc_2.run (the reference repeats it verbatim in main_bc_1.py:80-262 and
main_bc_2.py:78-262): model / optimiser / scheduler set-up, resume, initial evaluation, the iteration of :186-227, periodic
evaluation + stats + checkpoint (:229-260).

Two forms of the iteration, selected by `--autograd_step` (same numbers, tests/test_gpu_policy.py):
  default          `optimizer.step(o, d, a)` - the whole iteration as one enqueue of HIP launches (HipRMSprop / HipAdam);
  --autograd_step  the reference's own lines through the autograd bridge: model(...) -> F.nll_loss(F.log_softmax(...)) ->
                   scheduler.step() -> optimizer.zero_grad() -> loss.backward() -> sum of squared grad norms ->
                   nn.utils.clip_grad_norm_ -> torch.optim.RMSprop.step()          (main_bc_2.py:206-227, unchanged)
The dataset lives in HBM and every (T, B) batch is gathered there (bc_data.DeviceDataset)."""
import pickle

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from .bc_data import DeviceDataset
from .models import PolicyNet, make_optimizer
from .test_model import test
from .utils_bc import is_essential_save, sample_with_minimum_distance


def train(flags, obs, action, reward, done, save_path, to_env, stats=None, env=None, embedding_model=None, obs_shape=None,
          n_actions=None, policy_cls=PolicyNet):
    """stats: the unpickled stats of an interrupted run (resume) or None.  Returns the stats dict."""
    resume = stats is not None
    n_samples = len(reward)
    # torch's nll_loss raises for a target outside [0, A); the fused loss kernel only turns it into a NaN loss: check the data once
    assert int(np.min(action)) >= 0 and int(np.max(action)) < n_actions, \
        'actions in the data (%d..%d) do not fit num_actions=%d' % (int(np.min(action)), int(np.max(action)), n_actions)
    actor_model = policy_cls(obs_shape, n_actions, flags.batch_norm, max_unroll=flags.unroll_length,
                             max_batch=flags.batch_size).to(device=flags.device)
    host = bool(getattr(flags, 'disable_cuda', False))            # --disable_cuda: the library's host backend (BASELINE configs[0]); else a CPU device fails loudly
    if hasattr(actor_model, 'use_host_backend'):
        actor_model.use_host_backend(host)
    max_epochs = flags.max_frames // (flags.unroll_length * flags.batch_size) + 1
    autograd_step = bool(getattr(flags, 'autograd_step', False))
    if autograd_step:
        # main_bc_2.py:80-90, as written
        optimizer = torch.optim.RMSprop(actor_model.parameters(), lr=flags.learning_rate, momentum=flags.momentum,
                                        eps=flags.epsilon, alpha=flags.alpha)

        def lr_lambda(epoch):
            return 1 - epoch / max_epochs
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)
    else:
        optimizer = make_optimizer(flags, actor_model, max_epochs)
        scheduler = None
    if resume:
        checkpoint = torch.load(save_path + '.tar', weights_only=False, map_location='cpu')
        if embedding_model is not None and checkpoint.get('embedding_model_state_dict'):
            embedding_model.load_state_dict(checkpoint['embedding_model_state_dict'])
        actor_model.load_state_dict(checkpoint['actor_model_state_dict'])
        optimizer.load_state_dict(checkpoint['actor_model_optimizer_state_dict'])
        if scheduler is not None:
            scheduler.load_state_dict(checkpoint['scheduler_state_dict'])
        else:
            optimizer.last_epoch = checkpoint['scheduler_state_dict']['last_epoch']
    test_model = policy_cls(obs_shape, n_actions, flags.batch_norm, max_unroll=1, max_batch=1).to(device=flags.device)
    if hasattr(test_model, 'use_host_backend'):
        test_model.use_host_backend(host)
    test_model.eval()
    stat_keys = ['episode_return', 'episode_success']

    def evaluate():
        if env is None:                                         # no simulator attached: the schema and lengths stay the reference's
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
    dataset = DeviceDataset(obs, action, done, flags.device)    # resident in HBM; every (T,B) batch is gathered there (pvr_bc_gather)
    initial_agent_state = actor_model.initial_state(batch_size=flags.batch_size)
    for frames in range(init_frames, flags.max_frames, flags.batch_size * flags.unroll_length):
        epoch = frames // (flags.batch_size * flags.unroll_length)
        starting_i = sample_with_minimum_distance(n=n_samples, k=flags.batch_size, d=flags.unroll_length)
        o, a, d = dataset.gather(starting_i, flags.unroll_length)   # (T,B,obs) == np.stack(..., axis=1) of main_bc_2.py:194-201
        if autograd_step:
            # ---- main_bc_2.py:206-227, as written ----
            output, _ = actor_model(dict(obs=o, done=d), initial_agent_state)
            loss = F.nll_loss(F.log_softmax(torch.flatten(output['policy_logits'], 0, 1), dim=-1), target=torch.flatten(a, 0, 1).long())
            scheduler.step()
            optimizer.zero_grad()
            loss.backward()
            gradient_norm = 0.
            for p in actor_model.parameters():
                if p.grad is not None and p.requires_grad:
                    gradient_norm += p.grad.detach().data.norm(2).item() ** 2
            gradient_norm = gradient_norm ** 0.5
            nn.utils.clip_grad_norm_(actor_model.parameters(), flags.max_grad_norm)
            optimizer.step()
        else:
            optimizer.scheduler_step()                            # precedes the update (main_bc_2.py:216)
            loss, gradient_norm = optimizer.step(o, d, a)
        if (epoch + 1) % flags.eval_frequency == 0:
            if (flags.essential_save_only and is_essential_save(epoch, max_epochs, flags.eval_frequency)) or not flags.essential_save_only:
                ev = evaluate()
            else:
                ev = {k: np.nan for k in stat_keys}
            for k in stat_keys:
                stats[to_env][k].append(ev[k])
            stats[to_env]['frames'].append(frames)
            stats[to_env]['training_loss'].append(float(loss))              # (synchronises)
            stats[to_env]['gradient_norm'].append(float(gradient_norm))
            actor_model.check_status()                                      # a persistent launch that gave up raises here, not as a silent NaN
            print('  ', 'frames', frames, 'training loss', float(loss), 'gradient norm', float(gradient_norm))
            if not flags.disable_save:
                pickle.dump(stats, open(save_path + '.pickle', 'wb'), protocol=pickle.HIGHEST_PROTOCOL)
                sched_sd = scheduler.state_dict() if scheduler is not None else {'last_epoch': optimizer.last_epoch}
                torch.save({'embedding_model_state_dict': embedding_model.state_dict() if embedding_model is not None else {},
                            'actor_model_state_dict': actor_model.state_dict(),
                            'actor_model_optimizer_state_dict': optimizer.state_dict(),
                            'scheduler_state_dict': sched_sd,
                            'flags': {k: v for k, v in vars(flags).items() if k != 'device'}}, save_path + '.tar')
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    actor_model.check_status()
    actor_model.close()                                         # library handles are freed here, not at garbage-collection time
    test_model.close()
    if env is not None:
        env.close()
    return stats
