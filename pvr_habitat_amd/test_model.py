"""Greedy evaluation rollouts with the reference's episode / state semantics (src/test_model.py:4-22).

The environment is initialised ONCE and the recurrent state is created ONCE: both are carried across episode boundaries
(the policy zeroes its own state where `done` is set, src/models.py:66-72, and the torchbeast-style environment resets
itself inside `step`), so episode e+1 starts from the env_output that ended episode e.  One T=1, B=1 policy forward per
environment step; with `EmbeddingWrapper` in the environment every step also embeds the observation's frames on the GPU.
`tests/golden/glue_save_obs.npz` (test/*) holds the call order and statistics the reference's function produces on a
scripted environment; tests/test_glue_golden.py replays them through this one."""
import torch


def test(model, env, stat_keys, n_episodes=100):
    env_output = env.initial()
    agent_state = tuple(s.to(device=model.device) for s in model.initial_state(batch_size=1))
    stats = {k: [] for k in stat_keys}
    for _ in range(n_episodes):
        done = False
        while not done:
            with torch.no_grad():
                agent_output, agent_state = model(env_output, agent_state)
            env_output = env.step(agent_output['action'])
            done = bool(env_output['done'])
        for k in stat_keys:
            stats[k].append(float(env_output[k].numpy()[0][0]))
    return stats
