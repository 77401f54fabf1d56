"""CPU fp32 restatement of the reference BC policy and its training step (test infrastructure).

Follows:
  * reference src/models.py:13-89    PolicyNet: [BatchNorm1d] -> Linear+ReLU -> Linear+ReLU ->
                                     2-layer LSTM stepped one timestep at a time with the state
                                     multiplied by (1-done) (:66-72) -> policy / baseline heads ->
                                     argmax in eval (:82)
  * reference src/models.py:96-197   PolicyNetWithConv: x/255, per-frame transpose(1,3) (W<->H swap),
                                     5x[conv3x3 s2 p1 + ELU], cat on last axis, view(T*B,-1)
  * reference main_bc_2.py:80-90     RMSprop(lr, alpha, eps, momentum=0) + LambdaLR(1-epoch/max_epochs)
  * reference main_bc_2.py:209-227   nll_loss(log_softmax) mean; scheduler.step() BEFORE optimizer.step();
                                     sum of squared grad norms; clip_grad_norm_(40); RMSprop step
Third-party semantics restated (torch==1.9 nn.LSTM gate order i,f,g,o; BatchNorm1d momentum 0.1,
eps 1e-5, unbiased running_var; clip coef = max_norm/(norm+1e-6) clamped to 1; RMSprop
v=alpha*v+(1-alpha)*g^2, p-=lr*g/(sqrt(v)+eps)).

PARITY PINNING: tests/golden/make_golden.py imports the reference's own src/models.py and
torch.optim.RMSprop / LambdaLR / clip_grad_norm_ in the build container and stores their outputs
in tests/golden/policy_*.npz; tests/test_oracle_policy.py checks this file against them.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def to_params(sd):
    """numpy state_dict -> dict of fp32 leaf tensors (buffers stay plain tensors)."""
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.array(v, copy=True))
        if t.dtype == torch.float32 and not k.endswith(('running_mean', 'running_var')):
            t.requires_grad_(True)
        out[k] = t
    return out


def _lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def conv_features(p, obs_u8):
    """PolicyNetWithConv feature path (models.py:159-170). obs (T,B,H,W,3n) uint8 -> (T*B, 128n)."""
    T, B = obs_u8.shape[:2]
    x = torch.flatten(obs_u8, 0, 1).float() / 255.0
    feats = []
    for xi in torch.split(x, 3, -1):
        y = xi.transpose(1, 3)                       # (N,3,W,H): spatially transposed, as the reference
        for i in range(5):
            y = F.elu(F.conv2d(y, p['feat_extract.%d.weight' % (2 * i)], p['feat_extract.%d.bias' % (2 * i)], 2, 1))
        feats.append(y)
    return torch.cat(feats, -1).reshape(T * B, -1)


def forward(p, obs, done, state, batch_norm, training=True, conv=False):
    """Returns dict(policy_logits (T,B,A), baseline (T,B), action (T,B) argmax), new state.

    `p` is mutated like the module would be: BN running stats are updated when training."""
    T, B = obs.shape[:2]
    x = conv_features(p, obs) if conv else torch.flatten(obs, 0, 1).float()
    o = 0
    if batch_norm:
        if training:
            mean = x.mean(0)
            var_b = x.var(0, unbiased=False)
            n = x.shape[0]
            with torch.no_grad():
                p['fc.0.running_mean'].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
                p['fc.0.running_var'].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var_b.detach() * n / (n - 1))
                p['fc.0.num_batches_tracked'] += 1
        else:
            mean, var_b = p['fc.0.running_mean'], p['fc.0.running_var']
        x = (x - mean) / torch.sqrt(var_b + BN_EPS) * p['fc.0.weight'] + p['fc.0.bias']
        o = 1
    x = F.relu(x @ p['fc.%d.weight' % o].t() + p['fc.%d.bias' % o])
    x = F.relu(x @ p['fc.%d.weight' % (o + 2)].t() + p['fc.%d.bias' % (o + 2)])
    core_in = x.view(T, B, -1)
    notdone = (1 - done.float()).abs()
    h, c = state
    outs = []
    for t in range(T):
        nd = notdone[t].view(1, -1, 1)
        h, c = nd * h, nd * c                         # models.py:69
        hs, cs = [], []
        inp = core_in[t]
        for l in range(2):
            h2, c2 = _lstm_cell(inp, h[l], c[l], p['core.weight_ih_l%d' % l], p['core.weight_hh_l%d' % l],
                                p['core.bias_ih_l%d' % l], p['core.bias_hh_l%d' % l])
            hs.append(h2); cs.append(c2); inp = h2
        h, c = torch.stack(hs), torch.stack(cs)
        outs.append(inp)
    core_out = torch.cat(outs, 0)                     # (T*B, H), time-major
    logits = core_out @ p['policy.weight'].t() + p['policy.bias']
    baseline = core_out @ p['baseline.weight'].t() + p['baseline.bias']
    action = torch.argmax(logits, dim=1)              # eval branch (models.py:82)
    return dict(policy_logits=logits.view(T, B, -1), baseline=baseline.view(T, B),
                action=action.view(T, B)), (h, c)


class RMSpropState:
    """torch.optim.RMSprop(momentum=0, centered=False) + LambdaLR(1 - epoch/max_epochs)."""

    def __init__(self, p, lr=1e-4, alpha=0.99, eps=1e-5, max_epochs=1, max_grad_norm=40.0):
        self.lr0, self.alpha, self.eps = lr, alpha, eps
        self.max_epochs, self.max_grad_norm = max_epochs, max_grad_norm
        self.epoch = 0                                # LambdaLR.last_epoch
        self.sq = {k: torch.zeros_like(v) for k, v in p.items() if v.requires_grad}


def bc_step(p, opt, obs, done, actions, batch_norm, conv=False):
    """One iteration of main_bc_2.py:206-227.  Returns (loss, grad_norm, logits)."""
    B = obs.shape[1]
    H = p['core.weight_hh_l0'].shape[1]
    state = (torch.zeros(2, B, H), torch.zeros(2, B, H))
    for v in p.values():
        if v.requires_grad:
            v.grad = None
    out, _ = forward(p, obs, done, state, batch_norm, training=True, conv=conv)
    logits = out['policy_logits']
    loss = F.nll_loss(F.log_softmax(torch.flatten(logits, 0, 1), dim=-1), torch.flatten(actions, 0, 1).long())
    opt.epoch += 1                                    # scheduler.step() precedes optimizer.step()
    lr = opt.lr0 * (1 - opt.epoch / opt.max_epochs)
    loss.backward()
    grads = {k: v.grad for k, v in p.items() if v.requires_grad and v.grad is not None}
    gn = float(np.sqrt(sum(float(g.norm(2)) ** 2 for g in grads.values())))
    total = torch.sqrt(sum((g.norm(2) ** 2 for g in grads.values())))
    coef = torch.clamp(opt.max_grad_norm / (total + 1e-6), max=1.0)
    with torch.no_grad():
        for k, g in grads.items():
            g = g * coef
            opt.sq[k].mul_(opt.alpha).addcmul_(g, g, value=1 - opt.alpha)
            p[k].addcdiv_(g, opt.sq[k].sqrt().add_(opt.eps), value=-lr)
    return float(loss.detach()), gn, logits.detach()


# reference src/utils_bc.py:17-29
def ranks(sample):
    indices = sorted(range(len(sample)), key=lambda i: sample[i])
    return sorted(indices, key=lambda i: indices[i])


def sample_with_minimum_distance(rng, n=40, k=4, d=10):
    sample = rng.sample(range(n - (k - 1) * (d - 1)), k)
    return [s + (d - 1) * r for s, r in zip(sample, ranks(sample))]
