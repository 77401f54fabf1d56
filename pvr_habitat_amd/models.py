"""PolicyNet with the reference call surface; forward and the BC training step run in libpvr_hip.so.

Mirrors reference src/models.py:13-89 (`PolicyNet(observation_shape, num_actions, batch_norm=False)`,
`.device`, `.initial_state(B)`, `.forward(inputs, core_state) -> (dict, core_state)`), the optimiser set-up of
main_bc_2.py:80-90 (`RMSprop` + `LambdaLR(1 - epoch/max_epochs)`) and the update of main_bc_2.py:206-227.

All parameters are views of ONE flat fp32 buffer (layout from `pvr_policy_param_offset`), so the fused
HIP step (forward, loss, BPTT, grad-norm, clip, RMSprop) updates them in place while `state_dict()` keeps
the reference's keys and shapes (`fc.0.*` BatchNorm, `fc.1/fc.3` linears, `core.*_l{0,1}`, `policy.*`,
`baseline.*`).  Two ways to train, same arithmetic:
  * `HipRMSprop.step(obs, done, actions)`: the whole iteration of main_bc_2.py:206-227 as ONE enqueue (no autograd graph);
  * the reference's own lines, unchanged: in training mode with grad enabled `forward` returns `policy_logits` attached to a
    `torch.autograd.Function` whose backward is `pvr_policy_backward_dlogits`; it hands every parameter a view of one flat
    gradient buffer, so `loss.backward(); nn.utils.clip_grad_norm_(model.parameters(), 40); torch.optim.RMSprop(...).step()`
    (main_bc_2.py:209-227) run as written and update the flat buffer through the parameter views.
"""
import ctypes as C
import os
import math

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import _lib

HIDDEN = 1024


class PolicyBN(C.Structure):
    _fields_ = [('running_mean', C.c_void_p), ('running_var', C.c_void_p), ('num_batches_tracked', C.c_void_p)]


class PolicyDesc(C.Structure):
    _fields_ = [('obs_size', C.c_int32), ('hidden', C.c_int32), ('num_actions', C.c_int32), ('batch_norm', C.c_int32),
                ('max_t', C.c_int32), ('max_b', C.c_int32), ('conv_frames', C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)     # pvr_allreduce_fn (include/pvr_policy.h)


def _plib():
    L = _lib.lib()
    if not getattr(L, '_policy_bound', False):
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        L.pvr_policy_create.restype = C.c_int
        L.pvr_policy_create.argtypes = [C.POINTER(PolicyDesc), C.POINTER(vp)]
        L.pvr_policy_create_host.restype = C.c_int
        L.pvr_policy_create_host.argtypes = [C.POINTER(PolicyDesc), C.POINTER(vp)]
        L.pvr_policy_destroy.restype = None
        L.pvr_policy_destroy.argtypes = [vp]
        L.pvr_policy_param_count.restype = i64
        L.pvr_policy_param_count.argtypes = [vp]
        L.pvr_policy_trainable_count.restype = i64
        L.pvr_policy_trainable_count.argtypes = [vp]
        L.pvr_policy_param_offset.restype = i64
        L.pvr_policy_param_offset.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
        L.pvr_policy_forward.restype = C.c_int
        L.pvr_policy_forward.argtypes = [vp, vp, C.POINTER(PolicyBN), vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]
        L.pvr_policy_step.restype = C.c_int
        L.pvr_policy_step.argtypes = [vp, vp, vp, C.POINTER(PolicyBN), vp, vp, vp, i32, i32, f32, f32, f32, f32, vp, vp, vp]
        L.pvr_policy_backward.restype = C.c_int
        L.pvr_policy_backward.argtypes = [vp, vp, C.POINTER(PolicyBN), vp, vp, vp, i32, i32, vp, vp, vp, vp]
        L.pvr_policy_apply.restype = C.c_int
        L.pvr_policy_apply.argtypes = [vp, vp, vp, vp, f32, f32, f32, f32, vp, vp]
        L.pvr_policy_set_data_parallel.restype = C.c_int
        L.pvr_policy_set_data_parallel.argtypes = [vp, i32, i32, ALLREDUCE_FN, vp]
        L.pvr_policy_backward_dlogits.restype = C.c_int
        L.pvr_policy_backward_dlogits.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
        L.pvr_policy_apply_momentum.restype = C.c_int
        L.pvr_policy_apply_momentum.argtypes = [vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, vp, vp]
        L.pvr_policy_apply_adam.restype = C.c_int
        L.pvr_policy_apply_adam.argtypes = [vp, vp, vp, vp, vp, f32, f32, f32, f32, i64, f32, vp, vp]
        L.pvr_policy_last_grads.restype = C.c_int
        L.pvr_policy_last_grads.argtypes = [vp, vp, vp]
        L.pvr_policy_set_action_sampling.restype = C.c_int
        L.pvr_policy_set_action_sampling.argtypes = [vp, C.c_int32, C.c_uint64]
        L.pvr_policy_action_sampling_call.restype = C.c_uint64
        L.pvr_policy_action_sampling_call.argtypes = [vp]
        L.pvr_policy_set_action_sampling_call.restype = C.c_int
        L.pvr_policy_set_action_sampling_call.argtypes = [vp, C.c_uint64]
        L.pvr_policy_status.restype = C.c_int
        L.pvr_policy_status.argtypes = [vp]
        L.pvr_policy_recurrence_mode.restype = i32
        L.pvr_policy_recurrence_mode.argtypes = [vp]
        L.pvr_policy_debug_drop_block.restype = C.c_int
        L.pvr_policy_debug_drop_block.argtypes = [vp, i32]
        L.pvr_debug_set_gemm_mode.restype = C.c_int
        L.pvr_debug_set_gemm_mode.argtypes = [i32]
        L.pvr_op_gemm_f32.restype = C.c_int
        L.pvr_op_gemm_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
        L._policy_bound = True
    return L


class _Holder(nn.Module):
    """Plain container so parameter names nest like the reference modules (fc.1.weight, core.weight_ih_l0)."""


def _init(module, weight_init, bias_init, gain=1):
    weight_init(module.weight.data, gain=gain)
    bias_init(module.bias.data)
    return module


def _reference_init(obs_size, num_actions, batch_norm, hidden, conv=False):
    """Same torch modules, construction order and init calls as reference src/models.py:17-44 (and :104-148 for the
    conv variant), so that a given torch seed yields the reference's initial weights bit for bit.
    Returns {state_dict key: tensor}."""
    init_ = lambda m: _init(m, nn.init.orthogonal_, lambda x: nn.init.constant_(x, 0), nn.init.calculate_gain('relu'))
    feat = None
    if conv:
        layers, cin = [], 3
        for _ in range(5):
            layers += [init_(nn.Conv2d(in_channels=cin, out_channels=32, kernel_size=(3, 3), stride=2, padding=1)), nn.ELU()]
            cin = 32
        feat = nn.Sequential(*layers)
    fc = nn.Sequential(init_(nn.Linear(obs_size, hidden)), nn.ReLU(), init_(nn.Linear(hidden, hidden)), nn.ReLU())
    if batch_norm:
        fc = nn.Sequential(nn.BatchNorm1d(obs_size), *list(fc))
    core = nn.LSTM(hidden, hidden, 2)
    init_ = lambda m: _init(m, nn.init.orthogonal_, lambda x: nn.init.constant_(x, 0))
    policy = init_(nn.Linear(hidden, num_actions))
    baseline = init_(nn.Linear(hidden, 1))
    sd = {}
    for pfx, mod in ((('feat_extract', feat),) if conv else ()) + (('fc', fc), ('core', core), ('policy', policy), ('baseline', baseline)):
        for k, v in mod.state_dict().items():
            sd[pfx + '.' + k] = v.detach().clone()
    return sd


class _PolicyFunction(torch.autograd.Function):
    """Training-mode forward of the whole policy as one autograd node.  Inputs: the model, the prepared device tensors, then every
    trainable parameter (so autograd routes a gradient to each of them); outputs: logits (differentiable), baseline / action /
    final state (not differentiable: the BC loss reads only the logits, main_bc_2.py:211-214)."""

    @staticmethod
    def forward(ctx, model, x, done, h0, c0, T, B, *params):
        out = model._forward_raw(x, done, h0, c0, T, B, training=True)
        model._fwd_gen = getattr(model, '_fwd_gen', 0) + 1           # the workspace holds the activations of THIS forward only
        ctx.model, ctx.x, ctx.T, ctx.B, ctx.gen = model, x, T, B, model._fwd_gen
        ctx.mark_non_differentiable(*out[1:])
        return out

    @staticmethod
    def backward(ctx, dlogits, *unused):
        m = ctx.model
        if ctx.gen != getattr(m, '_fwd_gen', 0):
            raise RuntimeError('PolicyNet backward: the activations of this forward are gone - a later training-mode pvr_policy_forward '
                               'of the same policy overwrote its workspace (one backward per forward, right after it, as in the BC loop)')
        g = torch.empty(m._n_train, dtype=torch.float32, device=m.device)
        vp = lambda t: C.c_void_p(t.data_ptr())
        status = _plib().pvr_policy_backward_dlogits(m._handle, vp(m._flat), vp(ctx.x), vp(dlogits.contiguous().float()), ctx.T, ctx.B,
                                                     vp(g), _lib.stream_ptr())
        m._checked(status)
        m._last_flat_grad = g
        grads = []
        for k in m._order:
            o, shp = m._slots[k]
            grads.append(g[o:o + int(np.prod(shp))].view(shp) if o < m._n_train else None)     # baseline head: no gradient
        return (None,) * 7 + tuple(grads)


class PolicyNet(nn.Module):
    _conv_frames = 0
    _stats_groups = {}              # ranks of the gradient process group (None = the world) -> the SyncBN statistics group over them (one per process)

    def __init__(self, observation_shape, num_actions, batch_norm=False, max_unroll=100, max_batch=32):
        super(PolicyNet, self).__init__()
        if self._conv_frames_from(observation_shape):
            self._conv_frames = self._conv_frames_from(observation_shape)
            assert tuple(observation_shape[:2]) == (64, 64), 'PolicyNetWithConv is built for 64x64 frames (habitat_config/nav_task.yaml:11-12)'
            self.obs_size = 128 * self._conv_frames      # 32 channels x 2 x 2 per frame (models.py:120-121)
        else:
            self.obs_size = int(observation_shape[0])
        self.num_actions = int(num_actions)
        self.batch_norm = bool(batch_norm)
        self.hidden = HIDDEN
        self._max_t, self._max_b = int(max_unroll), int(max_batch)
        self._handle = None
        self._layout = None          # name -> (offset, shape), filled from the library on first GPU use
        sd = _reference_init(self.obs_size, self.num_actions, self.batch_norm, self.hidden, conv=self._conv_frames > 0)
        # flat buffer with the library's layout (computed here without the GPU: same rule as policy.hip add_slot)
        self._order = [k for k in sd if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]
        base = [k for k in self._order if k.startswith('baseline.')]
        self._order = [k for k in self._order if not k.startswith('baseline.')] + base
        off, self._slots = 0, {}
        for k in self._order:
            self._slots[k] = (off, tuple(sd[k].shape))
            if k == 'policy.bias':
                self._n_train = off + (sd[k].numel() + 3) // 4 * 4
            off += (sd[k].numel() + 3) // 4 * 4
        self._flat = torch.zeros(off, dtype=torch.float32)
        for k in self._order:
            o, shp = self._slots[k]
            self._flat[o:o + sd[k].numel()].copy_(sd[k].reshape(-1))
        # module tree with the reference names; parameters are views of the flat buffer
        for k in self._order:
            self._install(k, nn.Parameter(self._view(k), requires_grad=True))
        if self.batch_norm:
            bn = self._modules['fc']._modules['0']
            bn.register_buffer('running_mean', sd['fc.0.running_mean'])
            bn.register_buffer('running_var', sd['fc.0.running_var'])
            bn.register_buffer('num_batches_tracked', sd['fc.0.num_batches_tracked'])

    def _conv_frames_from(self, observation_shape):
        return 0

    # -- plumbing ---------------------------------------------------------------------------------------------
    def _view(self, k):
        o, shp = self._slots[k]
        return self._flat[o:o + int(np.prod(shp))].view(shp)

    def _install(self, key, param):
        parts = key.split('.')
        node = self
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Holder())
            node = node._modules[p]
        node.register_parameter(parts[-1], param)

    def _param(self, key):
        node = self
        parts = key.split('.')
        for p in parts[:-1]:
            node = node._modules[p]
        return node, parts[-1]

    def _apply(self, fn, recurse=True):
        # move the flat buffer once and re-point every parameter view at it (nn.Module._apply would break the aliasing)
        self._flat = fn(self._flat)
        for k in self._order:
            node, leaf = self._param(k)
            node._parameters[leaf].data = self._view(k)
        if self.batch_norm:
            bn = self._modules['fc']._modules['0']
            for b in ('running_mean', 'running_var', 'num_batches_tracked'):
                bn._buffers[b] = fn(bn._buffers[b])
        self._release()
        return self

    def _release(self):
        if self._handle is not None:
            _plib().pvr_policy_destroy(self._handle)
            self._handle = None
        self._dp_key = None                              # (a new handle has no collective installed)

    def close(self):
        """Free the library handle (workspace, streams) NOW, at a point the caller chooses - after its own synchronisation - instead of
        whenever the garbage collector finds the object.  The module stays usable: the next forward / step builds a new handle."""
        self._release()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def check_status(self):
        """Raise if a persistent launch of this policy gave up waiting (pvr_policy_status: PVR_ERR_TIMEOUT, once per event).  Does not
        synchronise: call it after a sync of your own (bc_loop does when it reads the loss) to learn about the steps just run; every
        forward / step performs the same check on entry, so the event also surfaces at the next call."""
        if self._handle is not None:
            _lib.check(_plib().pvr_policy_status(self._handle))

    def recurrence_mode(self):
        """0 per-step launches, 1 / 2 persistent forward recurrence (counter / data-as-flag hand-off) - what the next forward uses"""
        return int(_plib().pvr_policy_recurrence_mode(self._handle)) if self._handle is not None else -1

    @property
    def _host(self):
        """parameters in host memory -> the library's HOST backend (pvr_policy_create_host: plain C++ loops, forward + fused step);
        the reference's model lives wherever flags.device says (main_bc_2.py:64-66), the CPU when there is no GPU"""
        return not self._flat.is_cuda

    def use_host_backend(self, on=True):
        """allow this policy to run on the library's host (CPU) backend while its parameters are in host memory (reference: --disable_cuda)"""
        self._host_ok = bool(on)
        return self

    def _stream(self):
        return None if self._host else _lib.stream_ptr()

    def _ensure(self, T, B):
        if self._host:
            # explicit opt-in only (use_host_backend(): bc_loop does it for --disable_cuda): parameters forgotten on the CPU fail loudly
            if self._conv_frames or not getattr(self, '_host_ok', False):
                raise RuntimeError('PolicyNet parameters are on %s: call .to(device="cuda") first (or use_host_backend(True) for the CPU plan of '
                                   'the vector policy: forward + fused step, what --disable_cuda selects)' % self._flat.device)
        else:
            _lib.require_gpu()
        if self._handle is not None and T <= self._max_t and B <= self._max_b and getattr(self, '_handle_host', None) == self._host:
            return
        self._release()
        self._max_t, self._max_b = max(self._max_t, T), max(self._max_b, B)
        L = _plib()
        d = PolicyDesc(self.obs_size, self.hidden, self.num_actions, int(self.batch_norm), self._max_t, self._max_b,
                       self._conv_frames)
        h = C.c_void_p()
        _lib.check((L.pvr_policy_create_host if self._host else L.pvr_policy_create)(C.byref(d), C.byref(h)))
        self._handle, self._handle_host = h, self._host
        # training-mode forwards sample their action inside the library (models.py:78-80).  A handle rebuilt for a larger T / B or after
        # .to() continues this module's noise stream (key and position kept here) instead of replaying it from call 0
        # (keyed per forward from torch's generator: _arm_sampling; a handle built here starts with sampling on and the last key)
        _lib.check(L.pvr_policy_set_action_sampling(h, 1, C.c_uint64(getattr(self, '_sample_key', None) or (torch.initial_seed() & 0x3FFFFFFFFFFFFFFF))))
        assert L.pvr_policy_param_count(h) == self._flat.numel(), 'flat layout mismatch with libpvr_hip'
        assert L.pvr_policy_trainable_count(h) == self._n_train
        for k in self._order:
            n = C.c_int64()
            assert L.pvr_policy_param_offset(h, k.encode(), C.byref(n)) == self._slots[k][0], k

    def _bn_struct(self):
        if not self.batch_norm:
            return None
        bn = self._modules['fc']._modules['0']
        return PolicyBN(bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr())

    # -- reference surface ------------------------------------------------------------------------------------
    @property
    def device(self):
        return self._flat.device

    def initial_state(self, batch_size):
        return tuple(torch.zeros(2, batch_size, self.hidden) for _ in range(2))

    def _arm_sampling(self):
        """Key of the action noise of ONE training-mode forward.  The reference's torch.multinomial (models.py:78-80) consumes torch's
        GLOBAL generator at every such forward: two modules never see the same noise, torch.manual_seed() restarts it at any time, and
        nothing replays when a module is rebuilt.  torch's sample stream itself cannot be reproduced (its Philox offsets depend on its
        launch geometry), so the library's counter-based stream is keyed, per forward, by one 62-bit draw FROM that generator: the noise
        is a function of the generator state at the call, exactly the reference's dependency structure (same seed + same call order ->
        same actions)."""
        nonce = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self._sample_key = nonce
        _lib.check(_plib().pvr_policy_set_action_sampling(self._handle, 1, C.c_uint64(nonce)))

    def _forward_raw(self, x, done, h0, c0, T, B, training):
        """one pvr_policy_forward enqueue on prepared device tensors -> (logits, baseline, action, h, c)"""
        if training:
            self._arm_sampling()
        dev, A = self.device, self.num_actions
        logits = torch.empty((T, B, A), dtype=torch.float32, device=dev)
        baseline = torch.empty((T, B), dtype=torch.float32, device=dev)
        action = torch.empty((T, B), dtype=torch.int64, device=dev)
        h = torch.empty((2, B, self.hidden), dtype=torch.float32, device=dev)
        c = torch.empty_like(h)
        bn = self._bn_struct()
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(_plib().pvr_policy_forward(self._handle, vp(self._flat), C.byref(bn) if bn else None, vp(x), vp(done),
                                              vp(h0), vp(c0), T, B, int(training), vp(logits), vp(baseline), vp(action),
                                              vp(h), vp(c), self._stream()))
        return logits, baseline, action, h, c

    def forward(self, inputs, core_state=()):
        x = inputs['obs']                                     # (unroll_length, batch_size, obs_size)
        T, B = x.shape[0], x.shape[1]
        self._ensure(T, B)
        dev = self.device
        x = self._prep_obs(x, dev)
        done = inputs['done'].to(device=dev).to(torch.uint8).contiguous()
        want_grad = self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        h0 = c0 = None
        if len(core_state) == 2:
            if want_grad:
                # BPTT starts from a zero state (main_bc_2.py:207-209 passes initial_state() every iteration); a non-zero state would
                # need its own terms in the weight gradients, which the backward plan does not carry
                assert not bool(core_state[0].any()) and not bool(core_state[1].any()), \
                    'training with autograd starts every unroll from model.initial_state() (zeros), as the reference loop does'
            else:
                h0 = core_state[0].to(device=dev, dtype=torch.float32).contiguous()
                c0 = core_state[1].to(device=dev, dtype=torch.float32).contiguous()
        A = self.num_actions
        if want_grad and self._host:
            raise NotImplementedError('host backend: the autograd bridge (loss.backward() through the library) is a HIP-plan feature; use the '
                                      'fused iteration (HipRMSprop.step, what main_bc_2 runs by default) or torch.no_grad() for inference')
        if want_grad:
            params = [self._param(k)[0]._parameters[self._param(k)[1]] for k in self._order]
            logits, baseline, action, h, c = _PolicyFunction.apply(self, x, done, h0, c0, T, B, *params)
        else:
            logits, baseline, action, h, c = self._forward_raw(x, done, h0, c0, T, B, self.training)
        # training mode: `action` is already the library's sample of softmax(logits) (models.py:78-80; pvr_policy_set_action_sampling)
        return dict(policy_logits=logits, baseline=baseline, action=action), (h, c)

    def set_data_parallel(self, group=None, sync_bn=True, stats_group=None):
        """Hand the library its collective (pvr_policy_set_data_parallel) for this policy's handle: every following training
        backward - fused step, step_data_parallel or loss.backward() through the autograd bridge - all-reduces its gradient
        buckets over `group` (and uses global-batch BatchNorm statistics with sync_bn).  World size 1 / no process group
        uninstalls.  Re-installed when the handle, the group or the SyncBN choice changes.

        SyncBN statistics travel on their own process group (own RCCL communicator / stream, see make_allreduce_fn).  Pass it as
        `stats_group` when `group` is a proper subgroup of the world: creating it here calls dist.new_group, a collective over the
        DEFAULT group that every rank of the world - members of `group` or not - has to reach.  Without `stats_group` this method
        creates one on first use (main_bc_finetune: group = the world, every rank reaches its first step), keyed by the group's
        ranks (not by id(group), which a later group object can reuse)."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        sync = bool(self.batch_norm and sync_bn)
        ranks = tuple(dist.get_process_group_ranks(group)) if (world > 1 and group is not None) else None
        key = (self._handle.value if self._handle is not None else None, ranks, world, sync, id(stats_group) if stats_group is not None else None)
        if getattr(self, '_dp_key', None) == key:
            return world
        if world > 1:
            if sync and stats_group is None:
                cache = PolicyNet._stats_groups
                if ranks not in cache:
                    cache[ranks] = dist.new_group(ranks=list(ranks) if ranks is not None else None)
                stats_group = cache[ranks]
            if not sync:
                stats_group = None
            self._dp_cb, self._dp_errors = make_allreduce_fn(group, 'cuda', stats_group)       # (the ctypes thunk must stay alive while installed)
            _lib.check(_plib().pvr_policy_set_data_parallel(self._handle, world, int(sync), self._dp_cb, None))
        else:
            _lib.check(_plib().pvr_policy_set_data_parallel(self._handle, 1, 0, ALLREDUCE_FN(), None))
            self._dp_cb, self._dp_errors = None, []
        self._dp_key = key
        return world

    def _checked(self, status):
        """a failed collective: re-raise the exception the callback caught instead of the library's generic message"""
        if status != 0 and getattr(self, '_dp_errors', None):
            e = self._dp_errors[-1]
            del self._dp_errors[:]
            raise e
        _lib.check(status)

    def _prep_obs(self, x, dev):
        if self._conv_frames:
            assert x.dtype == torch.uint8 and tuple(x.shape[2:]) == (64, 64, 3 * self._conv_frames), x.shape
            return torch.flatten(x, 0, 1).to(device=dev).contiguous()     # raw uint8 frames; /255 happens in the kernel
        return torch.flatten(x, 0, 1).float().to(device=dev).contiguous()

    def last_grads(self):
        """Flat pre-clip gradient of the last fused step as {state_dict key: tensor} (parity tests)."""
        g = torch.empty(self._n_train, dtype=torch.float32, device=self.device)
        _lib.check(_plib().pvr_policy_last_grads(self._handle, C.c_void_p(g.data_ptr()), self._stream()))
        out = {}
        for k in self._order:
            o, shp = self._slots[k]
            if o < self._n_train:
                out[k] = g[o:o + int(np.prod(shp))].view(shp)
        return out


class _DevMem(object):
    """`count` fp32 values at a raw device address, as an object torch.as_tensor can wrap without a copy"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2}


def make_allreduce_fn(group=None, device='cuda', stats_group=None):
    """The collective libpvr_hip.so calls for data-parallel training (pvr_allreduce_fn, include/pvr_policy.h): an in-place SUM
    all-reduce of `count` floats at `buf`, enqueued on the HIP stream the library names - its communication stream for the
    gradient buckets, the compute stream for SyncBN statistics.  torch.distributed under backend "nccl" is RCCL over xGMI;
    "gloo" (CPU tests, or ranks sharing one GPU) works too.  ctypes would print and swallow an exception raised inside the
    callback, so it is caught here, kept in `errors`, and turned into a non-zero status: the library entry point then fails
    with PVR_ERR_COMM and the caller re-raises the original exception.
    stats_group: a second process group over the same ranks for the collectives the library issues on its COMPUTE stream (SyncBN
    statistics: 2 x obs_size floats that the very next kernel needs).  A torch process group owns one RCCL stream per device, so on a
    single group such a 2 KB all-reduce queues behind the 33 MB gradient bucket enqueued just before it and the compute stream waits
    for the whole transfer it was meant to overlap with; its own group = its own communicator and stream.
    Returns (ctypes thunk - keep it alive while installed -, errors list)."""
    import torch.distributed as dist
    errors, streams = [], {}

    def _cb(buf, count, stream, user):
        try:
            if device == 'cuda':
                t = torch.as_tensor(_DevMem(buf, count), device='cuda')
                sp = int(stream or 0)
                if sp == torch.cuda.current_stream().cuda_stream:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=stats_group if stats_group is not None else group)
                else:
                    if sp not in streams:
                        streams[sp] = torch.cuda.ExternalStream(sp)
                    with torch.cuda.stream(streams[sp]):
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            else:
                t = torch.from_numpy(np.ctypeslib.as_array((C.c_float * int(count)).from_address(buf)))
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except BaseException as e:            # noqa: B902 - nothing may propagate into the C caller
            errors.append(e)
            return 1
    return ALLREDUCE_FN(_cb), errors


class PolicyNetWithConv(PolicyNet):
    """reference src/models.py:96-197: raw uint8 (T,B,64,64,3n) observations -> 5 x (conv3x3 s2 + ELU) per frame ->
    the same BN / FC / LSTM / heads.  Extra state_dict keys `feat_extract.{0,2,4,6,8}.{weight,bias}`."""

    def _conv_frames_from(self, observation_shape):
        return int(observation_shape[2]) // 3


class _HipOptimizer(object):
    """Shared plumbing of the fused optimisers: LambdaLR(1 - epoch/max_epochs) stepped BEFORE the update as the reference does
    (main_bc_2.py:216), flat state buffers with the parameter layout, torch-compatible state_dict, the data-parallel step."""
    _state_names = ()

    def __init__(self, model, lr, max_grad_norm, max_epochs):
        self.model, self.lr0 = model, float(lr)
        self.max_grad_norm, self.max_epochs = float(max_grad_norm), max_epochs
        self.last_epoch = 0
        self.steps = 0
        self._grads = None            # caller-owned flat gradient (two-half path)
        for n in self._state_names:
            setattr(self, n, torch.zeros_like(model._flat))

    def scheduler_step(self):
        self.last_epoch += 1

    def current_lr(self):
        if self.max_epochs is None:
            return self.lr0
        return self.lr0 * (1 - self.last_epoch / self.max_epochs)

    def _state_to(self, dev):
        for n in self._state_names:
            if getattr(self, n).device != dev:
                setattr(self, n, getattr(self, n).to(dev))

    def _backward(self, obs, done, actions):
        """forward + loss + backward into the flat gradient (all-reduced over the ranks when data parallelism is installed)"""
        m = self.model
        T, B = obs.shape[0], obs.shape[1]
        dev = m.device
        self._state_to(dev)
        if self._grads is None or self._grads.device != dev:
            self._grads = torch.zeros(m._n_train, dtype=torch.float32, device=dev)
        x = m._prep_obs(obs, dev)
        d = done.to(device=dev).to(torch.uint8).contiguous()
        a = actions.to(device=dev).long().contiguous()
        stats = torch.zeros(2, dtype=torch.float32, device=dev)
        bn = m._bn_struct()
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        m._checked(_plib().pvr_policy_backward(m._handle, vp(m._flat), C.byref(bn) if bn else None, vp(x), vp(d), vp(a), T, B,
                                               vp(self._grads), vp(stats), None, _lib.stream_ptr()))
        return stats

    def _apply(self, stats):
        raise NotImplementedError

    def step(self, obs, done, actions):
        """obs (T,B,obs) float, done (T,B) bool, actions (T,B) int -> (loss, grad_norm) device scalars."""
        self.model._ensure(obs.shape[0], obs.shape[1])
        stats = self._backward(obs, done, actions)
        self._apply(stats)
        self.steps += 1
        return stats[0], stats[1]

    def step_data_parallel(self, obs, done, actions, group=None, sync_bn=True):
        """Finetune configuration (BASELINE config 4): every rank runs forward/backward on its slice of the batch.
        Inside pvr_policy_backward the gradient leaves in four buckets, each all-reduced (RCCL over xGMI under backend 'nccl') on
        the library's communication stream as soon as backward has finalised it - LSTM layer 1 + policy head while layer 0 still
        runs its BPTT, layer 0 during the fc / conv backward, ... - and divided by the world size (the loss is a mean over the
        global batch); every rank then applies the identical clipped update.  With sync_bn (default) BatchNorm uses
        global-batch statistics, so N ranks x B/N sequences reproduce one rank x B; sync_bn=False keeps per-rank statistics
        (torch DDP default).  Without an initialised process group (or world size 1) this is the single-GPU iteration."""
        m = self.model
        m._ensure(obs.shape[0], obs.shape[1])
        m.set_data_parallel(group, sync_bn)
        stats = self._backward(obs, done, actions)
        self._apply(stats)
        self.steps += 1
        return stats[0], stats[1]

    # torch-compatible checkpoint layout (main_bc_2.py:255-257): state[i] per trainable parameter in model.parameters() order
    def _param_group(self):
        raise NotImplementedError

    def state_dict(self):
        state = {}
        for i, k in enumerate(self.model._order):
            o, shp = self.model._slots[k]
            if o < self.model._n_train:
                state[i] = {'step': torch.tensor(float(self.steps))}
                for n in self._state_names:
                    state[i][n] = getattr(self, n)[o:o + int(np.prod(shp))].view(shp).clone()
        g = dict(self._param_group(), lr=self.current_lr(), initial_lr=self.lr0, params=list(range(len(self.model._order))))
        return {'state': state, 'param_groups': [g], 'last_epoch': self.last_epoch}

    def load_state_dict(self, sd):
        for i, k in enumerate(self.model._order):
            if i in sd['state']:
                o, shp = self.model._slots[k]
                for n in self._state_names:
                    if n in sd['state'][i]:
                        getattr(self, n)[o:o + int(np.prod(shp))].copy_(sd['state'][i][n].reshape(-1))
                self.steps = int(sd['state'][i]['step'])
        self.last_epoch = sd.get('last_epoch', self.last_epoch)


class HipRMSprop(_HipOptimizer):
    """torch.optim.RMSprop(centered=False) + LambdaLR(1 - epoch/max_epochs) as used by main_bc_2.py:80-90, fused with the
    forward/backward of the BC loss (main_bc_2.py:206-227).  momentum == 0 (the reference default, src/arguments.py:63-64) runs the
    whole iteration as one enqueue (pvr_policy_step); momentum != 0 keeps torch's extra buffer (pvr_policy_apply_momentum).

    `scheduler_step()` mirrors the reference's `scheduler.step()` call, which precedes `optimizer.step()`
    (main_bc_2.py:216), so update k uses lr * (1 - (k+1)/max_epochs)."""

    def __init__(self, model, lr=1e-4, alpha=0.99, eps=1e-5, momentum=0, max_grad_norm=40.0, max_epochs=None):
        self.alpha, self.eps, self.momentum = float(alpha), float(eps), float(momentum)
        self._state_names = ('square_avg', 'momentum_buffer') if self.momentum != 0 else ('square_avg',)
        super().__init__(model, lr, max_grad_norm, max_epochs)

    def _param_group(self):
        return {'momentum': self.momentum, 'alpha': self.alpha, 'eps': self.eps, 'centered': False, 'weight_decay': 0}

    def _apply(self, stats):
        m = self.model
        vp = lambda t: C.c_void_p(t.data_ptr())
        if self.momentum != 0:
            _lib.check(_plib().pvr_policy_apply_momentum(m._handle, vp(m._flat), vp(self.square_avg), vp(self.momentum_buffer), vp(self._grads),
                                                         self.current_lr(), self.alpha, self.eps, self.momentum, self.max_grad_norm, vp(stats),
                                                         _lib.stream_ptr()))
        else:
            _lib.check(_plib().pvr_policy_apply(m._handle, vp(m._flat), vp(self.square_avg), vp(self._grads), self.current_lr(), self.alpha,
                                                self.eps, self.max_grad_norm, vp(stats), _lib.stream_ptr()))

    def step(self, obs, done, actions, return_logits=False):
        """obs (T,B,obs) float, done (T,B) bool, actions (T,B) int -> (loss, grad_norm) device scalars."""
        if self.momentum != 0:
            assert not return_logits
            return super().step(obs, done, actions)
        m = self.model
        T, B = obs.shape[0], obs.shape[1]
        m._ensure(T, B)
        dev = m.device
        self._state_to(dev)
        x = m._prep_obs(obs, dev)
        d = done.to(device=dev).to(torch.uint8).contiguous()
        a = actions.to(device=dev).long().contiguous()
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        logits = torch.empty((T, B, m.num_actions), dtype=torch.float32, device=dev) if return_logits else None
        bn = m._bn_struct()
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        m._checked(_plib().pvr_policy_step(m._handle, vp(m._flat), vp(self.square_avg), C.byref(bn) if bn else None, vp(x), vp(d),
                                           vp(a), T, B, self.current_lr(), self.alpha, self.eps, self.max_grad_norm, vp(stats),
                                           vp(logits), m._stream()))
        self.steps += 1
        return (stats[0], stats[1], logits) if return_logits else (stats[0], stats[1])


class HipAdam(_HipOptimizer):
    """torch.optim.Adam(betas, eps; amsgrad off, no weight decay) on the same flat buffers, with the same clip and LambdaLR order.
    Not in the reference (its scripts use RMSprop, main_bc_2.py:80-86): `--optimizer adam`, for BASELINE.json's north_star."""
    _state_names = ('exp_avg', 'exp_avg_sq')

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=40.0, max_epochs=None):
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        super().__init__(model, lr, max_grad_norm, max_epochs)

    def _param_group(self):
        return {'betas': self.betas, 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False}

    def _apply(self, stats):
        m = self.model
        vp = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(_plib().pvr_policy_apply_adam(m._handle, vp(m._flat), vp(self.exp_avg), vp(self.exp_avg_sq), vp(self._grads), self.current_lr(),
                                                 self.betas[0], self.betas[1], self.eps, self.steps + 1, self.max_grad_norm, vp(stats),
                                                 _lib.stream_ptr()))


def make_optimizer(flags, model, max_epochs):
    """the optimiser of main_bc_2.py:80-90 from the reference's flags (+ --optimizer adam)"""
    if getattr(flags, 'optimizer', 'rmsprop') == 'adam':
        return HipAdam(model, lr=flags.learning_rate, max_grad_norm=flags.max_grad_norm, max_epochs=max_epochs)
    return HipRMSprop(model, lr=flags.learning_rate, momentum=flags.momentum, eps=flags.epsilon, alpha=flags.alpha,
                      max_grad_norm=flags.max_grad_norm, max_epochs=max_epochs)
