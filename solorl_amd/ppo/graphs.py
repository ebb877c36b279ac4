"""HIP-graph capture of the two launch-bound loops of PPO training on a GPU (PyTorch's CUDAGraph API is
hipGraph on ROCm).

At 4096 envs the engine's step kernel takes ~0.2 ms, a policy forward ~10 small kernels and a PPO
mini-batch step (gather, forward, losses, backward, clip, Adam) ~100 kernels of a few microseconds each:
run eagerly, both loops are bound by launch latency and Python, not by the GPU.  Captured once and
replayed, a whole T-step rollout is ONE graph launch and every mini-batch step one (two with a
gradient all-reduce between them when world > 1).

  GraphedRollout  act -> solorl_step (the C-ABI launch is captured like any other kernel: it is enqueued on
                  torch's capturing stream and never synchronises) -> storage.append, T times
  GraphedPPO      PPO.update with identical arithmetic (agents/ppo/ppo.py:34-89); mini-batch indices are
                  read on the device from a permutation buffer at a device-side offset, so a replay needs
                  no host-side argument

Both fall back to nothing: they are only constructed for CUDA devices; train() uses them unless
--no-hip-graphs is given."""
import torch
import torch.nn as nn

from . import dist as D
from .fused import ClipAdam, MiniBatchGrad, fused_ppo_loss, policy_act, policy_kernels_supported, policy_params
from .ppo import PPO


def _kernels_enabled():
    """SOLORL_PPO_KERNELS=0 keeps the rollout's act and the mini-batch step on their PyTorch paths (A/B runs)."""
    import os
    return os.environ.get("SOLORL_PPO_KERNELS", "1") != "0"


def capture_kwargs():
    """With a process group up, its watchdog thread issues HIP calls of its own; a capture in the default "global" error mode
    would be invalidated by them.  Thread-local mode confines the capture's checks to the capturing thread."""
    return {"capture_error_mode": "thread_local"} if D.world() > 1 else {}


def _side_stream_warmup(fn, iters=3):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(iters):
            fn()
    torch.cuda.current_stream().wait_stream(s)


def probe_captured_allreduce(dev, numel, replays=8, time_it=False):
    """Does an all-reduce(sum) of `numel` floats -- the real bucket size -- run at every REPLAY of a captured graph?  (A collective that
    ran at capture time only would leave stale data.)  Returns (ok, seconds per replay or None); ok is the same on every rank.

    Protocol (a rank whose capture raised must not sit in a different collective than its peers): every step that can
    differ between ranks is followed by an EAGER MIN all-reduce of a flag that every rank reaches --
      1. eager all-reduce (communicator set up outside any capture)
      2. capture, in a try block                       -> agree: did EVERY rank capture?  no: return False everywhere, nobody replays
      3. `replays` replays with different inputs, each result checked against the closed form   -> agree on the outcome."""
    import time
    import torch.distributed as dist
    w, r = D.world(), D.rank()

    def agree(flag):
        t = torch.tensor([1 if flag else 0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    probe = torch.zeros(int(numel), device=dev)
    dist.all_reduce(probe)
    torch.cuda.synchronize()
    g, captured = None, True
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, **capture_kwargs()):
            dist.all_reduce(probe, op=dist.ReduceOp.SUM)
    except Exception:
        captured = False
    if not agree(captured):
        return False, None
    ok = True
    for k in range(1, replays + 1):
        probe.fill_(float(k * (r + 1)))
        g.replay()
        torch.cuda.synchronize()
        ok = ok and bool((probe == float(k * w * (w + 1) // 2)).all().item())
    ok = agree(ok)
    sec = None
    if ok and time_it:
        probe.zero_()                         # (sums of zeros: the timing loop cannot overflow)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / 50
    return ok, sec


class GraphedRollout:
    """One graph = num_steps x (policy.act, env.step, storage.append).  Episode statistics need nothing here: the step
    kernel accumulates them on the device (include/solorl.h `ep_stats`), replayed or not."""

    def __init__(self, envs, actor_critic, storage, num_steps):
        self.envs, self.ac, self.storage, self.T = envs, actor_critic, storage, num_steps
        self.graph = None

    def _body(self):
        st = self.storage
        assert st.step == 0, "graphed rollouts start at storage step 0"
        # the arithmetic of train.rollout() with every result written where it is stored: the policy's outputs and the engine's
        # observations / rewards / done flags go straight into storage-side rows (the C ABI takes caller-owned pointers) -- two
        # launches per step (act, env step) instead of 35, and launches are what a step costs besides the env kernel
        fused = _kernels_enabled() and policy_kernels_supported(self.ac)
        if fused:                            # one launch for the whole of Policy.act (csrc/solorl_ppo.hip)
            self._pp = policy_params(self.ac)
        # the normal draws of all T steps and masks = 1 - done of all T steps are one launch each per rollout, not per step
        noise = torch.randn_like(st.actions) if fused else None
        done = torch.empty(self.T, st.num_agents, dtype=torch.uint8, device=st.device)
        # ONE launch per rollout step where the engine supports it (solorl_step_act: each wavefront of the step kernel evaluates the policy
        # on the observations it has just produced, in the time it would otherwise idle until the slowest wavefront ends); the first
        # step's action still needs its own policy launch.  SOLORL_STEP_ACT=0 keeps the two-launch form (A/B runs).
        import os
        self.step_act = bool(fused and os.environ.get("SOLORL_STEP_ACT", "1") != "0" and hasattr(self.envs, "step_act_supported")
                             and self.envs.step_act_supported(self._pp))
        for t in range(self.T):
            if self.step_act:
                if t == 0:
                    policy_act(self._pp, st.obs[0], noise[0], st.value_preds[0], st.actions[0], st.action_log_probs[0])
                if t + 1 < self.T:
                    self.envs.step_act_inplace(st.actions[t], self._pp, noise[t + 1], st.value_preds[t + 1], st.actions[t + 1], st.action_log_probs[t + 1],
                                               obs_out=st.obs[t + 1], rew_out=st.rewards[t], done_out=done[t])
                else:
                    self.envs.step_inplace(st.actions[t], obs_out=st.obs[t + 1], rew_out=st.rewards[t], done_out=done[t])
                continue
            if fused:
                policy_act(self._pp, st.obs[t], noise[t], st.value_preds[t], st.actions[t], st.action_log_probs[t])
            else:
                self.ac.act_into(st.obs[t], st.value_preds[t], st.actions[t], st.action_log_probs[t])
            self.envs.step_inplace(st.actions[t], obs_out=st.obs[t + 1], rew_out=st.rewards[t], done_out=done[t])
        torch.sub(torch.ones((), device=st.device), done, out=st.masks[1:].view(self.T, st.num_agents))

    def _capturable(self):
        # contact-count sorting (opt-in, SOLORL_SORT=1) ping-pongs two state buffers on the host side of
        # solorl_step: a captured sequence is only self-consistent over an even number of steps
        import os
        sort = os.environ.get("SOLORL_SORT", "0") != "0"
        return not sort or self.T % 2 == 0

    def __call__(self):
        if not self._capturable():
            from .train import rollout
            return rollout(self.envs, self.ac, self.storage, self.T)
        if self.graph is None:
            # no warm-up replay of the body: the env must not be stepped twice.  Everything the body allocates
            # is allocated inside the capture (private pool); cuBLAS/hipBLASLt workspaces were created by the
            # eager policy calls that precede the first rollout (train(): reset + get_value warm-up).
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **capture_kwargs()):
                self._body()
            self.graph = g
        self.graph.replay()


class GraphedPPO(PPO):
    """PPO with the mini-batch step captured.  Adam runs in capturable mode with the learning rate held in a
    device tensor (update_linear_schedule fills it), otherwise the arithmetic is PPO.update's."""

    def __init__(self, actor_critic, clip_param, ppo_epoch, mini_batch_size, value_loss_coef, entropy_coef, lr=None,
                 l2_coef=0.0, max_grad_norm=None, use_clipped_value_loss=True, fused_loss=True, fused_mlp=True):
        super().__init__(actor_critic, clip_param, ppo_epoch, mini_batch_size, value_loss_coef, entropy_coef, lr=lr,
                         l2_coef=l2_coef, max_grad_norm=max_grad_norm, use_clipped_value_loss=use_clipped_value_loss)
        self.fused_loss = fused_loss
        # hand-written mini-batch kernel (csrc/solorl_ppo.hip) where it is built for this policy's shape; otherwise autograd
        self.fused_mlp = fused_mlp and _kernels_enabled() and policy_kernels_supported(actor_critic)
        dev = next(actor_critic.parameters()).device
        # (fused: one multi-tensor launch per optimizer step instead of a dozen -- a captured mini-batch step is ~100 launches of a
        # few microseconds each, so launches are what it costs)
        self.optimizer = torch.optim.Adam(actor_critic.parameters(), lr=torch.tensor(float(lr), device=dev),
                                          weight_decay=l2_coef, capturable=True, fused=True)
        self._built_for = None

    # ---- capture
    def _build(self, storage):
        dev = storage.device
        n, m = storage.num_samples, self.mini_batch_size
        flat = lambda x: x.reshape(n, *x.shape[2:])
        self._src = (flat(storage.obs[:-1]), flat(storage.actions), flat(storage.value_preds[:-1]), flat(storage.returns[:-1]),
                     flat(storage.action_log_probs))
        assert all(s.data_ptr() == b.data_ptr() for s, b in zip(self._src, (storage.obs, storage.actions, storage.value_preds,
                                                                            storage.returns, storage.action_log_probs)))
        self._adv = torch.zeros(n, 1, device=dev)
        self._perm = torch.zeros(n, dtype=torch.long, device=dev)
        self._off = torch.zeros((), dtype=torch.long, device=dev)
        self._ar = torch.arange(m, device=dev)
        self._stats = torch.zeros(3, device=dev)
        clip = self.clip_param
        self._mb = None
        if self.fused_mlp and m % MiniBatchGrad.SLICE == 0:
            self._mb = MiniBatchGrad(self.actor_critic, storage, m, clip, self.value_loss_coef, self.entropy_coef, self.use_clipped_value_loss,
                                     self._perm, self._off, self._adv)

        self._ca = None
        if self._mb is not None:             # norm clip + Adam + cursor in one launch; lr stays the optimizer's device tensor
            g0 = self.optimizer.param_groups[0]
            self._ca = ClipAdam(self._mb, g0["lr"], self.max_grad_norm, weight_decay=g0["weight_decay"], betas=g0["betas"], eps=g0["eps"],
                                offset=self._off, offset_increment=m)

        def fwd_bwd():
            if self._mb is not None:     # three launches: gather + forward + losses + back-propagation; weight gradients; their sum
                self._mb()
                return
            idx = self._perm.index_select(0, self._ar + self._off)
            self._off += m
            obs_b, act_b, vpred_b, ret_b, old_lp_b = (s.index_select(0, idx) for s in self._src)
            adv_b = self._adv.index_select(0, idx)
            if self.fused_loss:       # one HIP launch for log-prob, both losses and their head gradients (ppo/fused.py)
                values, mean, logstd = self.actor_critic.heads(obs_b)
                loss, value_loss, action_loss, entropy = fused_ppo_loss(mean, logstd, values, act_b, old_lp_b, adv_b, vpred_b, ret_b, clip,
                                                                        self.value_loss_coef, self.entropy_coef, self.use_clipped_value_loss)
            else:
                values, logp, entropy = self.actor_critic.evaluate_actions(obs_b, act_b)
                ratio = torch.exp(logp - old_lp_b)
                action_loss = -torch.min(ratio * adv_b, torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv_b).mean()
                if self.use_clipped_value_loss:
                    v_clipped = vpred_b + (values - vpred_b).clamp(-clip, clip)
                    value_loss = 0.5 * torch.max((values - ret_b).pow(2), (v_clipped - ret_b).pow(2)).mean()
                else:
                    value_loss = 0.5 * (ret_b - values).pow(2).mean()
                loss = value_loss * self.value_loss_coef + action_loss - entropy * self.entropy_coef
            self.bucket.zero()
            loss.backward()
            self._stats += torch.stack([value_loss.detach(), action_loss.detach(), entropy.detach()])

        def opt_step():
            if self._ca is not None:
                self._ca()
                return
            if self.max_grad_norm is not None:
                # clip_grad_norm_ (agents/ppo/ppo.py:75-76) on the flat bucket every .grad is a view of: the 2-norm of the
                # per-tensor norms is the norm of the concatenation
                flat = self.bucket.flat
                flat.mul_((self.max_grad_norm / (flat.norm() + 1e-6)).clamp(max=1.0))
            self.optimizer.step()

        # warm-up on a side stream (allocator, autograd and optimizer state), then put everything back
        params = [p for p in self.actor_critic.parameters()]
        saved = [p.detach().clone() for p in params]
        self._perm.copy_(torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(0)))   # (own generator: the
                                                                                   # global RNG stream stays the eager path's)
        def warm():
            self._off.zero_()
            fwd_bwd()
            opt_step()
        _side_stream_warmup(warm)
        with torch.no_grad():
            for p, s in zip(params, saved):
                p.copy_(s)
            for st in self.optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
            if self._ca is not None:
                self._ca.zero_state()
        self._stats.zero_(); self._off.zero_()
        if self._mb is not None:
            self._mb.reset()
        # world > 1: one flat-bucket all-reduce per optimizer step, between the gradients and the clip.  Default: the
        # SPLIT form -- gradients graph -> eager all-reduce -> optimizer graph (clip + Adam with grad_scale = 1 / world in one launch).
        # SOLORL_CAPTURE_ALLREDUCE=1 with the nccl (= RCCL) backend captures the collective with the rest, so that a mini-batch step is
        # ONE graph replay; a probe graph of the real bucket size first checks on every rank that a captured all-reduce really runs
        # at replay (probe_captured_allreduce).  Opt-in because it has never run on two GPUs.
        w = D.world()
        self._split = False
        if w > 1:
            self._split = not self._collective_is_capturable(dev)
        self.allreduce_in_graph = w > 1 and not self._split
        if self._ca is not None:
            self._ca.S.grad_scale = 1.0 / w          # the bucket arrives SUMMED (all_reduce_sum); the scale is fused into the step
        reduce_ = self.bucket.all_reduce_sum if self._ca is not None else self.bucket.all_reduce_mean
        self._reduce = reduce_
        pool = torch.cuda.graph_pool_handle()
        self._g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g1, pool=pool, **capture_kwargs()):
            fwd_bwd()
            if not self._split:
                if w > 1:
                    reduce_()
                opt_step()
        if self._split:                       # the all-reduce of the flat bucket runs between two graphs
            self._g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g2, pool=pool, **capture_kwargs()):
                opt_step()
        with torch.no_grad():                 # capture does not execute, but be explicit about the state we start from
            self._stats.zero_(); self._off.zero_()
        self._built_for = (id(storage), n, m)

    def _collective_is_capturable(self, dev):
        """Whether the bucket all-reduce goes INSIDE the captured mini-batch step.  Opt-in (SOLORL_CAPTURE_ALLREDUCE=1) until a real
        multi-GPU run has exercised it: no box of this project has had two GPUs, so the split form (graph -> eager all-reduce -> graph)
        stays the default and `allreduce_in_graph` is recorded with every result."""
        import os
        want = D.backend() == "nccl" and os.environ.get("SOLORL_CAPTURE_ALLREDUCE", "0") == "1" and os.environ.get("SOLORL_SPLIT_ALLREDUCE", "0") == "0"
        if not want:
            return False
        ok, _ = probe_captured_allreduce(dev, self.bucket.flat.numel())
        return ok

    def update(self, storage):
        if not storage.rewards.is_cuda:
            return super().update(storage)
        if self._built_for != (id(storage), storage.num_samples, self.mini_batch_size):
            self._build(storage)
        n, m = storage.num_samples, self.mini_batch_size
        adv = storage.returns[:-1] - storage.value_preds[:-1]
        mean, std = D.global_mean_std(adv, unbiased=True)
        self._adv.copy_(((adv - mean) / (std + 1e-5)).reshape(n, 1))
        self._stats.zero_()
        if self._mb is not None:
            self._mb.reset()
        n_updates = 0
        for _ in range(self.ppo_epoch):
            self._perm.copy_(torch.randperm(n, device=storage.device))
            self._off.zero_()
            for _s in range(0, n - m + 1, m):
                self._g1.replay()
                if self._split:
                    self._reduce()
                    self._g2.replay()
                n_updates += 1
        if self._mb is not None:
            return self._mb.losses(n_updates)
        v, a, e = (self._stats / max(n_updates, 1)).tolist()
        return v, a, e
