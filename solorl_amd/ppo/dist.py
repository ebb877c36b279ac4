"""Data-parallel plumbing: one process per GPU, envs sharded across ranks (no data-path
collective), policy gradients all-reduced over RCCL/xGMI.

SURVEY.md 8(e): the gradient is ONE flat fp32 bucket -- 20 057 parameters = 80 KB for Solo12
pointGoal / hidden 64 -- so the all-reduce is latency-bound, not link-bandwidth-bound: a single
`all_reduce(sum)` per optimizer step on a persistent flat buffer (no per-parameter calls, no
bucketing logic), then a scale by 1/world.  A second 3-float all-reduce reproduces the reference's
whole-batch advantage normalisation (agents/ppo/ppo.py:35-37)."""
import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(device_type="cuda"):
    """torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; backend nccl (= RCCL) on
    GPUs, gloo on CPU (tests)."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    if w > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        pin_rccl_from_env()
        if device_type == "cuda":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
        else:
            dist.init_process_group("gloo")
    return rank(), world()


def pin_rccl_from_env():
    """the 80 KB gradient bucket is latency-bound, a tree / direct exchange suits it better than a ring.  RCCL reads
    its algorithm / protocol filters (NCCL_ALGO, NCCL_PROTO) when the communicator is created, so they have to be in the environment
    BEFORE init_process_group: SOLORL_RCCL_ALGO / SOLORL_RCCL_PROTO (e.g. Tree / LL) are exported as those here.  Unset (the default)
    leaves the choice to RCCL's own tuner: a filter that excludes every algorithm of some collective (broadcast only has Ring)
    fails the run, and no box of this project has had two GPUs to try one on -- bench.py records what was in effect and
    the measured bucket latency beside an 8-byte all-reduce, which is what says whether the tuner's choice is latency-bound."""
    for src, dst in (("SOLORL_RCCL_ALGO", "NCCL_ALGO"), ("SOLORL_RCCL_PROTO", "NCCL_PROTO")):
        v = os.environ.get(src)
        if v:
            os.environ[dst] = v
    return rccl_env()


def rccl_env():
    return {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "RCCL_MSCCL_ENABLE")
            if os.environ.get(k) is not None}


class FlatGradBucket:
    """All parameters' gradients viewed into one contiguous buffer (set once; autograd then
    accumulates straight into it), all-reduced with a single collective."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        w = world()
        if w > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / w)

    def all_reduce_sum(self):
        """The collective alone (the consumer applies 1 / world itself: solorl_ppo_clip_adam's grad_scale); RCCL collectives can
        be captured in a HIP graph, so with the nccl backend this call sits INSIDE the captured mini-batch step."""
        if world() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)


def backend():
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def global_mean_std(x, unbiased=True, eps=0.0):
    """mean/std of a tensor sharded over ranks == torch.mean/std of the concatenation."""
    s = torch.stack([x.sum(), (x * x).sum(), torch.tensor(float(x.numel()), device=x.device, dtype=x.dtype)])
    if world() > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    n = s[2]
    mean = s[0] / n
    var = (s[1] - n * mean * mean) / (n - 1 if unbiased else n)
    return mean, torch.sqrt(torch.clamp(var, min=0.0)) + eps


def broadcast_parameters(module, src=0):
    if world() > 1:
        for p in module.parameters():
            dist.broadcast(p.data, src)
        for b in module.buffers():
            dist.broadcast(b.data, src)
