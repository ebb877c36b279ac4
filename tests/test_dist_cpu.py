"""The N>1 path on CPU: world_size-2 gloo processes.  Two ranks, each holding half of the
envs' rollout, must produce the SAME policy update as one process holding all of them: flat-bucket
gradient all-reduce + global advantage normalisation.  Also the bench's timing reduction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from solorl_amd.ppo import Policy, RolloutStorage, PPO
from solorl_amd.ppo import dist as D
from solorl_amd.vec_env import Box

T, N, O, A = 6, 8, 76, 12


def make_storage(seed, n):
    g = torch.Generator().manual_seed(seed)
    s = RolloutStorage(T, n, (O,), A, torch.device("cpu"))
    s.obs.copy_(torch.randn(s.obs.shape, generator=g)); s.rewards.copy_(torch.randn(s.rewards.shape, generator=g))
    s.value_preds.copy_(torch.randn(s.value_preds.shape, generator=g)); s.actions.copy_(torch.randn(s.actions.shape, generator=g))
    s.action_log_probs.copy_(torch.randn(s.action_log_probs.shape, generator=g) - 12)
    s.masks.copy_((torch.rand(s.masks.shape, generator=g) > 0.1).float())
    s.compute_returns(torch.randn(n, 1, generator=g), True, 0.99, 0.95)
    return s


def shard(full, r, w):
    n = full.num_agents // w
    s = RolloutStorage(T, n, (O,), A, torch.device("cpu"))
    for k in ("obs", "rewards", "value_preds", "returns", "actions", "action_log_probs", "masks"):
        getattr(s, k).copy_(getattr(full, k)[:, r * n:(r + 1) * n])
    return s


def new_policy():
    torch.manual_seed(11)
    return Policy((O,), Box(-np.ones(A), np.ones(A)), None, {"hidden_size": 32})


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    D.init_from_env("cpu")
    assert D.world() == world and D.rank() == rank
    full = make_storage(5, N)
    pol = new_policy()
    if rank == 1:                                   # replicas must be identical after the broadcast
        with torch.no_grad():
            for p in pol.parameters():
                p.add_(1.0)
    D.broadcast_parameters(pol)
    agent = PPO(pol, 0.1, 2, T * N // world, 0.5, 0.01, lr=1e-3, max_grad_norm=0.5)
    losses = agent.update(shard(full, rank, world))
    # bench.py's reduction: max over ranks of the elapsed time
    el = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    m, s = D.global_mean_std(shard(full, rank, world).returns[:-1])
    # identical replicas after the update: every rank's weights equal rank 0's, bit for bit
    for p in pol.parameters():
        ref = p.detach().clone(); dist.broadcast(ref, 0)
        assert torch.equal(ref, p.detach())
    if rank == 0:
        torch.save(dict(state=pol.state_dict(), losses=losses, el=float(el), mean=float(m), std=float(s)), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_w_ranks_equal_one_process(tmp_path, world):
    """world 2 and world 8 (the node size of BASELINE configs 4-5; gloo, one env per rank at 8): W ranks holding 1/W of the rollout each
    produce the update one process computes from all of it -- no deadlock, identical replicas."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=False)
    assert got["el"] == float(world)
    full = make_storage(5, N)
    assert got["mean"] == pytest.approx(float(full.returns[:-1].mean()), rel=1e-5)
    assert got["std"] == pytest.approx(float(full.returns[:-1].std()), rel=1e-5)
    pol = new_policy()
    agent = PPO(pol, 0.1, 2, T * N, 0.5, 0.01, lr=1e-3, max_grad_norm=0.5)
    agent.update(full)
    ref = pol.state_dict()
    for k, v in got["state"].items():
        assert torch.allclose(v, ref[k], atol=5e-6), k


def test_flat_bucket_views_gradients():
    pol = new_policy()
    b = D.FlatGradBucket(pol.parameters())
    assert b.flat.numel() == sum(p.numel() for p in pol.parameters())
    x = torch.randn(4, O)
    v, lp, ent = pol.evaluate_actions(x, torch.randn(4, A))
    (v.sum() + lp.sum() + ent).backward()
    off = 0
    for p in pol.parameters():
        assert p.grad.data_ptr() == b.flat[off:].data_ptr()      # autograd accumulated into the bucket
        off += p.numel()
    assert float(b.flat.abs().sum()) > 0
    b.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in pol.parameters())
