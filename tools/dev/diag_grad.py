"""Per-tensor error of MiniBatchGrad against an f64 autograd reference at several mini-batch sizes (dev diagnostic, GPU)."""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_train_gpu import _random_policy
from solorl_amd.ppo import RolloutStorage
from solorl_amd.ppo import dist as D
from solorl_amd.ppo.fused import MiniBatchGrad
dev = torch.device("cuda:0")
O, A, clip, vc, ec = 76, 12, 0.1, 0.5, 0.01
for (T, N, m, offset) in ((16, 128, 1024, 512), (16, 1024, 8192, 1024), (16, 4096, 32768, 4096), (16, 4096, 32768, 0)):
    pol = _random_policy(dev, O, A, seed=1)
    st = RolloutStorage(T, N, (O,), A, dev)
    torch.manual_seed(2)
    st.obs.normal_(); st.actions.normal_(); st.value_preds.normal_(); st.returns.normal_()
    n = T * N
    flat = lambda x: x.reshape(n, *x.shape[2:])
    with torch.no_grad():
        _, lp, _ = pol.evaluate_actions(flat(st.obs[:-1]), flat(st.actions))
        st.action_log_probs.copy_((lp + 0.15 * torch.randn_like(lp)).view(T, N, 1))
    adv = torch.randn(n, 1, device=dev)
    perm = torch.randperm(n, device=dev)
    off = torch.full((), offset, dtype=torch.long, device=dev)
    bucket = D.FlatGradBucket(pol.parameters())
    mb = MiniBatchGrad(pol, st, m, clip, vc, ec, True, perm, off, adv)
    bucket.flat.fill_(123.0)
    mb()
    gk = bucket.flat.clone()
    idx = perm[offset:offset + m]
    obs_b, act_b = flat(st.obs[:-1])[idx].double(), flat(st.actions)[idx].double()
    vp, ret, old, ad = [x.double() for x in (flat(st.value_preds[:-1])[idx], flat(st.returns[:-1])[idx], flat(st.action_log_probs)[idx], adv[idx])]
    pol64 = copy.deepcopy(pol).double()
    for p in pol64.parameters(): p.grad = None
    v, lp, ent = pol64.evaluate_actions(obs_b, act_b)
    r = torch.exp(lp - old)
    al = -torch.min(r * ad, torch.clamp(r, 1 - clip, 1 + clip) * ad).mean()
    vcl = vp + (v - vp).clamp(-clip, clip)
    vl = 0.5 * torch.max((v - ret).pow(2), (vcl - ret).pow(2)).mean()
    (vl * vc + al - ent * ec).backward()
    print("m = %d, offset %d" % (m, offset))
    o = 0
    for name, p in pol64.named_parameters():
        k = p.numel()
        e = (gk[o:o + k].double() - p.grad.flatten()).abs()
        print("  %-28s max err %.3g at %d (of %d; shape %s), max |grad| %.3g, mean err %.3g" % (name, e.max().item(), e.argmax().item(), k, tuple(p.shape), p.grad.abs().max().item(), e.mean().item()))
        o += k
