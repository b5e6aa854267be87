"""GPU: training-step tail through the C ABI -- hn_surv_nll and hn_l1_adam_step -- against fixtures
generated from the reference (tests/golden/g7_*, tools/gen_goldens_train.py) and against the CPU oracle, then a whole
training step of a small model (forward, loss, backward into the flat gradient buffer, fused L1 + Adam under torch's
OneCycleLR) against the reference-equivalent CPU pipeline (oracle forward + autograd + calc_reg_loss + torch.optim.Adam)."""
import pytest
import torch

from conftest import assert_close, load_golden, rel_err
from oracle import healnet_cpu as O
from oracle import train_cpu as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


def test_surv_nll_matches_reference_fixtures(hn, manifest):
    g = load_golden("g7_surv_nll")
    meta = manifest["g7_surv_nll"]
    for case in meta["cases"]:
        t = case["tag"]
        logits = g[t + "logits"].to(DEV).requires_grad_(True)
        w = g[t + "weights"].to(DEV) if case["weighted"] else None
        out = hn.train.surv_nll_loss(logits, g[t + "y"].to(DEV), g[t + "c"].to(DEV), weights=w, alpha=meta["alpha"], eps=meta["eps"])
        assert abs(float(out.loss) - float(g[t + "loss"])) <= 2e-6 * max(1.0, abs(float(g[t + "loss"]))), t
        assert_close(out.hazards.cpu(), g[t + "hazards"], rel=1e-6, floor=1e-7, what=t + "hazards")
        assert_close(out.survival.cpu(), g[t + "survival"], rel=2e-6, floor=1e-7, what=t + "survival")
        assert_close(out.risk.cpu(), g[t + "risk"], rel=2e-6, what=t + "risk")
        (out.loss * 3.0).backward()
        assert_close(logits.grad.cpu() / 3.0, g[t + "dlogits"], rel=1e-5, floor=1e-6, what=t + "dlogits")


def test_surv_nll_large_batch_vs_oracle(hn):
    gen = torch.Generator().manual_seed(9)
    b, k = 1000, 4
    logits = torch.randn(b, k, generator=gen) * 3
    y = torch.randint(0, k, (b,), generator=gen)
    c = torch.randint(0, 2, (b,), generator=gen)
    w = torch.rand(k, generator=gen) + 0.1
    loss, dl, hz, sv = T.surv_nll(logits, y, c, w)
    lg = logits.to(DEV).requires_grad_(True)
    out = hn.train.surv_nll_loss(lg, y.to(DEV), c.to(DEV), weights=w.to(DEV))
    out.loss.backward()
    assert abs(float(out.loss) - float(loss)) <= 1e-5 * abs(float(loss))
    assert_close(lg.grad.cpu(), dl, rel=1e-5, floor=1e-6, what="dlogits b=1000")


def test_fused_l1_adam_matches_reference_fixture(hn, manifest):
    """torch.optim.Adam + OneCycleLR + calc_reg_loss, 4 steps, as recorded from the reference's own pieces."""
    g = load_golden("g7_l1_adam")
    m = manifest["g7_l1_adam"]

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(g[f"p{i}_init"].clone()) for i in range(len(m["shapes"]))])

    model = Holder().to(DEV)
    flat = hn.train.flatten_parameters(model)
    opt = hn.train.FusedL1Adam(flat, lr=m["lr"], l1=m["l1"])
    sched = torch.optim.lr_scheduler.OneCycleLR(optimizer=opt, max_lr=m["max_lr"], epochs=m["epochs"], steps_per_epoch=m["steps_per_epoch"])
    coef = [g[f"coef{i}"].to(DEV) for i in range(len(m["shapes"]))]
    for t in range(m["n_steps"]):
        opt.zero_grad()
        assert abs(opt.param_groups[0]["lr"] - float(g["lrs"][t])) < 1e-12
        assert abs(opt.param_groups[0]["betas"][0] - float(g["beta1s"][t])) < 1e-12
        loss = sum(((p * c).sum() ** 2 + (p * p * c).sum()) for p, c in zip(model.ps, coef)) * (0.1 * (t + 1))
        (loss / m["gc"]).backward()            # plain autograd: AccumulateGrad adds into the flat views
        opt.step()
        sched.step()
        assert abs(float(opt.reg_loss) - float(g["reg_losses"][t])) <= 1e-5 * float(g["reg_losses"][t])
        for i, p in enumerate(model.ps):
            assert p.data_ptr() >= flat.params.data_ptr()          # still a view of the flat buffer
            assert_close(p.detach().cpu(), g[f"p{i}_step{t}"], rel=3e-6, floor=1e-7, what=f"p{i} step{t}")


def test_whole_training_step_vs_reference_equivalent_cpu(hn):
    """3 steps: HealNet forward (tape) -> hn_surv_nll -> hn_fusion_backward into the flat gradient buffer -> fused
    L1 + Adam, against oracle forward + autograd + l1 * sum|p| + torch.optim.Adam on the CPU (the reference's loop)."""
    torch.manual_seed(11)
    kw = dict(n_modalities=2, channel_dims=[6, 3], num_spatial_axes=[1, 2], out_dims=4, depth=2, l_c=8, l_d=16, x_heads=2,
              l_heads=2, cross_dim_head=8, latent_dim_head=8)
    model = hn.HealNet(**kw).train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = O.FusionConfig(**kw)
    gen = torch.Generator().manual_seed(4)
    b = 6
    tab = torch.rand(b, 1, 6, generator=gen)
    img = torch.rand(b, 5, 7, 3, generator=gen)
    y = torch.randint(0, 4, (b,), generator=gen)
    c = torch.randint(0, 2, (b,), generator=gen)
    w = torch.rand(4, generator=gen) + 0.3
    l1, lr, gc = 2e-4, 3e-3, 2

    # ---- reference-equivalent CPU pipeline
    cpu = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    opt_c = torch.optim.Adam(list(cpu.values()), lr=lr)
    losses_c = []
    for _ in range(3):
        opt_c.zero_grad()
        logits = O.fusion_forward(cpu, cfg, [tab, img])
        hz = torch.sigmoid(logits)
        S = torch.cumprod(1 - hz, dim=1)
        Y, cc = y.view(b, 1), c.view(b, 1).float()
        Sp = torch.cat([torch.ones_like(cc), S], 1)
        unc = -(1 - cc) * (torch.log(torch.gather(Sp, 1, Y).clamp(min=1e-7)) + torch.log(torch.gather(hz, 1, Y).clamp(min=1e-7)))
        cen = -cc * torch.log(torch.gather(Sp, 1, Y + 1).clamp(min=1e-7))
        neg = (cen + unc) * torch.gather((w / w.sum()).view(1, -1).expand_as(hz), 1, Y)
        surv = (0.6 * neg + 0.4 * unc).mean()
        reg = l1 * sum(p.abs().sum() for p in cpu.values())
        (surv / gc + reg).backward()
        opt_c.step()
        losses_c.append((float(surv), float(reg)))

    # ---- this build
    model.to(DEV)
    flat = hn.train.flatten_parameters(model)
    opt = hn.train.FusedL1Adam(flat, lr=lr, l1=l1)
    ins = [tab.to(DEV), img.to(DEV)]
    for t in range(3):
        opt.zero_grad()
        out = hn.train.surv_nll_loss(model(list(ins)), y.to(DEV), c.to(DEV), weights=w.to(DEV))
        (out.loss / gc).backward()
        opt.step()
        assert abs(float(out.loss) - losses_c[t][0]) <= 2e-4 * abs(losses_c[t][0]), t
        assert abs(float(opt.reg_loss) - losses_c[t][1]) <= 1e-4 * abs(losses_c[t][1]), t
    got = model.state_dict()
    for k, v in cpu.items():
        # Adam normalises the step: tiny gradient differences can flip early updates of near-zero-gradient entries,
        # so compare against the update size (3 steps of <= lr each) rather than the parameter magnitude
        assert float((got[k].cpu() - v.detach()).abs().max()) <= 0.05 * 3 * lr, k


def test_flat_layout_survives_model_zero_grad_and_detects_reallocation(hn):
    """model.zero_grad() sets .grad to None (torch >= 2.0): the next backward then returns ordinary gradient tensors; the
    optimizer must fold them back into the flat buffer instead of stepping on zeros.  A re-allocated parameter raises."""
    torch.manual_seed(3)
    kw = dict(n_modalities=1, channel_dims=[3], num_spatial_axes=[2], out_dims=2, depth=1, l_c=8, l_d=16, x_heads=2, l_heads=2,
              cross_dim_head=8, latent_dim_head=8)
    ref = hn.HealNet(**kw).train().to(DEV)
    model = hn.HealNet(**kw).train().to(DEV)
    model.load_state_dict(ref.state_dict())
    x = torch.rand(4, 5, 6, 3, device=DEV)
    flat_r, flat_m = hn.train.flatten_parameters(ref), hn.train.flatten_parameters(model)
    opt_r, opt_m = hn.train.FusedL1Adam(flat_r, lr=1e-2), hn.train.FusedL1Adam(flat_m, lr=1e-2)
    for step in range(3):
        opt_r.zero_grad()
        ref([x]).square().sum().backward()
        opt_r.step()
        model.zero_grad()                                   # the "wrong" call: .grad becomes None
        model([x]).square().sum().backward()
        opt_m.step()
    for a, b_ in zip(ref.parameters(), model.parameters()):
        assert torch.allclose(a, b_, rtol=1e-6, atol=1e-7)
    model.latents.data = model.latents.data.clone()         # re-allocation behind the flat buffer's back
    with pytest.raises(RuntimeError, match="flat buffer"):
        opt_m.step()
