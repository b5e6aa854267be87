#!/usr/bin/env python3
"""Golden fixtures for the training-step tail, generated from the REFERENCE:

  g7_surv_nll   healnet/models/survival_loss.py::nll_loss exactly as healnet/main.py:441-447 calls it
                (hazards = sigmoid(logits), S = cumprod(1 - hazards), optional class weights), with the gradient
                d loss / d logits through the reference's own autograd graph
  g8_temperature_softmax   healnet/models/healnet.py::temperature_softmax (:354-365)
  g7_l1_adam    healnet/utils/train_utils.py::calc_reg_loss (L1 over all parameters) added to a data loss, then
                torch.optim.Adam + OneCycleLR exactly as healnet/main.py:390-394,464-467 drives them, 4 steps

Runs only in the build container (imports the two reference files by path; both need torch / numpy only).
Merges its entries into tests/golden/manifest.json.

    python tools/gen_goldens_train.py
"""
import importlib.util, json, os, sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    sl = load("/root/reference/healnet/models/survival_loss.py", "ref_survival_loss")
    tu = load("/root/reference/healnet/utils/train_utils.py", "ref_train_utils")
    man_path = os.path.join(GOLD, "manifest.json")
    manifest = json.load(open(man_path))

    # ---------------------------------------------------------------- g7_surv_nll
    gen = torch.Generator().manual_seed(77)
    arrays, cases = {}, []
    specs = [(1, 4, False, 1.0), (5, 4, False, 2.0), (32, 4, True, 2.0), (32, 4, False, 2.0), (9, 7, True, 3.0), (16, 4, True, 30.0),
             (8, 2, False, 1.0)]
    for i, (b, k, weighted, spread) in enumerate(specs):
        logits = (torch.randn(b, k, generator=gen) * spread).requires_grad_(True)
        if spread >= 30.0:   # saturated sigmoids: the eps clamps of the loss become active
            with torch.no_grad():
                logits[0] = torch.tensor([60.0, -60.0, 60.0, 90.0])
                logits[1] = torch.tensor([-90.0, -60.0, -120.0, -60.0])
        y = torch.randint(0, k, (b,), generator=gen)
        c = torch.randint(0, 2, (b,), generator=gen)
        w = (torch.rand(k, generator=gen) + 0.2) if weighted else None
        hazards = torch.sigmoid(logits)
        surv = torch.cumprod(1 - hazards, dim=1)
        loss = sl.nll_loss(hazards=hazards, S=surv, Y=y, c=c, weights=w)
        (g,) = torch.autograd.grad(loss, logits)
        tag = f"c{i}_"
        arrays.update({tag + "logits": logits.detach().numpy(), tag + "y": y.numpy(), tag + "c": c.numpy(),
                       tag + "loss": np.float64(loss.item()), tag + "dlogits": g.numpy(),
                       tag + "hazards": hazards.detach().numpy(), tag + "survival": surv.detach().numpy(),
                       tag + "risk": (-surv.sum(dim=1)).detach().numpy()})
        if w is not None:
            arrays[tag + "weights"] = w.numpy()
        cases.append(dict(tag=tag, b=b, k=k, weighted=weighted))
    np.savez_compressed(os.path.join(GOLD, "g7_surv_nll.npz"), **arrays)
    manifest["g7_surv_nll"] = {"bytes": os.path.getsize(os.path.join(GOLD, "g7_surv_nll.npz")), "cases": cases,
                               "alpha": 0.4, "eps": 1e-7}

    # ---------------------------------------------------------------- g7_l1_adam
    torch.manual_seed(5)
    shapes = [(7, 5), (33,), (4, 3, 2), (1,), (130, 17)]
    params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    with torch.no_grad():
        params[1][:5] = 0.0                      # sign(0) = 0 entries
    coef = [torch.randn(*s) for s in shapes]

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList(params)

    model = Holder()
    l1, lr, max_lr, epochs, steps_per_epoch, gc = 3e-4, 1e-3, 8e-3, 2, 3, 2
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.OneCycleLR(optimizer=opt, max_lr=max_lr, epochs=epochs, steps_per_epoch=steps_per_epoch)
    arrays = {f"p{i}_init": p.detach().numpy().copy() for i, p in enumerate(params)}
    arrays.update({f"coef{i}": c.numpy() for i, c in enumerate(coef)})
    n_steps = 4
    lrs, beta1s, regs = [], [], []
    for t in range(n_steps):
        opt.zero_grad()
        lrs.append(opt.param_groups[0]["lr"])
        beta1s.append(opt.param_groups[0]["betas"][0])
        data_loss = sum(((p * c).sum() ** 2 + (p * p * c).sum()) for p, c in zip(params, coef)) * (0.1 * (t + 1))
        reg = tu.calc_reg_loss(model, l1, "healnet", ["omic", "slides"])
        regs.append(float(reg))
        total = data_loss / gc + reg + 0.0           # healnet/main.py:464
        total.backward()
        opt.step()
        opt.zero_grad()
        sched.step()
        for i, p in enumerate(params):
            arrays[f"p{i}_step{t}"] = p.detach().numpy().copy()
    arrays["lrs"] = np.array(lrs)
    arrays["beta1s"] = np.array(beta1s)
    arrays["reg_losses"] = np.array(regs)
    np.savez_compressed(os.path.join(GOLD, "g7_l1_adam.npz"), **arrays)
    manifest["g7_l1_adam"] = {"bytes": os.path.getsize(os.path.join(GOLD, "g7_l1_adam.npz")), "shapes": [list(s) for s in shapes],
                              "l1": l1, "lr": lr, "max_lr": max_lr, "epochs": epochs, "steps_per_epoch": steps_per_epoch,
                              "gc": gc, "n_steps": n_steps}
    with open(man_path, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote g7_surv_nll, g7_l1_adam")


def g8_temperature_softmax():
    """tests/golden/g8_temperature_softmax.npz: the reference's temperature_softmax (healnet/models/healnet.py:354-365)."""
    ref = load("/root/reference/healnet/models/healnet.py", "ref_healnet")
    gen = torch.Generator().manual_seed(31)
    arr = {}
    for i, (shape, T) in enumerate([((16, 128, 77), 0.5), ((3, 5, 1), 0.5), ((7, 1000), 2.0), ((4, 6, 130), 1.0)]):
        x = torch.randn(*shape, generator=gen) * 4
        arr[f"x{i}"] = x.numpy(); arr[f"y{i}"] = ref.temperature_softmax(x, temperature=T).numpy(); arr[f"t{i}"] = np.float32(T)
    x = torch.randn(5, 9, 11, generator=gen)
    arr["xd"] = x.numpy(); arr["yd"] = ref.temperature_softmax(x, temperature=0.5, dim=1).numpy()
    np.savez_compressed(os.path.join(GOLD, "g8_temperature_softmax.npz"), **arr)


if __name__ == "__main__":
    main()
    g8_temperature_softmax()
