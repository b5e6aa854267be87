"""CPU restatement of the training-step tail -- TEST INFRASTRUCTURE, like oracle/healnet_cpu.py:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product never does.

Parity PINNED: tests/test_oracle_golden.py checks every function here against fixtures generated from the reference
itself (tools/gen_goldens_train.py -> tests/golden/g7_*.npz).

  surv_nll      healnet/models/survival_loss.py:9-43 (nll_loss) on hazards = sigmoid(logits), S = cumprod(1 - hazards)
                as healnet/main.py:439-447 calls it; returns the loss and (autograd) d loss / d logits
  l1_adam_step  healnet/utils/train_utils.py:5-14 (calc_reg_loss: l1 * sum |p|, gradient l1 * sign(p)) followed by
                torch.optim.Adam's single-tensor update (healnet/main.py:390, 464-467)
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch


def surv_nll(logits: torch.Tensor, y: torch.Tensor, c: torch.Tensor, weights: Optional[torch.Tensor] = None,
             alpha: float = 0.4, eps: float = 1e-7) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (loss, dloss/dlogits, hazards, survival); follows survival_loss.py:22-43 line by line."""
    logits = logits.detach().clone().requires_grad_(True)
    hazards = torch.sigmoid(logits)                                  # main.py:439
    S = torch.cumprod(1 - hazards, dim=1)                            # main.py:440
    b = len(y)
    Y = y.view(b, 1)
    cc = c.view(b, 1).float()
    S_padded = torch.cat([torch.ones_like(cc), S], 1)                # :27
    unc = -(1 - cc) * (torch.log(torch.gather(S_padded, 1, Y).clamp(min=eps)) + torch.log(torch.gather(hazards, 1, Y).clamp(min=eps)))
    cen = -cc * torch.log(torch.gather(S_padded, 1, Y + 1).clamp(min=eps))
    neg = cen + unc
    if weights is not None:                                          # :33-39
        w = (weights / torch.sum(weights)).view(1, -1).expand_as(hazards)
        neg = neg * torch.gather(w, 1, Y)
    loss = ((1 - alpha) * neg + alpha * unc).mean()                  # :41-42
    (g,) = torch.autograd.grad(loss, logits)
    return loss.detach(), g, hazards.detach(), S.detach()


def l1_adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, l1: float, lr: float,
                 beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, grad_scale: float = 1.0) -> float:
    """In place on (p, m, v); returns the reg_loss l1 * sum |p| of the parameters before the update.
    torch/optim/adam.py::_single_tensor_adam with weight_decay = 0, amsgrad = False, maximize = False."""
    reg = float(l1) * float(p.abs().sum())
    grad = g * grad_scale + l1 * torch.sign(p)                       # autograd of l1 * p.abs().sum()
    m.lerp_(grad, 1 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)
    return reg
