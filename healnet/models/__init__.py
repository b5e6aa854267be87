"""``from healnet.models import HealNet, Attention`` (reference ``healnet/models/__init__.py:1-11``).  The reference's list also
names ``FCNN`` (a baseline) and the survival-loss classes; the baselines are outside this build's scope and raise a
clear ImportError-style message on access instead of failing the whole package import."""
from healnet_amd.healnet import HealNet, Attention  # noqa: F401
from healnet_amd.train import surv_nll_loss  # noqa: F401  (fused form of survival_loss.nll_loss: SURVEY.md 8 f1)

__all__ = ["HealNet", "Attention", "surv_nll_loss"]

_OUT_OF_SCOPE = {"FCNN": "healnet.baselines (not part of the fusion hot path)",
                 "CrossEntropySurvLoss": "use healnet.models.surv_nll_loss (healnet_amd.train.surv_nll_loss), the fused NLL survival loss",
                 "CoxPHSurvLoss": "the Cox partial-likelihood loss is not part of the fusion hot path"}


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise AttributeError(f"healnet.models.{name} is not provided by the MI355X-native build: {_OUT_OF_SCOPE[name]}")
    raise AttributeError(f"module 'healnet.models' has no attribute {name!r}")
