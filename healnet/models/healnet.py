"""``healnet.models.healnet`` (reference ``healnet/models/healnet.py``): the same public names, bound to the HIP-backed classes and
functions of ``healnet_amd.healnet``."""
from healnet_amd.healnet import (GELU, SELU, Attention, FeedForward, HealNet, PreNorm, cache_fn, default, exists,  # noqa: F401
                                 fourier_encode, temperature_softmax)

__all__ = ["HealNet", "Attention", "PreNorm", "FeedForward", "GELU", "SELU", "cache_fn", "default", "exists", "fourier_encode",
           "temperature_softmax"]
