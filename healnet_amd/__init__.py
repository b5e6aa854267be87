"""healnet_amd -- MI355X-native (gfx950) implementation of HEALNet's iterative cross-attention
fusion stack, a drop-in for ``healnet.models.HealNet`` / ``healnet.models.Attention`` of
konst-int-i/healnet.  ``import healnet_amd as healnet`` keeps ``from healnet import HealNet`` style code working.
"""
from .healnet import (GELU, SELU, Attention, FeedForward, HealNet, PreNorm, cache_fn, default, exists, fourier_encode,
                      fourier_encode_concat, latent_block, temperature_softmax)
from .etl import MMDataset
from . import ops as _ops  # noqa: F401  (registers torch.ops.healnet_hip.*)
from . import train  # noqa: F401  (survival loss + fused L1/Adam step)

__all__ = ["HealNet", "Attention", "PreNorm", "FeedForward", "MMDataset", "fourier_encode_concat", "temperature_softmax", "fourier_encode",
           "GELU", "SELU", "cache_fn", "exists", "default", "latent_block"]
__version__ = "0.1.0"
