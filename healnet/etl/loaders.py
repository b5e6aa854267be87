"""``healnet.etl.loaders.MMDataset`` (reference ``healnet/etl/loaders.py:21-41``)."""
from healnet_amd.etl import MMDataset  # noqa: F401

__all__ = ["MMDataset"]
