"""``from healnet.etl import MMDataset`` (reference ``healnet/etl/__init__.py:1``, ``README.md:71``).  ``TCGADataset`` (the TCGA
file-system pipeline) is outside the hot path; ``DeviceLoader`` is this build's input staging."""
from healnet_amd.etl import MMDataset, DeviceLoader, bag_padding_mask  # noqa: F401

__all__ = ["MMDataset", "DeviceLoader", "bag_padding_mask"]
