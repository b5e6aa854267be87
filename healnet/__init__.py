"""Import-compatible front of the MI355X-native build: code written against konst-int-i/healnet keeps its import lines
(``from healnet import HealNet`` -- reference ``healnet/__init__.py:1``, ``README.md:70``).  Everything resolves to
``healnet_amd``; there is no second implementation here."""
from .models import HealNet  # noqa: F401

__all__ = ["HealNet"]
