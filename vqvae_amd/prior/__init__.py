"""Autoregressive prior over geodesic codes: the consumer of codes.npy, trained data-parallel."""
from .transformer import Transformer  # noqa: F401
