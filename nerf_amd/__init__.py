"""nerf_amd — MI355X-native volume renderer with the call surface of brandontrabucco/nerf."""
from .model import NeRF  # noqa: F401

__all__ = ["NeRF"]
