"""pmgt_amd — MI355X-native PMGT pre-training hot path (HIP kernels behind the reference's Python surface)."""
from .configuration_pmgt import PMGTConfig  # noqa: F401

__all__ = ["PMGTConfig"]
