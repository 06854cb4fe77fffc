"""iprgan — MI355X-native engine for ipr-gan's G+D training step.

Mirrors the reference's package surface for the hot path: ``iprgan.networks``, ``iprgan.models``,
``iprgan.tools``, ``iprgan.configs.Config``; compute goes through libiprgan_hip.so (include/iprgan.h).
"""
from . import attacks, configs, models, networks, optim, tools  # noqa: F401
from .configs import Config  # noqa: F401

__version__ = '0.1.0'
