"""The part of ``morgana.viz`` that sits on the training path: ``synthesis.MLPG`` (called by ``predict`` of the shipped models)."""
from . import synthesis  # noqa: F401
