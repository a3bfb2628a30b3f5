"""morgana_amd - MI355X-native training hot path behind morgana's plugin surface.

``utils`` / ``losses`` / ``data`` / ``base_models`` / ``experiment_builder`` / ``lr_schedules`` mirror the reference
modules of the same names for the path ``ExperimentBuilder.train_epoch`` drives; the per-frame compute is in
``libmorgana_hip.so`` (C ABI: include/morgana_hip.h).  Importing the package does not need a GPU; calling an op does.
"""
from . import _lib  # noqa: F401
from .functional import get_precision, set_precision  # noqa: F401

__version__ = '0.1.0'
