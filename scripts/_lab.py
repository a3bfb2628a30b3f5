"""Imported first by the scripts that use experiment knobs or timing probes: those are compiled into the lab build only
(make -C morgana_amd/csrc lab -> morgana_amd/libmorgana_hip_lab.so; the product library refuses the knobs), so point the package at it
unless the caller chose a library."""
import os

_LAB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'morgana_amd', 'libmorgana_hip_lab.so')
if os.path.exists(_LAB):
    os.environ.setdefault('MORGANA_HIP_LIB', _LAB)

import sys as _sys
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _REPO not in _sys.path:
    _sys.path.insert(0, _REPO)
