"""Import shim: the package directory is named ``contact-human-dynamics_amd`` (not a
valid identifier), so ``import chd_amd`` exposes it under an importable name."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module('contact-human-dynamics_amd')
sys.modules[__name__] = _pkg
