"""Sibling-import shim: put this
directory first on sys.path) makes the reference's ``import utils`` resolve to the MI355X implementation."""
import sys as _sys
from pathlib import Path as _Path

_root = str(_Path(__file__).resolve().parents[2])
if _root not in _sys.path:
    _sys.path.insert(0, _root)
import vfr_amd  # noqa: E402,F401
from vfr_amd.utils import *  # noqa: E402,F401,F403
from vfr_amd import utils as _impl  # noqa: E402

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
