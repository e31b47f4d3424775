"""Import alias: ``import vfr_amd`` loads the package directory ``video-fragments-retrieval_amd/``.

The directory name is fixed by the project layout and is not a valid Python identifier, so this
one-file shim registers it under the importable name ``vfr_amd`` (sub-modules resolve normally:
``from vfr_amd import models, evaluate``).
"""
import importlib.util as _ilu
import pathlib as _pl
import sys as _sys

_dir = _pl.Path(__file__).resolve().parent / "video-fragments-retrieval_amd"
_spec = _ilu.spec_from_file_location("vfr_amd", _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["vfr_amd"] = _mod
_spec.loader.exec_module(_mod)
