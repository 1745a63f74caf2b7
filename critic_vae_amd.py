"""Import shim: the package directory is ``critic-vae_amd/`` (hyphen, fixed by the project
layout), which Python cannot import by name.  ``import critic_vae_amd`` lands here and is
re-pointed at that directory as a regular package."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "critic-vae_amd")
_spec = importlib.util.spec_from_file_location(
    "critic_vae_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["critic_vae_amd"] = _mod
_spec.loader.exec_module(_mod)
