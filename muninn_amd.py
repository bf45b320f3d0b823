"""Import helper: the package directory is ``sqlite-muninn_amd`` (hyphenated project name), which
Python cannot import by name.  ``import muninn_amd`` registers it as ``sqlite_muninn_amd``."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(_ROOT, "sqlite-muninn_amd")


def load():
    if "sqlite_muninn_amd" in sys.modules:
        return sys.modules["sqlite_muninn_amd"]
    spec = importlib.util.spec_from_file_location("sqlite_muninn_amd", os.path.join(_PKG, "__init__.py"),
                                                  submodule_search_locations=[_PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["sqlite_muninn_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


pkg = load()
