"""Import user code that registers experiments / interfaces / datasets / agents: every ``*.py`` file below
a directory matching a pattern, or one module by path. Role of ``realhf/base/importing.py``."""
import importlib
import importlib.util
import os
import re
import sys
from typing import Optional


def import_module(path: str, pattern: Optional[re.Pattern] = None, package_root: Optional[str] = None):
    """Import every python file under `path` whose name matches `pattern` (default: all). Files inside the
    installed package are imported by dotted name so their registrations run exactly once."""
    pattern = pattern or re.compile(r".*\.py$")
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(path):
        for f in sorted(files):
            if not pattern.match(f) or f == "__init__.py":
                continue
            full = os.path.abspath(os.path.join(dirpath, f))
            if full.startswith(here + os.sep):
                rel = os.path.relpath(full, os.path.dirname(here))[:-3].replace(os.sep, ".")
                importlib.import_module(rel)
            else:
                import_usercode(full, "areal_user_" + re.sub(r"\W", "_", os.path.relpath(full, path))[:-3])


def import_usercode(module_path: str, module_name: str):
    spec = importlib.util.spec_from_file_location(module_name, module_path)
    if spec is None or spec.loader is None:
        raise ImportError(f"cannot import {module_path}")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[module_name] = mod
    spec.loader.exec_module(mod)
    return mod
