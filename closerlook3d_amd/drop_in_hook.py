"""Import hook that puts the engine's operators behind the reference's UNCHANGED `models/backbones/resnet.py`.

`resnet.py:3` binds its operator class with a relative import, `from ..local_aggregation_operators import
LocalAggregation`, so a path entry cannot redirect it (a package's submodule is looked up in the package's own
directory).  `install()` adds one `sys.meta_path` finder that answers the import of
`models.local_aggregation_operators` with a module whose names are those of
`closerlook3d_amd.local_aggregation_operators` (same classes, constructor / forward signatures and state-dict keys
as the reference's file, reference models/local_aggregation_operators.py:16-464).  Nothing else is intercepted and
nothing heavy is imported until that import happens.

`drop_in/sitecustomize.py` calls `install()` at interpreter start when `drop_in/` is on PYTHONPATH (INTEGRATION.md,
level 3); `CL3D_FUSED_OPERATORS=0` keeps the reference's own operator file (levels 1-2 only).
"""
import importlib
import importlib.abc
import importlib.util
import sys

TARGET = "models.local_aggregation_operators"
SOURCE = "closerlook3d_amd.local_aggregation_operators"


class _AliasLoader(importlib.abc.Loader):
    def create_module(self, spec):
        return None

    def exec_module(self, module):
        src = importlib.import_module(SOURCE)
        for name, value in vars(src).items():
            if not name.startswith("__"):
                setattr(module, name, value)
        module.__doc__ = src.__doc__
        module.__cl3d_engine__ = SOURCE


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != TARGET:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(), origin=SOURCE)


def install():
    """Idempotent.  Returns True when the finder was added by this call."""
    if any(isinstance(f, _Finder) for f in sys.meta_path):
        return False
    if TARGET in sys.modules and not getattr(sys.modules[TARGET], "__cl3d_engine__", None):
        raise RuntimeError(f"{TARGET} is already imported from the reference tree; install the hook first")
    sys.meta_path.insert(0, _Finder())
    return True


def uninstall():
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, _Finder)]
