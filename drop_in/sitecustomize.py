"""Runs at interpreter start when `drop_in/` is on PYTHONPATH: puts the engine's fused operators behind the
reference's unchanged `models/backbones/resnet.py` (closerlook3d_amd/drop_in_hook.py; INTEGRATION.md level 3).
`CL3D_FUSED_OPERATORS=0` leaves the reference's own `models/local_aggregation_operators.py` in place.

A `sitecustomize` further down sys.path (distribution hooks) is still run, after this one."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)

if _REPO not in sys.path:
    sys.path.append(_REPO)  # so `closerlook3d_amd` resolves when only drop_in/ was put on PYTHONPATH

if os.environ.get("CL3D_FUSED_OPERATORS", "1") != "0":
    from closerlook3d_amd import drop_in_hook
    drop_in_hook.install()


def _chain():
    import importlib.machinery
    for entry in sys.path:
        if os.path.abspath(entry or os.getcwd()) == _HERE:
            continue
        spec = importlib.machinery.PathFinder.find_spec("sitecustomize", [entry])
        if spec is not None and spec.origin and os.path.abspath(spec.origin) != os.path.abspath(__file__):
            with open(spec.origin) as f:
                exec(compile(f.read(), spec.origin, "exec"), {"__name__": "sitecustomize", "__file__": spec.origin})
            return


try:
    _chain()
except Exception:  # a broken distribution hook must not take the interpreter down (CPython ignores it too)
    pass
