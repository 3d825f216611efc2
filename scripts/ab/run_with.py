"""Run a script of this repository with module attributes of the package overridden first (A/B runs):
    python scripts/ab/run_with.py closerlook3d_amd.fused.PW_CSR_FIRST=False bench.py --no-cpu-baseline ..."""
import importlib
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
while args and "=" in args[0] and not args[0].endswith(".py"):
    target, value = args.pop(0).split("=", 1)
    module, attr = target.rsplit(".", 1)
    setattr(importlib.import_module(module), attr, eval(value))
script = args[0]
sys.argv = args
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, run_name="__main__")
