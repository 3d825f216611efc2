"""Recipe: compile the REFERENCE's own native extension for gfx950 into oracle/_ref/ref_ext.so.

Test infrastructure only.  Runs where /root/reference exists (the build container); the GPU box
only ever sees the prebuilt ``oracle/_ref/ref_ext.so`` (git-ignored, travels with gpurun).

What it does, and what it does not:
  * the sources are compiled *from where they lie* under /root/reference; because that tree is
    read-only and PyTorch's CUDA->HIP source pass writes its output next to its input, the recipe
    stages a throw-away copy under a temp dir outside the repo, never inside it;
  * it drives ``torch.utils.cpp_extension.load`` -- the stock PyTorch-ROCm way to build any
    ``CUDAExtension`` (it is what ``python setup.py install`` of the reference does on a ROCm
    machine): PyTorch's bundled hipify pass + hipcc --offload-arch=gfx950 -O2 (the reference's own
    optimisation level, setup.py:27-28) against the real rocThrust / ATen headers of this image.
    No header, library or tool is stood in for; the reference's setup.py is not executed;
  * nothing but the resulting shared object (and ninja's intermediates, deleted afterwards) is
    written to oracle/_ref/.

The module exposes the reference's five functions (bindings.cpp:8-14) on CUDA(HIP) tensors.
"""
import glob
import os
import shutil
import sys
import tempfile

REF_EXT_SRC = "/root/reference/pytorch/ops/pt_custom_ops/_ext_src"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")


def available():
    return os.path.isdir(REF_EXT_SRC)


def build(verbose=False):
    if not available():
        raise RuntimeError("reference tree not present; oracle/_ref can only be built in the build container")
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    os.environ.setdefault("MAX_JOBS", "8")
    from torch.utils import cpp_extension as ce

    os.makedirs(OUT_DIR, exist_ok=True)
    stage = tempfile.mkdtemp(prefix="cl3d_ref_stage_")
    try:
        shutil.copytree(REF_EXT_SRC, os.path.join(stage, "_ext_src"))
        root = os.path.join(stage, "_ext_src")
        srcs = sorted(glob.glob(os.path.join(root, "src", "*.cpp")) + glob.glob(os.path.join(root, "src", "*.cu")))
        bdir = os.path.join(stage, "build")
        os.makedirs(bdir)
        ce.load(name="ref_ext", sources=srcs, extra_include_paths=[os.path.join(root, "include")],
                extra_cflags=["-O2"], extra_cuda_cflags=["-O2"], build_directory=bdir,
                verbose=verbose, is_python_module=False)
        shutil.copy2(os.path.join(bdir, "ref_ext.so"), os.path.join(OUT_DIR, "ref_ext.so"))
    finally:
        shutil.rmtree(stage, ignore_errors=True)
    return os.path.join(OUT_DIR, "ref_ext.so")


REF_GRID_SRC = "/root/reference/pytorch/ops/cpp_wrappers"


def grid_available():
    return os.path.isdir(os.path.join(REF_GRID_SRC, "cpp_subsampling"))


def build_grid():
    """oracle/_ref/libgrid_dataset_ref.so: the reference's dataset-side grid_subsampling.cpp + cloud.cpp (compiled
    from where they lie, g++ -std=c++11 like the reference's setup.py) behind oracle/grid_dataset_shim.cpp."""
    import subprocess
    if not grid_available():
        raise RuntimeError("reference tree not present")
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, "libgrid_dataset_ref.so")
    sub = os.path.join(REF_GRID_SRC, "cpp_subsampling")
    cmd = ["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-I", sub,
           os.path.join(HERE, "grid_dataset_shim.cpp"), os.path.join(sub, "grid_subsampling", "grid_subsampling.cpp"),
           os.path.join(REF_GRID_SRC, "cpp_utils", "cloud", "cloud.cpp"), "-o", out]
    subprocess.run(cmd, check=True)
    return out


def load_grid():
    """ctypes handle of oracle/_ref/libgrid_dataset_ref.so (CPU code: runs anywhere the file travels to)."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(OUT_DIR, "libgrid_dataset_ref.so"))
    lib.cl3d_ref_dataset_grid_subsampling.restype = ctypes.c_int
    lib.cl3d_ref_dataset_grid_subsampling.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_float] + \
        [ctypes.c_void_p] * 3
    return lib


def load():
    """Import oracle/_ref/ref_ext.so as a Python module (needs torch; GPU needed to call it)."""
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)

    so = os.path.join(OUT_DIR, "ref_ext.so")
    if not os.path.exists(so):
        return None
    spec = importlib.util.spec_from_file_location("ref_ext", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
