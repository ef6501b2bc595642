"""Loader / builder of the compiled PyTorch binding (csrc/dpc_torch.cpp -> csrc/build_torch/dpc_torch_ext.so).

The binding is the same thin layer as ops.ProjectFused (tensors <-> pointers, current stream, autograd node) in C++;
it computes nothing itself -- every number still comes out of the C-ABI library the ctypes loader holds.  `module()`
returns the imported extension or None (not built, or built from other sources): the callers then take the ctypes
path, which is the same HIP product path, only slower on the host.  `build()` compiles it in-tree with
torch.utils.cpp_extension (host compiler only; __graft_entry__.build() calls it)."""
import hashlib
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "dpc_torch.cpp")
HDR = os.path.join(ROOT, "include", "dpc_hip.h")
BUILD_DIR = os.path.join(HERE, "csrc", "build_torch")
NAME = "dpc_torch_ext"
SO = os.path.join(BUILD_DIR, NAME + ".so")
STAMP = os.path.join(BUILD_DIR, NAME + ".srchash")

_MODULE = None
_TRIED = False


def _source_hash():
    import torch
    h = hashlib.sha256()
    for p in (SRC, HDR):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(torch.__version__.encode())          # the extension is ABI-tied to the torch it was built against
    return h.hexdigest()


def build(verbose=False):
    """compile the binding in-tree (no GPU needed) and import it"""
    global _MODULE, _TRIED
    from torch.utils import cpp_extension
    os.makedirs(BUILD_DIR, exist_ok=True)
    mod = cpp_extension.load(name=NAME, sources=[SRC], build_directory=BUILD_DIR, extra_include_paths=[os.path.join(ROOT, "include")],
                             extra_cflags=["-O2"], extra_ldflags=["-ldl"], with_cuda=True, verbose=verbose)
    with open(STAMP, "w") as f:
        f.write(_source_hash())
    _MODULE, _TRIED = mod, True
    return mod


def module():
    """the imported extension, or None when it is not built (or was built from other sources / another torch)"""
    global _MODULE, _TRIED
    if _TRIED:
        return _MODULE
    _TRIED = True
    if os.environ.get("DPC_BINDING", "") == "ctypes":
        return None
    try:
        if not (os.path.exists(SO) and os.path.exists(STAMP)):
            return None
        with open(STAMP) as f:
            if f.read().strip() != _source_hash():
                return None
        import torch  # noqa: F401  (the extension links against torch's libraries)
        spec = importlib.util.spec_from_file_location(NAME, SO)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _MODULE = mod
    except Exception:  # noqa: BLE001 -- an unusable extension is not an error: the ctypes binding drives the same library
        _MODULE = None
    return _MODULE


def reset():
    """forget the cached decision (tests switch bindings through DPC_BINDING)"""
    global _MODULE, _TRIED
    _MODULE, _TRIED = None, False
