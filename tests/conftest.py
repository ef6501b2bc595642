import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


os.environ.setdefault("DPC_TEST_HOOKS", "1")         # allow installing the CPU emulation library (refused otherwise)
os.environ.setdefault("DPC_POISON_BUFFERS", "1")   # NaN-fill kernel buffers: unwritten reads cannot hide


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the suites need the in-tree native libraries; build them if this is a fresh checkout
    # (hipcc cross-compiles gfx950 without a GPU; on the GPU box the prebuilt .so travels along)
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "differentiable-point-clouds_amd", "csrc", "libdpc_hip.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.dirname(lib)])


@pytest.fixture(scope="session", autouse=True)
def _variant_gpu_library():
    """kernel experiments: DPC_GPU_LIB=<csrc/libdpc_x.so> runs the -m gpu tests against a variant build of the HIP library
    (through the ctypes binding: the compiled one opens the shipped file)."""
    path = os.environ.get("DPC_GPU_LIB")
    if not path:
        yield
        return
    os.environ["DPC_BINDING"] = "ctypes"
    import dpc_amd
    prev = dpc_amd._capi.set_library(dpc_amd._capi.DpcLibrary(os.path.abspath(path)))
    yield
    dpc_amd._capi.set_library(prev)


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def emu_library():
    """CPU emulation build of the kernel source (tests/hipemu): same .hip file,
    compiled with g++ against a thread-per-lane HIP model.  Test-only."""
    import subprocess
    emu_dir = os.path.join(ROOT, "tests", "hipemu")
    subprocess.check_call(["make", "-s", "-j8", "-C", emu_dir])
    import dpc_amd
    # (kernel experiments: DPC_EMU_LIB names a variant built with `make -C tests/hipemu OUT=... EXTRA=-D...`)
    name = os.environ.get("DPC_EMU_LIB", "libdpc_emu.so")
    lib = dpc_amd._capi.DpcLibrary(os.path.join(emu_dir, name), host_memory=True)
    # emulation-only hook: dead groups taken by the z kernels' wavefronts since the last call (the sparse walk's tests)
    import ctypes
    lib.dpc_emu_dead_groups_take = lib._dll.dpc_emu_dead_groups_take
    lib.dpc_emu_dead_groups_take.restype = ctypes.c_longlong
    lib.dpc_emu_dead_groups_take.argtypes = []
    lib.dpc_emu_deals_take = lib._dll.dpc_emu_deals_take         # ... wavefronts dealt another tile of their work-group (zdeal_tiles)
    lib.dpc_emu_deals_take.restype = ctypes.c_longlong
    lib.dpc_emu_deals_take.argtypes = []
    return lib


@pytest.fixture()
def emu(emu_library):
    import dpc_amd
    prev = dpc_amd._capi.set_library(emu_library)
    yield emu_library
    dpc_amd._capi.set_library(prev)


@pytest.fixture(params=[1, 2], ids=["poison_ff", "poison_marks_00"])
def poison_mode(request, monkeypatch):
    """DPC_POISON_BUFFERS modes 1 and 2 (ops._poison_mode): 0xff over everything the kernels must define -- and, mode 2, 0x00
    over the point index whose tail holds the chunk marks.  A 0xff mark reads 'this chunk is there', so a consumer of marks
    nobody wrote passes under mode 1 and drops chunks under mode 2."""
    import dpc_amd
    monkeypatch.setattr(dpc_amd.ops, "_POISON", request.param)
    return request.param
