"""ctypes binding of the C ABI in include/dpc_hip.h.

The product path loads exactly one library, ``csrc/libdpc_hip.so`` (hipcc,
gfx950), and fails loudly if it is missing or if a tensor is not on a ROCm
device: there is NO CPU fallback.  (The test-suite can install the CPU
*emulation build of the same kernel source* with ``set_library`` -- see
tests/hipemu -- to check kernel logic without a GPU; nothing in this package
does that on its own.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdpc_hip.so")

DPC_COLLAPSE_DRC = 0
DPC_COLLAPSE_MAX = 1
DPC_MAX_TAPS = 63

_ERRORS = {
    -1: "DPC_E_NULL (required pointer is null)",
    -2: "DPC_E_SHAPE (non-positive or unsupported sizes)",
    -3: "DPC_E_TAPS (even or too large kernel size)",
    -4: "DPC_E_WORKSPACE (workspace too small or misaligned)",
    -5: "DPC_E_MODE (unsupported parameter combination)",
}


class DpcShape(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("Dz", ctypes.c_int32),
                ("D", ctypes.c_int32), ("Kx", ctypes.c_int32), ("Ky", ctypes.c_int32),
                ("Kz", ctypes.c_int32)]


class DpcParams(ctypes.Structure):
    _fields_ = [("camera_distance", ctypes.c_float), ("focal_length", ctypes.c_float),
                ("eps", ctypes.c_float), ("max_depth", ctypes.c_float),
                ("pose_is_quaternion", ctypes.c_int32), ("collapse_mode", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("dropout_keep", ctypes.c_int32), ("dropout_seed", ctypes.c_uint32),
                ("dropout_state", ctypes.c_void_p), ("l2_target", ctypes.c_void_p), ("l2_grad", ctypes.c_void_p),
                ("l2_weight", ctypes.c_float), ("views_per_cloud", ctypes.c_int32),
                ("sil_gt", ctypes.c_void_p), ("sil_err_parts", ctypes.c_void_p), ("sil_weight", ctypes.c_void_p),
                ("sil_dloss", ctypes.c_void_p), ("sil_proj", ctypes.c_void_p), ("sil_C", ctypes.c_int32),
                ("sil_S", ctypes.c_int32)]


_P = ctypes.c_void_p
_SP = ctypes.POINTER(DpcShape)
_PP = ctypes.POINTER(DpcParams)

# name -> (restype, argtypes); mirrors include/dpc_hip.h one to one
SIGNATURES = {
    "dpc_version": (ctypes.c_char_p, []),
    "dpc_abi_struct_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "dpc_workspace_bytes": (ctypes.c_size_t, [_SP, ctypes.c_int]),
    "dpc_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "dpc_profile_count": (ctypes.c_int, []),
    "dpc_profile_get": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                       ctypes.POINTER(ctypes.c_float)]),
    "dpc_compiled_taps": (ctypes.c_int, [ctypes.c_int]),
    "dpc_debug_copy": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t, ctypes.c_int]),
    "dpc_debug_read": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, ctypes.c_int]),
    "dpc_debug_fill": (ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.c_float, ctypes.c_int]),
    "dpc_saved_layout": (ctypes.c_int, [_SP, _PP]),
    "dpc_project_forward": (ctypes.c_int, [_P, _SP, _PP] + [_P] * 8 + [_P] * 8 + [_P, ctypes.c_size_t]),
    "dpc_project_backward": (ctypes.c_int, [_P, _SP, _PP] + [_P] * 8 + [_P] * 6 + [_P] * 3 + [_P] * 5
                             + [_P, ctypes.c_size_t]),
    "dpc_transform_fwd": (ctypes.c_int, [_P, _SP, _PP] + [_P] * 5),
    "dpc_transform_bwd": (ctypes.c_int, [_P, _SP, _PP] + [_P] * 10),
    "dpc_voxelize_fwd": (ctypes.c_int, [_P, _SP, _P, _P]),
    "dpc_voxelize_bwd": (ctypes.c_int, [_P, _SP, _P, _P, _P]),
    "dpc_voxelize_values_fwd": (ctypes.c_int, [_P, _SP, ctypes.c_int, _P, _P, _P]),
    "dpc_voxelize_values_bwd": (ctypes.c_int, [_P, _SP, ctypes.c_int, _P, _P, _P, _P, _P]),
    "dpc_blur3d": (ctypes.c_int, [_P, _SP] + [_P] * 6 + [ctypes.c_int]),
    "dpc_drc_fwd": (ctypes.c_int, [_P, _SP, _PP, _P, _P, _P, ctypes.c_int]),
    "dpc_drc_bwd": (ctypes.c_int, [_P, _SP, _PP, _P, _P, _P, _P, ctypes.c_int]),
    "dpc_max_collapse_fwd": (ctypes.c_int, [_P, _SP, _P, _P, ctypes.c_int]),
    "dpc_max_collapse_bwd": (ctypes.c_int, [_P, _SP, _P, _P, _P, ctypes.c_int]),
    "dpc_silhouette_loss_fwd": (ctypes.c_int, [_P] + [ctypes.c_int] * 4 + [_P] * 7),
    "dpc_silhouette_loss_bwd": (ctypes.c_int, [_P] + [ctypes.c_int] * 4 + [_P] * 5),
    "dpc_student_loss": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P, ctypes.c_float, _P, _P]),
    "dpc_point_index_ints": (ctypes.c_size_t, [_SP]),
    "dpc_set_chunk_sparse": (ctypes.c_int, [ctypes.c_int]),
    "dpc_set_sparse_walk": (ctypes.c_int, [ctypes.c_int]),
    "dpc_sil_parts_per_view": (ctypes.c_size_t, [_SP]),
    "dpc_silhouette_select": (ctypes.c_int, [_P] + [ctypes.c_int] * 3 + [_P] * 6),
    "dpc_nn_distance": (ctypes.c_int, [_P] + [ctypes.c_int] * 3 + [_P] * 5),
    "dpc_gauss_voxelize_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "dpc_gauss_voxelize_fwd": (ctypes.c_int, [_P] + [ctypes.c_int] * 3 + [_P, ctypes.c_float, ctypes.c_int] + [_P] * 4),
    "dpc_gauss_voxelize_bwd": (ctypes.c_int, [_P] + [ctypes.c_int] * 3 + [_P, ctypes.c_float, ctypes.c_int] + [_P] * 5
                               + [ctypes.c_size_t]),
}


class DpcError(RuntimeError):
    pass


class DpcLibrary(object):
    """A loaded C-ABI library.  ``host_memory`` marks the CPU emulation build
    (pointers are host pointers, no stream); only tests create such objects."""

    def __init__(self, path, host_memory=False):
        if not os.path.exists(path):
            raise DpcError("%s not found" % path)
        self.path = path
        self.host_memory = bool(host_memory)
        self._dll = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self._dll, name)        # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        for which, mirror in ((0, DpcShape), (1, DpcParams)):          # a stale struct mirror must not get as far as a launch
            if self.dpc_abi_struct_bytes(which) != ctypes.sizeof(mirror):
                raise DpcError("%s: sizeof(%s) is %d in the library, %d in this binding (rebuild the library)"
                               % (path, mirror.__name__, self.dpc_abi_struct_bytes(which), ctypes.sizeof(mirror)))

    def version(self):
        return self.dpc_version().decode()

    def compiled_taps(self, K):
        """smallest tap count >= K with compile-time-unrolled kernels, 0 if none (cached: sits on the per-step path)"""
        c = self.__dict__.setdefault("_compiled_taps", {})
        if K not in c:
            c[K] = int(self.dpc_compiled_taps(int(K)))
        return c[K]

    def profile(self, on):
        self.check(self.dpc_profile_enable(1 if on else 0), "dpc_profile_enable")

    def profile_records(self):
        """[(label, milliseconds)] of every launch since profile(True); syncs."""
        out = []
        for i in range(self.dpc_profile_count()):
            name, ms = ctypes.c_char_p(), ctypes.c_float()
            self.check(self.dpc_profile_get(i, ctypes.byref(name), ctypes.byref(ms)), "dpc_profile_get")
            out.append((name.value.decode(), float(ms.value)))
        return out

    def saves_xy(self, B, N, D, K, Dz=None):
        """True when the fused path keeps the xy-blurred grid (not G2) for backward at this shape
        (bit 3 of dpc_saved_layout): k_zfwd is then read-only, which bench.py's byte model needs to know."""
        shape = DpcShape(int(B), int(N), int(Dz or D), int(D), int(K), int(K), int(K))
        params = DpcParams(2.0, 1.875, 1e-5, 10.0, 1, DPC_COLLAPSE_DRC, 0, 0, 0)
        return bool(self.dpc_saved_layout(ctypes.byref(shape), ctypes.byref(params)) & 8)

    def chunk_sparse(self, B, N, D, K, Dz=None):
        """True when the fused path stores / loads only the chunks within the blur's reach of a point at this shape
        (bit 4 of dpc_saved_layout; dpc_set_chunk_sparse forces it): bench.py's byte model counts those chunks then."""
        shape = DpcShape(int(B), int(N), int(Dz or D), int(D), int(K), int(K), int(K))
        params = DpcParams(2.0, 1.875, 1e-5, 10.0, 1, DPC_COLLAPSE_DRC, 0, 0, 0)
        layout = self.dpc_saved_layout(ctypes.byref(shape), ctypes.byref(params))
        return layout >= 0 and bool(layout & 2) and bool(layout & 16)

    @staticmethod
    def check(rc, what):
        if rc == 0:
            return
        if rc < 0:
            raise DpcError("%s: %s" % (what, _ERRORS.get(rc, "error %d" % rc)))
        raise DpcError("%s: hipError_t %d" % (what, rc))


_ACTIVE = None


def get_library():
    """The product loader: libdpc_hip.so or an exception.  Never falls back."""
    global _ACTIVE
    if _ACTIVE is None:
        if not os.path.exists(LIB_PATH):
            raise DpcError(
                "HIP extension %s is not built. Build it with `python __graft_entry__.py` "
                "(or `make -C differentiable-point-clouds_amd/csrc`). There is no CPU fallback."
                % LIB_PATH)
        _ACTIVE = DpcLibrary(LIB_PATH, host_memory=False)
    return _ACTIVE


def set_library(lib):
    """Install a specific DpcLibrary; returns the previously active one (possibly None).
    `None` (re-)selects the product library.  Anything that is not a device library -- i.e. the
    CPU emulation build of the test tier -- is refused unless DPC_TEST_HOOKS=1 is set (the test
    suite's conftest sets it), so that a product process cannot end up computing on the host."""
    global _ACTIVE
    if lib is not None and getattr(lib, "host_memory", False) and os.environ.get("DPC_TEST_HOOKS") != "1":
        raise DpcError("a host-memory (emulation) library can only be installed with DPC_TEST_HOOKS=1 (tests)")
    prev, _ACTIVE = _ACTIVE, lib
    return prev
