"""ctypes binding of libldpc_hip.so (C ABI declared in include/ldpc_hip.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = 'libldpc_hip.so'

# every symbol include/ldpc_hip.h declares
SYMBOLS = (
    'ldpc_last_error', 'ldpc_device_count', 'ldpc_create_qc', 'ldpc_create_csr', 'ldpc_destroy', 'ldpc_release_pool',
    'ldpc_get_dims', 'ldpc_set_minsum_scaling', 'ldpc_set_pass_counters', 'ldpc_set_stream', 'ldpc_sync', 'ldpc_decode', 'ldpc_decode_chained',
    'ldpc_reset_state', 'ldpc_decode_device', 'ldpc_decode_device_noalloc',
    'ldpc_channel_awgn_device', 'ldpc_eval_awgn', 'ldpc_get_launch_info',
    'ldpc_comm_unique_id', 'ldpc_comm_create', 'ldpc_comm_destroy', 'ldpc_allreduce_counters',
    'ldpc_dense_to_qc',
)

_lib = None


def library_path():
    return os.environ.get('LDPC_HIP_LIBRARY', os.path.join(_HERE, LIB_NAME))


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP/HSA runtime.  A PyTorch-ROCm wheel bundles its own
    libamdhip64.so (same SONAME as /opt/rocm's); if this library pulled in the system
    copy first, a later `import torch` would load a second runtime, which then finds no
    GPU.  So when torch is installed, its copy is loaded first (without importing torch)
    and libldpc_hip.so binds to it through the shared SONAME."""
    import importlib.util
    import sys
    if 'torch' in sys.modules or os.environ.get('LDPC_HIP_SYSTEM_RUNTIME') == '1':
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), 'lib')
    for name in ('libhsa-runtime64.so', 'libamdhip64.so'):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load the HIP library or raise; never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C ldpc_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.')
    _share_hip_runtime_with_torch()
    L = ctypes.CDLL(path)
    c = ctypes
    vp, i32, i64, u64, dbl, flt = c.c_void_p, c.c_int, c.c_int64, c.c_uint64, c.c_double, c.c_float
    L.ldpc_last_error.restype = c.c_char_p
    L.ldpc_last_error.argtypes = []
    L.ldpc_device_count.restype = i32
    L.ldpc_device_count.argtypes = []
    L.ldpc_create_qc.restype = i32
    L.ldpc_create_qc.argtypes = [i32, i32, i32, i32, vp, vp, c.POINTER(vp)]
    L.ldpc_create_csr.restype = i32
    L.ldpc_create_csr.argtypes = [i32, i32, i32, vp, vp, c.POINTER(vp)]
    L.ldpc_destroy.restype = None
    L.ldpc_destroy.argtypes = [vp]
    L.ldpc_release_pool.restype = i32
    L.ldpc_release_pool.argtypes = []
    L.ldpc_get_dims.restype = i32
    L.ldpc_get_dims.argtypes = [vp, c.POINTER(i32), c.POINTER(i32), c.POINTER(i32)]
    L.ldpc_set_minsum_scaling.restype = i32
    L.ldpc_set_minsum_scaling.argtypes = [vp, ctypes.c_float, ctypes.c_float]
    L.ldpc_set_pass_counters.restype = i32
    L.ldpc_set_pass_counters.argtypes = [vp, vp]
    L.ldpc_set_stream.restype = i32
    L.ldpc_set_stream.argtypes = [vp, vp, i32]
    L.ldpc_sync.restype = i32
    L.ldpc_sync.argtypes = [vp]
    L.ldpc_decode.restype = i32
    L.ldpc_decode.argtypes = [vp, i32, i32, i32, i32, i64, vp, i32, vp, vp, vp, vp, vp]
    L.ldpc_decode_chained.restype = i32
    L.ldpc_decode_chained.argtypes = L.ldpc_decode.argtypes
    L.ldpc_reset_state.restype = i32
    L.ldpc_reset_state.argtypes = [vp]
    L.ldpc_decode_device.restype = i32
    L.ldpc_decode_device.argtypes = [vp, i32, i32, i32, i32, i64, vp, i32, flt, vp, vp, vp, vp, vp]
    L.ldpc_decode_device_noalloc.restype = i32
    L.ldpc_decode_device_noalloc.argtypes = L.ldpc_decode_device.argtypes
    L.ldpc_channel_awgn_device.restype = i32
    L.ldpc_channel_awgn_device.argtypes = [vp, dbl, u64, u64, i64, vp, vp, vp]
    L.ldpc_eval_awgn.restype = i32
    L.ldpc_eval_awgn.argtypes = [vp, i32, i32, i32, i32, i32, vp, i64, u64, u64, vp, vp]
    L.ldpc_get_launch_info.restype = i32
    L.ldpc_get_launch_info.argtypes = [vp, i32, i32, c.POINTER(i32), c.POINTER(i32), c.POINTER(i32),
                                       c.POINTER(i32)]
    L.ldpc_dense_to_qc.restype = i32
    L.ldpc_dense_to_qc.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32]
    L.ldpc_comm_unique_id.restype = i32
    L.ldpc_comm_unique_id.argtypes = [vp]
    L.ldpc_comm_create.restype = i32
    L.ldpc_comm_create.argtypes = [i32, i32, i32, vp, c.POINTER(vp)]
    L.ldpc_comm_destroy.restype = None
    L.ldpc_comm_destroy.argtypes = [vp]
    L.ldpc_allreduce_counters.restype = i32
    L.ldpc_allreduce_counters.argtypes = [vp, vp, i32]
    _lib = L
    return L
