"""ctypes binding of libnsof.so (the C ABI declared in include/nsof.h).

The shared library is the product: if it is missing or cannot be loaded this module
raises -- there is no Python / CPU fallback path.
"""
import ctypes as C
import importlib.util
import os

# The ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a
# queue run one after the other.  The pipelined host entry (nsof_farneback_u8_batch) overlaps upload, compute and
# download on three streams next to the caller's own -- with PyTorch in the process that is more than four, and the
# copies then serialise behind the kernels (measured: 1.67 k instead of 2.55 k 1080p pairs/s host to host).  The
# variable is read when the runtime initialises, i.e. at the first HIP call; an explicit setting is left alone.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NSOF_LIB", os.path.join(_HERE, "libnsof.so"))   # NSOF_LIB: A/B builds of the same ABI

OPT_POLYEXP_F32 = 1
OPT_EXACT_ROWSUMS = 2
OPT_ROW_BANDS = 3
OPT_PYR_FMA = 4
OPT_SMALL_BATCH_JOBS = 5
OPT_DEBUG_FAULT = 100   # test hook: see include/nsof.h
NSOF_OK, NSOF_EINVAL, NSOF_ESHAPE, NSOF_EDEVICE, NSOF_ENOMEM, NSOF_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
K_PREP, K_POLYEXP, K_UPSAMPLE, K_UPDMAT, K_BLUR, K_ACCUM, K_ITERATE, K_SEGMENT, K_MORPH, K_REMAP, K_SSIM, K_COUNT = range(12)

_vp, _i, _d, _f, _sz, _pd, _i64 = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_size_t, C.c_ssize_t, C.c_int64

class PairDesc(C.Structure):
    """``nsof_pair_desc`` of include/nsof.h: one frame pair of a shape-heterogeneous batch."""
    _fields_ = [("prev", _vp), ("prev_stride", _pd), ("next", _vp), ("next_stride", _pd), ("width", _i), ("height", _i),
                ("flow", _vp), ("flow_stride", _pd)]


# name -> (restype, argtypes); every symbol include/nsof.h declares
SIGNATURES = {
    "nsof_create": (_i, [_i, C.POINTER(_vp)]),
    "nsof_destroy": (None, [_vp]),
    "nsof_last_error": (C.c_char_p, [_vp]),
    "nsof_abi_version": (_i, []),
    "nsof_set_stream": (_i, [_vp, _vp]),
    "nsof_synchronize": (_i, [_vp]),
    "nsof_set_option": (_i, [_vp, _i, _i]),
    "nsof_get_option": (_i, [_vp, _i, C.POINTER(_i)]),
    "nsof_farneback_u8": (_i, [_vp, _vp, _pd, _vp, _pd, _i, _i, _vp, _pd, _d, _i, _i, _i, _i, _d, _i]),
    "nsof_farneback_u8_batch_dev": (_i, [_vp, _i, _vp, _vp, _pd, _pd, _i, _i, _vp, _d, _i, _i, _i, _i, _d, _i]),
    "nsof_farneback_u8_sequence_dev": (_i, [_vp, _i, _vp, _pd, _pd, _i, _i, _vp, _d, _i, _i, _i, _i, _d, _i]),
    "nsof_farneback_u8_batch": (_i, [_vp, _i, C.POINTER(PairDesc), _d, _i, _i, _i, _i, _d, _i]),
    "nsof_farneback_u8_batch_desc_dev": (_i, [_vp, _i, C.POINTER(PairDesc), _d, _i, _i, _i, _i, _d, _i]),
    "nsof_farneback_u8_roi_sequence_dev": (_i, [_vp, _i, _vp, _pd, _pd, _i, _i, _vp, _vp, _i, _vp, _d, _i, _i, _i, _i, _d, _i, _i,
                                                C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "nsof_host_alloc": (_vp, [_sz]),
    "nsof_host_free": (None, [_vp]),
    "nsof_farneback_effective_levels": (_i, [_i, _i, _d, _i]),
    "nsof_farneback_level_size": (_i, [_i, _i, _d, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_d)]),
    "nsof_stage_pyr_level": (_i, [_vp, _i, _vp, _pd, _pd, _i, _i, _d, _i, _vp]),
    "nsof_stage_polyexp": (_i, [_vp, _i, _vp, _i, _i, _i, _d, _vp]),
    "nsof_stage_recip": (_i, [_vp, C.c_longlong, _vp, _vp, _vp]),
    "nsof_stage_update_matrices": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp]),
    "nsof_stage_blur_solve": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "nsof_stage_iterate": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "nsof_stage_iterate_upsample": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _d, _vp]),
    "nsof_stage_flow_upsample": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _i, _d]),
    "nsof_prof_enable": (_i, [_vp, C.c_uint]),
    "nsof_prof_collect": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(C.c_longlong)]),
    "nsof_kernel_name": (C.c_char_p, [_i]),
    "nsof_accum_create": (_i, [_vp, _i, _i, _i, _i, _f, _f, C.POINTER(_vp)]),
    "nsof_accum_destroy": (None, [_vp]),
    "nsof_accum_set_dense": (_i, [_vp, _i]),
    "nsof_accum_reset": (_i, [_vp]),
    "nsof_accum_step_events": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "nsof_accum_set_events": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "nsof_accum_set_slice_times": (_i, [_vp, _vp, _vp, _i64]),
    "nsof_accum_run": (_i, [_vp, _i64, _i64, _i64]),
    "nsof_accum_surface_u8_dev": (_i, [_vp, _i, _i, _vp, _pd]),
    "nsof_accum_run_surface": (_i, [_vp, _i64, _i64, _i, _i, _vp, _pd]),
    "nsof_accum_run_frames": (_i, [_vp, _i64, _i64, _i64, _i, _i, _vp, _pd, _pd]),
    "nsof_accum_set_frames_path": (_i, [_vp, _i]),
    "nsof_accum_read_state": (_i, [_vp, _i, _vp, _vp, C.POINTER(_i64)]),
    "nsof_accum_write_state": (_i, [_vp, _i, _vp, _vp, _i64]),
    "nsof_accum_update_state_dev": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "nsof_accum_resistance_dev": (_i, [_vp, _vp, _vp, _sz]),
    "nsof_accum_bincount_2d": (_i, [_vp, _vp, _vp, _sz, _i, _i, _vp]),
    "nsof_accum_read_w": (_i, [_vp, _i, _vp]),
    "nsof_accum_read_resistance": (_i, [_vp, _i, _vp]),
    "nsof_accum_snapshot_count": (_i64, [_vp]),
    "nsof_accum_read_snapshots": (_i, [_vp, _i, _vp, _i64]),
    "nsof_accum_block_current": (_i, [_vp, _i, _i64, _i, _d, _vp]),
    "nsof_accum_block_current_dev": (_i, [_vp, _i, _i64, _i, _d, _vp]),
    "nsof_accum_frames_f64": (_i, [_vp, _vp, _i, _i, _i, _d, _i, _d, _d, _vp, _vp]),
    "nsof_accum_slice_bounds": (_i64, [_vp, _i64, _i64, _vp, _i64]),
    "nsof_roi_from_surface": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i]),
    "nsof_roi_from_surface_dev": (_i, [_vp, _vp, _i, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "nsof_structuring_element": (_i, [_i, _i, _i, _vp]),
    "nsof_morph_binary_u8_dev": (_i, [_vp, _i, _vp, _pd, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _pd]),
    "nsof_motion_mask_dev": (_i, [_vp, _vp, _pd, _i, _i, _d, _i, _i, _vp, _pd]),
    "nsof_motion_mask": (_i, [_vp, _vp, _pd, _i, _i, _d, _i, _i, _vp, _pd]),
    "nsof_gray_u8_dev": (_i, [_vp, _vp, _pd, _i, _i, _i, _vp, _pd]),
    "nsof_remap_linear_u8_dev": (_i, [_vp, _vp, _pd, _i, _i, _i, _vp, _pd, _vp, _pd, _i, _i, _i, _i, _vp, _pd]),
    "nsof_predict_warp_u8_dev": (_i, [_vp, _vp, _pd, _i, _i, _i, _vp, _pd, _i, _i, _i, _i, _i, _i, _vp, _pd]),
    "nsof_predict_warp_u8": (_i, [_vp, _vp, _pd, _i, _i, _i, _vp, _pd, _i, _i, _i, _i, _i, _i, _vp, _pd]),
    "nsof_ssim_u8_dev": (_i, [_vp, _vp, _pd, _i, _vp, _pd, _i, _i, _i, _d, C.POINTER(_d)]),
    "nsof_ssim_u8": (_i, [_vp, _vp, _pd, _i, _vp, _pd, _i, _i, _i, _d, C.POINTER(_d)]),
}

_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (soname
    libamdhip64.so.7, same as the system ROCm one libnsof.so is linked against).  If libnsof.so pulled in the
    system runtime first and torch were imported later, the process would hold two HIP/HSA runtimes and the
    second one finds no GPU.  Loading torch's copy first makes the dynamic loader satisfy libnsof.so's
    DT_NEEDED from it (soname match), whichever order the caller imports things in.
    NSOF_HIP_RUNTIME=system skips this (for processes that never import torch)."""
    if os.environ.get("NSOF_HIP_RUNTIME", "") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load libnsof.so (once).  Raises OSError with a build hint if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(
                f"{LIB_PATH} not found: build it with `make -C {os.path.dirname(_HERE)}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`); nsof has no CPU fallback")
        _preload_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here == ABI mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
