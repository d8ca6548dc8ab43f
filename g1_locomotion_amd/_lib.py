"""ctypes binding of the C-ABI in include/srbdqp.h (libsrbdqp.so, built in-tree by __graft_entry__.build()).

There is no Python/NumPy fallback for the hot path: if the shared library is missing, or no HIP device is
usable, this module raises.  The binding mirrors include/srbdqp.h one to one.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRBDQP_LIB: a diagnostic build of the same library (tools/build_variant.sh), never a different implementation
LIB_PATH = os.environ.get("SRBDQP_LIB") or os.path.join(_HERE, "libsrbdqp.so")

NX, NU, NC = 13, 12, 4
ROWS_PER_STEP = 20

OK = 0
E_INVALID, E_NO_DEVICE, E_HIP, E_NOMEM = -1, -2, -3, -4
SOLVED, MAX_ITER, NUMERICAL, CONTACT_BOUND = 1, 2, -1, -2
FLAG_TIMING = 1
FLAG_NO_SPIN = 2
FLAG_SETUP4 = 4
FLAG_F64_TILES = 8     # _f32 calls: fp64 tiles for every QP
FLAG_F32_TILES = 16    # _f32 calls: fp32 tiles for the eligible QPs of small batches too
FLAG_NO_LAT = 32       # staged calls on the general kernel: the batch instantiation instead of the low-latency one
FLAG_DEFER_TAIL = 64   # one-wave kernel: continuations of unconverged QPs ride in the next solve on the stream (srbdqp_flush completes them)
PENDING = 0
KERNEL_AUTO, KERNEL_COMPACT, KERNEL_SPLIT, KERNEL_WAVE, KERNEL_WRENCH = 0, 3, 4, 5, 6   # 1, 2: the retired round-1 baselines

EXPORTS = (
    "srbdqp_default_config", "srbdqp_create", "srbdqp_destroy", "srbdqp_last_error",
    "srbdqp_solve_batch_f64", "srbdqp_solve_batch_device_f64", "srbdqp_solve_batch_f32", "srbdqp_solve_batch_device_f32",
    "srbdqp_assemble_f64", "srbdqp_assemble_wrench_f64",
    "srbdqp_ragged_create", "srbdqp_ragged_destroy", "srbdqp_ragged_last_error", "srbdqp_ragged_flush", "srbdqp_solve_ragged_device_f64", "srbdqp_solve_ragged_f64",
    "srbdqp_solve_ragged_device_f32", "srbdqp_solve_ragged_f32", "srbdqp_solve_ragged_warm_device_f64", "srbdqp_solve_ragged_warm_device_f32",
    "srbdqp_set_schedule_hint", "srbdqp_flush", "srbdqp_shard_range", "srbdqp_gather_u0_f64", "srbdqp_stage_ptrs", "srbdqp_solve_staged_f64", "srbdqp_update_f64", "srbdqp_prepare_staged_f64", "srbdqp_solve_prepared_f64", "srbdqp_set_stamp_buffer", "srbdqp_synchronize", "srbdqp_last_kernel_ms", "srbdqp_last_kernel_parts_ms", "srbdqp_kernel_name", "srbdqp_batch1_launch_path", "srbdqp_version",
    # include/srbdqp_cascade.h
    "srbdqp_swing_f64", "srbdqp_swing_device_f64", "srbdqp_wbid_reference_f64", "srbdqp_wbid_reference_device_f64",
    "srbdqp_mpc_inputs_f64", "srbdqp_mpc_inputs_device_f64",
)


class SrbdqpError(RuntimeError):
    pass


class Gait(C.Structure):
    """srbdqp_gait (include/srbdqp_cascade.h)."""
    _fields_ = [("struct_size", C.c_int32), ("period_steps", C.c_int32), ("double_support_steps", C.c_int32), ("reserved0", C.c_int32),
                ("com_target", C.c_double * 3), ("hip_offset_y", C.c_double)]


class Config(C.Structure):
    """struct srbdqp_config (include/srbdqp.h)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("horizon", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32),
        ("kernel", C.c_int32), ("max_iter", C.c_int32), ("check_every", C.c_int32), ("max_contacts_per_step", C.c_int32), ("rho_restart_iter", C.c_int32), ("rho_restart_count", C.c_int32),
        ("dt", C.c_double), ("mass", C.c_double), ("inertia", C.c_double * 3), ("mu", C.c_double),
        ("fz_min", C.c_double), ("fz_max", C.c_double), ("q_diag", C.c_double * NX), ("r_diag", C.c_double),
        ("force_scale", C.c_double), ("rho", C.c_double), ("rho_eq_scale", C.c_double), ("sigma", C.c_double),
        ("alpha", C.c_double), ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("rho_fz_scale", C.c_double),
    ]


class Stage(C.Structure):
    """struct srbdqp_stage (include/srbdqp.h): host addresses of the pinned, GPU-mapped staging arrays."""
    _fields_ = [("capacity", C.c_int32), ("reserved", C.c_int32)] + [(k, C.c_void_p) for k in (
        "x0", "x_ref", "foot", "contact", "pcom", "warm_u", "warm_y", "u", "x", "y", "status", "iters")]


_lib = None


def _preload_shared_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so (soname
    libamdhip64.so.7, same as /opt/rocm's) and load it by file name, so if libsrbdqp.so pulled in the system
    copy first, a later `import torch` would map a second runtime and one of the two loses the device.  When a
    torch install is present (bench.py / tests use it for HBM buffers and torch.distributed) map ITS runtime
    first -- without importing torch -- so libsrbdqp.so's NEEDED libamdhip64.so.7 binds to it and a later
    `import torch` finds the same file already loaded.  Without torch the system runtime is used.
    """
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libsrbdqp.so (once).  Raises SrbdqpError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SrbdqpError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  g1_locomotion_amd has no CPU fallback.")
    _preload_shared_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    dp, u8p, i32p = C.c_void_p, C.c_void_p, C.c_void_p   # raw addresses: numpy or device pointers
    lib.srbdqp_default_config.argtypes = [C.POINTER(Config)]
    lib.srbdqp_default_config.restype = C.c_int
    lib.srbdqp_create.argtypes = [C.POINTER(Config), C.POINTER(H)]
    lib.srbdqp_create.restype = C.c_int
    lib.srbdqp_destroy.argtypes = [H]
    lib.srbdqp_destroy.restype = C.c_int
    lib.srbdqp_last_error.argtypes = [H]
    lib.srbdqp_last_error.restype = C.c_char_p
    lib.srbdqp_solve_batch_f64.argtypes = [H, C.c_int32, dp, dp, dp, u8p, dp, dp, dp, dp, dp, dp, i32p, i32p]
    lib.srbdqp_solve_batch_f64.restype = C.c_int
    lib.srbdqp_solve_batch_device_f64.argtypes = [H, C.c_int32, dp, dp, dp, u8p, dp, dp, dp, dp, dp, dp, i32p, i32p, C.c_void_p]
    lib.srbdqp_solve_batch_device_f64.restype = C.c_int
    lib.srbdqp_solve_batch_f32.argtypes = [H, C.c_int32, dp, dp, dp, u8p, dp, dp, dp, dp, dp, dp, i32p, i32p]
    lib.srbdqp_solve_batch_f32.restype = C.c_int
    lib.srbdqp_solve_batch_device_f32.argtypes = [H, C.c_int32, dp, dp, dp, u8p, dp, dp, dp, dp, dp, dp, i32p, i32p, C.c_void_p]
    lib.srbdqp_solve_batch_device_f32.restype = C.c_int
    lib.srbdqp_assemble_f64.argtypes = [H, C.c_int32, dp, dp, dp, u8p, dp, dp, dp, dp, dp]
    lib.srbdqp_assemble_f64.restype = C.c_int
    lib.srbdqp_assemble_wrench_f64.argtypes = [H, C.c_int32, dp, dp, dp, u8p, dp, dp, dp, dp, dp]
    lib.srbdqp_assemble_wrench_f64.restype = C.c_int
    lib.srbdqp_ragged_create.argtypes = [C.POINTER(Config), C.c_void_p, C.c_int32, C.POINTER(H)]
    lib.srbdqp_ragged_create.restype = C.c_int
    lib.srbdqp_ragged_destroy.argtypes = [H]
    lib.srbdqp_ragged_destroy.restype = C.c_int
    lib.srbdqp_ragged_flush.argtypes = [H, C.c_void_p]
    lib.srbdqp_ragged_flush.restype = C.c_int
    lib.srbdqp_ragged_last_error.argtypes = [H]
    lib.srbdqp_ragged_last_error.restype = C.c_char_p
    lib.srbdqp_solve_ragged_device_f64.argtypes = [H, C.c_int32, C.c_void_p, dp, dp, dp, u8p, dp, dp, i32p, i32p, C.c_void_p]
    lib.srbdqp_solve_ragged_device_f64.restype = C.c_int
    lib.srbdqp_solve_ragged_f64.argtypes = [H, C.c_int32, C.c_void_p, dp, dp, dp, u8p, dp, dp, i32p, i32p]
    lib.srbdqp_solve_ragged_device_f32.argtypes = [H, C.c_int32, C.c_void_p, dp, dp, dp, u8p, dp, dp, i32p, i32p, C.c_void_p]
    lib.srbdqp_solve_ragged_device_f32.restype = C.c_int
    lib.srbdqp_solve_ragged_f32.argtypes = [H, C.c_int32, C.c_void_p, dp, dp, dp, u8p, dp, dp, i32p, i32p]
    lib.srbdqp_solve_ragged_f32.restype = C.c_int
    for _fn in (lib.srbdqp_solve_ragged_warm_device_f64, lib.srbdqp_solve_ragged_warm_device_f32):
        _fn.argtypes = [H, C.c_int32, C.c_void_p, dp, dp, dp, u8p, dp, dp, dp, dp, dp, i32p, i32p, C.c_void_p]
        _fn.restype = C.c_int
    lib.srbdqp_solve_ragged_f64.restype = C.c_int
    lib.srbdqp_set_schedule_hint.argtypes = [H, C.c_void_p, C.c_int32]
    lib.srbdqp_set_schedule_hint.restype = C.c_int
    lib.srbdqp_flush.argtypes = [H, C.c_void_p]
    lib.srbdqp_flush.restype = C.c_int
    lib.srbdqp_shard_range.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.srbdqp_shard_range.restype = C.c_int
    lib.srbdqp_gather_u0_f64.argtypes = [H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.srbdqp_gather_u0_f64.restype = C.c_int
    lib.srbdqp_stage_ptrs.argtypes = [H, C.POINTER(Stage)]
    lib.srbdqp_stage_ptrs.restype = C.c_int
    lib.srbdqp_solve_staged_f64.argtypes = [H, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.srbdqp_solve_staged_f64.restype = C.c_int
    lib.srbdqp_update_f64.argtypes = [H, dp, dp, dp, u8p, dp, dp, dp, dp, i32p, i32p]
    lib.srbdqp_update_f64.restype = C.c_int
    lib.srbdqp_prepare_staged_f64.argtypes = [H, C.c_int32, C.c_int32]
    lib.srbdqp_prepare_staged_f64.restype = C.c_int
    lib.srbdqp_solve_prepared_f64.argtypes = [H, C.c_int32, C.c_int32, C.c_int32]
    lib.srbdqp_solve_prepared_f64.restype = C.c_int
    lib.srbdqp_set_stamp_buffer.argtypes = [H, C.c_void_p]
    lib.srbdqp_set_stamp_buffer.restype = C.c_int
    lib.srbdqp_synchronize.argtypes = [H]
    lib.srbdqp_synchronize.restype = C.c_int
    lib.srbdqp_last_kernel_ms.argtypes = [H]
    lib.srbdqp_last_kernel_ms.restype = C.c_double
    lib.srbdqp_last_kernel_parts_ms.argtypes = [H, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.srbdqp_last_kernel_parts_ms.restype = C.c_int
    lib.srbdqp_kernel_name.argtypes = [H]
    lib.srbdqp_kernel_name.restype = C.c_char_p
    lib.srbdqp_batch1_launch_path.argtypes = [H]
    lib.srbdqp_batch1_launch_path.restype = C.c_char_p
    lib.srbdqp_swing_f64.argtypes = [H, C.c_int64, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp]
    lib.srbdqp_swing_f64.restype = C.c_int
    lib.srbdqp_swing_device_f64.argtypes = [H, C.c_int64, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp, C.c_void_p]
    lib.srbdqp_swing_device_f64.restype = C.c_int
    lib.srbdqp_wbid_reference_f64.argtypes = [H, C.c_int64, dp, dp, dp, C.c_int32, dp, dp, dp, dp]
    lib.srbdqp_wbid_reference_f64.restype = C.c_int
    lib.srbdqp_wbid_reference_device_f64.argtypes = [H, C.c_int64, dp, dp, dp, C.c_int32, dp, dp, dp, dp, C.c_void_p]
    lib.srbdqp_wbid_reference_device_f64.restype = C.c_int
    lib.srbdqp_mpc_inputs_f64.argtypes = [H, C.c_int64, dp, dp, dp, dp, C.c_void_p, C.POINTER(Gait), dp, dp, C.c_void_p, dp, dp]
    lib.srbdqp_mpc_inputs_f64.restype = C.c_int
    lib.srbdqp_mpc_inputs_device_f64.argtypes = [H, C.c_int64, dp, dp, dp, dp, C.c_void_p, C.POINTER(Gait), dp, dp, C.c_void_p, dp, dp, C.c_void_p]
    lib.srbdqp_mpc_inputs_device_f64.restype = C.c_int
    lib.srbdqp_version.argtypes = []
    lib.srbdqp_version.restype = C.c_char_p
    _lib = lib
    return lib


_raw = None


def load_raw():
    """A second ctypes view of the same library whose functions carry NO argtypes: a call with pre-built c_void_p arguments skips the
    per-argument conversion (0.37 us instead of 0.82 us for srbdqp_update_f64's eleven arguments).  Only for call sites that bind
    every argument once as a ctypes object -- MPC.update(); an int passed here would be truncated to 32 bits."""
    global _raw
    if _raw is None:
        load()
        _raw = C.CDLL(LIB_PATH)
    return _raw


def default_config() -> Config:
    cfg = Config()
    cfg.struct_size = C.sizeof(Config)        # the library checks it BEFORE it writes (include/srbdqp.h)
    rc = load().srbdqp_default_config(C.byref(cfg))
    if rc != OK:
        raise SrbdqpError(f"srbdqp_default_config failed ({rc})")
    return cfg


def check(rc: int, handle=None):
    if rc != OK:
        msg = load().srbdqp_last_error(handle)
        raise SrbdqpError(f"srbdqp error {rc}: {msg.decode() if msg else '?'}")
