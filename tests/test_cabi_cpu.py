"""CPU-side tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/srbdqp.h declares,
validates its inputs, and FAILS LOUDLY (no CPU fallback) when there is no GPU.  No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import srbd_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    """Every function declared in include/*.h (comments stripped)."""
    names = set()
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if hdr.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", hdr)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(srbdqp_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol(built_lib):
    from g1_locomotion_amd import _lib
    decl = _declared_functions()
    assert len(decl) >= 10
    for name in decl:
        assert hasattr(built_lib, name), f"{name} declared in include/*.h but not exported"
    assert set(_lib.EXPORTS) == set(decl), (set(_lib.EXPORTS) ^ set(decl))


def test_config_struct_and_defaults_match_the_oracle(built_lib):
    from g1_locomotion_amd import _lib
    cfg = _lib.default_config()
    assert cfg.struct_size == C.sizeof(_lib.Config)
    p = orc.SrbdParams()
    assert cfg.horizon == 10 and cfg.max_iter == p.max_iter and cfg.check_every == p.check_every and cfg.rho_restart_iter == p.rho_restart_iter
    assert cfg.rho_restart_count == 0 and orc.default_restart(10) == (55, 2) and orc.default_restart(4) == (55, 2)   # 0 = automatic (srbdqp.h): one rule for every kernel and batch size
    assert orc.default_restart(12) == (70, 2) and orc.default_restart(16) == (80, 3) and orc.default_restart(20) == (125, 1) and orc.default_restart(24) == (100, 2)
    for k in ("dt", "mass", "mu", "fz_min", "fz_max", "r_diag", "force_scale", "rho_eq_scale", "sigma", "alpha", "eps_abs", "eps_rel"):
        assert getattr(cfg, k) == getattr(p, k), k
    assert cfg.rho == 0.0 and orc.auto_rho(cfg.horizon) == p.rho          # 0 = chosen from the horizon (oracle auto_rho)
    assert cfg.rho_fz_scale == 0.0 and orc.auto_rho_fz_scale(cfg.horizon) == p.rho_fz_scale
    assert tuple(cfg.inertia) == tuple(p.inertia) and tuple(cfg.q_diag) == tuple(p.q_diag)
    # the header's status / error codes are the ones the Python side and the oracle use
    hdr = open(os.path.join(ROOT, "include", "srbdqp.h")).read()
    for name, val in (("SRBDQP_SOLVED", orc.STATUS_SOLVED), ("SRBDQP_MAX_ITER", orc.STATUS_MAX_ITER), ("SRBDQP_NUMERICAL", orc.STATUS_NUMERICAL),
                      ("SRBDQP_CONTACT_BOUND", _lib.CONTACT_BOUND), ("SRBDQP_E_NO_DEVICE", _lib.E_NO_DEVICE)):
        m = re.search(rf"#define\s+{name}\s+\(?(-?\d+)\)?", hdr)
        assert m and int(m.group(1)) == val, name


def test_create_rejects_bad_configs_before_touching_the_device(built_lib):
    from g1_locomotion_amd import _lib
    for mutate in (lambda c: setattr(c, "horizon", 7), lambda c: setattr(c, "struct_size", 8), lambda c: setattr(c, "rho", -1.0),
                   lambda c: setattr(c, "max_iter", 0), lambda c: setattr(c, "max_contacts_per_step", 5), lambda c: c.q_diag.__setitem__(0, -1.0),
                   lambda c: c.inertia.__setitem__(1, 0.0), lambda c: setattr(c, "alpha", 2.5), lambda c: setattr(c, "eps_abs", -1e-6),
                   lambda c: setattr(c, "fz_min", 2000.0), lambda c: setattr(c, "r_diag", -1.0), lambda c: setattr(c, "eps_rel", float("nan"))):
        cfg = _lib.default_config()
        mutate(cfg)
        h = C.c_void_p()
        assert built_lib.srbdqp_create(C.byref(cfg), C.byref(h)) == _lib.E_INVALID
        assert not h.value and built_lib.srbdqp_last_error(None)
    assert built_lib.srbdqp_create(None, None) == _lib.E_INVALID
    assert built_lib.srbdqp_destroy(None) == _lib.OK
    assert built_lib.srbdqp_solve_batch_f64(None, 1, *([None] * 12)) == _lib.E_INVALID


def test_no_gpu_means_a_loud_failure_not_a_cpu_fallback(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    from g1_locomotion_amd import BatchMPC, MPC, SrbdqpError, _lib
    cfg = _lib.default_config()
    h = C.c_void_p()
    assert built_lib.srbdqp_create(C.byref(cfg), C.byref(h)) == _lib.E_NO_DEVICE
    with pytest.raises(SrbdqpError, match="no usable HIP device"):
        BatchMPC(horizon=10)
    m = MPC(dt=0.04)
    with pytest.raises(SrbdqpError):
        m.init_matrices()


def test_missing_library_is_a_loud_failure(monkeypatch, tmp_path):
    from g1_locomotion_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libsrbdqp.so"))
    with pytest.raises(_lib.SrbdqpError, match="no CPU fallback"):
        _lib.load()


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under g1_locomotion_amd/ may import, link or execute it."""
    pkg = os.path.join(ROOT, "g1_locomotion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "srbd_oracle" not in txt.replace("oracle/srbd_oracle.py", "").replace("oracle admm_solve", "") or f.endswith((".hpp", ".hip")), f
                assert "import c_oracle" not in txt and "libsrbd_oracle" not in txt, f


def test_mpc_object_keeps_the_reference_surface():
    """Attributes and shapes the reference's caller touches (g1_mujoco_sim/src/run_simulation.py:73-82,96,103,169-170)."""
    from g1_locomotion_amd import mpc
    MPC = mpc.MPC(dt=0.04)
    assert MPC.x0.shape == (13, 1) and MPC.x_ref_hor.shape == (MPC.HORIZON_LENGTH, 13) and MPC.HORIZON_LENGTH == 10
    assert MPC.g == -9.80665 and MPC.x0[12] == MPC.g and np.all(MPC.x_ref_hor[:, -1] == MPC.g)
    MPC.x0[0:3] = np.zeros((3, 1)); MPC.x_ref_hor[0, :] = MPC.x0[:].copy().reshape(13)       # the caller's own statements
    MPC.x_ref_hor[0:, 3:6] = [5.26790425e-02, 7.44339342e-05, 5.97983255e-01]
    for name in ("init_matrices", "update", "solve"):
        assert callable(getattr(MPC, name))


def _stub():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ctypes_stub", os.path.join(ROOT, "examples", "ctypes_stub.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_documented_ctypes_stub_matches_the_library(built_lib):
    """INTEGRATION.md section 2 shows examples/ctypes_stub.py verbatim: its struct must be the library's struct, field for field, and
    the document must really contain the file."""
    from g1_locomotion_amd import _lib
    stub = _stub()
    lib = stub.load()
    cfg = stub.default_config(lib)
    assert C.sizeof(stub.Config) == cfg.struct_size == C.sizeof(_lib.Config)
    assert [(n, t) for n, t in stub.Config._fields_] == [(n, t) for n, t in _lib.Config._fields_]
    full = _lib.default_config()
    assert bytes(cfg) == bytes(full)
    # the header's struct, field by field, in order
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "srbdqp.h")).read(), flags=re.S)
    body = re.search(r"typedef struct srbdqp_config \{(.*?)\} srbdqp_config;", hdr, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            for part in decl.split(None, 1)[1].split(","):
                names.append(re.match(r"\s*(\w+)", part).group(1))
    assert names == [n for n, _ in stub.Config._fields_]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(os.path.join(ROOT, "examples", "ctypes_stub.py")).read()
    assert src.strip() in doc, "INTEGRATION.md section 2 no longer shows examples/ctypes_stub.py verbatim"


def test_default_config_refuses_a_struct_of_another_size(built_lib):
    """A binding built against an older header passes a shorter struct: nothing may be written past it."""
    from g1_locomotion_amd import _lib
    buf = (C.c_char * (C.sizeof(_lib.Config) + 64))()
    C.memset(buf, 0x5A, len(buf))
    cfg = C.cast(buf, C.POINTER(_lib.Config))
    for claimed in (C.sizeof(_lib.Config) - 8, 0, 8, C.sizeof(_lib.Config) + 8):
        cfg.contents.struct_size = claimed
        assert built_lib.srbdqp_default_config(cfg) == _lib.E_INVALID
        assert bytes(buf)[4:] == b"\x5a" * (len(buf) - 4), "srbdqp_default_config wrote into a struct it refused"
    cfg.contents.struct_size = C.sizeof(_lib.Config)
    assert built_lib.srbdqp_default_config(cfg) == _lib.OK
    assert bytes(buf)[C.sizeof(_lib.Config):] == b"\x5a" * 64


def test_shard_range_equals_the_python_rule(built_lib):
    lib = built_lib
    """srbdqp_shard_range (the C consumer's view of SURVEY row e) cuts a fleet exactly as g1_locomotion_amd.shard.shard_bounds does; bad requests are refused."""
    import ctypes as C
    from g1_locomotion_amd.shard import shard_bounds
    first, count = C.c_int64(), C.c_int64()
    for total in (0, 1, 7, 4096, 524288, 524291):
        for world in (1, 2, 3, 8):
            covered = 0
            for rank in range(world):
                assert lib.srbdqp_shard_range(total, world, rank, C.byref(first), C.byref(count)) == 0
                lo, hi = shard_bounds(total, rank, world)
                assert (first.value, first.value + count.value) == (lo, hi)
                covered += count.value
            assert covered == total
    assert lib.srbdqp_shard_range(10, 0, 0, C.byref(first), C.byref(count)) != 0
    assert lib.srbdqp_shard_range(10, 2, 2, C.byref(first), C.byref(count)) != 0
    assert lib.srbdqp_shard_range(-1, 2, 0, C.byref(first), C.byref(count)) != 0
