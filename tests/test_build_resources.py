"""Scratch memory of the kernels that ship by default (round-4 verdict, item 3a): parsed from the -Rpass-analysis=kernel-resource-usage remarks of the build
(__graft_entry__.build() and tools/build.sh write them to $TMPDIR/srbdqp_build.log; tools/resource_table.py prints the table).  A kernel that spills keeps
per-lane values in scratch memory -- HBM traffic the algorithm does not have (the N = 12 bucket of configs[4] moved 5.3 x its algorithmic bytes in round 4).

Every kernel a default-configured handle can launch must have ScratchSize 0, except the ones listed in KNOWN with the bytes they are allowed (the fp32 iterations
on fp64 tiles -- the mixed-gait QPs of an _f32 call -- and the N = 24 / N = 20 kernels: DESIGN.md section 9); a listed kernel that gets worse, or a new kernel
that spills, fails the test.  The assembly-dump instantiations (MODE = 1 / DUMP) only serve the parity tests and are not held to it."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# kernel (as tools/resource_table.py prints it) -> scratch bytes per lane it may use
KNOWN = {
    "srbdqp_compact_kernel<12, 2, false, false, false>": 20,
    "srbdqp_wrench_kernel<8, float, float, 0, 3, double, 5, 0>": 20,
    "srbdqp_wrench_kernel<12, float, float, 0, 3, double, 5, 0>": 116,
    "srbdqp_wrench_kernel<16, float, float, 0, 3, double, 5, 0>": 228,
    "srbdqp_wrench_kernel<10, float, float, 0, 3, float, 5, 0>": 28,
    "srbdqp_wrench_kernel<24, float, float, 0, 3, float, 5, 0>": 64,
    "srbdqp_wrench_kernel<24, float, float, 0, 2, double, 5, 3>": 92,
    "srbdqp_wrench_kernel<24, double, double, 0, 1, double, 5, 3>": 20,
}


def _is_dump(name):
    m = re.match(r"srbdqp_wrench_kernel<\d+, \w+, \w+, (\d)", name)
    if m and m.group(1) == "1":
        return True                                             # MODE = 1: srbdqp_assemble_wrench_f64
    m = re.match(r"srbdqp_setup1_kernel<\d+, \d+, \w+, (\w+)", name)
    if m and m.group(1) == "true":
        return True                                             # DUMP: srbdqp_assemble_f64 on the one-wave kernel
    m = re.match(r"srbdqp_compact_kernel<\d+, \d+, \w+, (\w+)", name)
    return bool(m and m.group(1) == "true")


@pytest.fixture(scope="module")
def rows():
    import resource_table
    log = os.path.join(os.environ.get("TMPDIR", "/tmp"), "srbdqp_build.log")
    lib = os.path.join(ROOT, "g1_locomotion_amd", "libsrbdqp.so")
    src = os.path.join(ROOT, "g1_locomotion_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith((".hip", ".hpp")))
    if not (os.path.exists(log) and os.path.exists(lib) and os.path.getmtime(log) >= newest and "Function Name" in open(log).read()):
        # no log of the current sources (e.g. the library came with the snapshot): compile the device code once more for its remarks (about two minutes)
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-c", "--cuda-device-only", "-o", os.devnull,
               os.path.join(src, "srbdqp.hip"), "-Rpass-analysis=kernel-resource-usage"]
        with open(log, "w") as lf:
            subprocess.check_call(cmd, stderr=lf)
    return resource_table.parse(log)


def test_the_kernels_a_default_handle_launches_keep_nothing_in_scratch_memory(rows):
    assert len(rows) > 90, "the build log does not hold the kernels' resource remarks"
    bad, seen = [], set()
    for r in rows:
        name = r["name"].strip()
        seen.add(name)
        if _is_dump(name):
            continue
        allowed = KNOWN.get(name, 0)
        if r["scratch"] > allowed:
            bad.append((name, r["scratch"], allowed))
    assert not bad, "kernels with more scratch memory than allowed (bytes per lane, allowed): %r" % bad
    # the list must not rot: a kernel that no longer spills (or no longer exists) leaves it
    stale = [k for k in KNOWN if k not in seen or next(r["scratch"] for r in rows if r["name"].strip() == k) == 0]
    assert not stale, "KNOWN lists kernels that no longer spill: %r" % stale


def test_the_headline_kernels_are_at_two_waves_per_simd_without_scratch(rows):
    by = {r["name"].strip(): r for r in rows}
    for k in ("srbdqp_wave_defer_kernel<10, 2, false>", "srbdqp_wave_defer_kernel<10, 2, true>", "srbdqp_setup1_kernel<10, 2, true, false, false, true>",
              "srbdqp_setup1_kernel<10, 2, true, false, false, false>"):
        assert by[k]["scratch"] == 0 and by[k]["vgprs"] <= 256 and by[k]["occupancy"] >= 2, (k, by[k])
    for k in ("srbdqp_wrench_kernel<10, double, double, 0, 3, double, 5, 0>", "srbdqp_wrench_kernel<12, double, double, 0, 3, double, 5, 0>",
              "srbdqp_wrench_kernel<20, float, float, 0, 3, float, 5, 0>"):
        assert by[k]["scratch"] == 0 and by[k]["occupancy"] >= 3, (k, by[k])
    k = "srbdqp_wrench_kernel_in<10, 2>"                      # the reference's own batch-1 call: one workgroup's worth of registers
    assert by[k]["scratch"] == 0
