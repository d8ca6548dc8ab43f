"""What csrc/srbdqp_aql.hpp assumes about the library it lives in, checked on the built library without a GPU: the gfx950 code object can be found the way the
launcher finds it (ELF section .hip_fatbin -> clang offload bundle -> the gfx950 entry), and the six *_kernel_in instantiations it dispatches by hand have exactly
their two explicit arguments (no hidden arguments to fill in), no private segment (no scratch set-up) and no static LDS beside the dynamic allocation."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _gfx950_code_object(path):
    d = open(path, "rb").read()
    assert d[:4] == b"\x7fELF"
    e_shoff, = struct.unpack_from("<Q", d, 0x28)
    e_shentsize, e_shnum, e_shstrndx = struct.unpack_from("<HHH", d, 0x3A)
    sh = [struct.unpack_from("<IIQQQQIIQQ", d, e_shoff + i * e_shentsize) for i in range(e_shnum)]
    names_off = sh[e_shstrndx][4]
    for s in sh:
        name = d[names_off + s[0]:d.index(b"\0", names_off + s[0])].decode()
        if name != ".hip_fatbin":
            continue
        b = d[s[4]:s[4] + s[5]]
        assert b[:24] == b"__CLANG_OFFLOAD_BUNDLE__", "the fat binary must stay an uncompressed offload bundle (the launcher does not decompress)"
        n, = struct.unpack_from("<Q", b, 24)
        p = 32
        for _ in range(n):
            off, sz, tl = struct.unpack_from("<QQQ", b, p)
            triple = b[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and sz:
                return b[off:off + sz]
    raise AssertionError("no gfx950 code object in " + path)


@pytest.fixture(scope="module")
def kernel_notes(tmp_path_factory):
    import __graft_entry__ as g
    g.build()
    if not os.path.exists(READELF):
        pytest.skip("no llvm-readelf in this image")
    co = tmp_path_factory.mktemp("aql") / "srbdqp_gfx950.co"
    co.write_bytes(_gfx950_code_object(os.path.join(ROOT, "g1_locomotion_amd", "libsrbdqp.so")))
    return subprocess.run([READELF, "--notes", str(co)], capture_output=True, text=True, check=True).stdout


def _block(notes, symbol):
    i = notes.index(".symbol:         " + symbol)
    j = notes.rindex("- .agpr_count", 0, i)
    k = notes.index(".wavefront_size", i)
    return notes[j:k]


# sizeof(KArgs) = 568; sizeof(StagedIn<N>) = 8 (13 + 13 N + 12 N + 3 N) + 4 N   (csrc/srbdqp_common.hpp)
@pytest.mark.parametrize("N,x,family", [(10, 2, "wrench"), (8, 2, "wrench"), (4, 1, "wrench"), (10, 2, "compact"), (8, 2, "compact"), (4, 2, "compact")])
def test_batch1_kernels_fit_the_aql_launcher(kernel_notes, N, x, family):
    sym = ("_ZN6srbdqp23srbdqp_wrench_kernel_inILi%dELi%dEEEvNS_5KArgsENS_8StagedInIXT_EEE.kd" if family == "wrench"
           else "_ZN6srbdqp24srbdqp_compact_kernel_inILi%dELi%dEEEvNS_5KArgsENS_8StagedInIXT_EEE.kd") % (N, x)     # the names srbdqp.hip formats
    blk = _block(kernel_notes, sym)
    staged_in = 8 * (13 + 28 * N) + 4 * N
    staged_in = (staged_in + 7) & ~7
    assert "hidden_" not in blk, "a kernel with hidden arguments cannot be dispatched by the hand-written packet"
    assert int(re.search(r"\.kernarg_segment_size: (\d+)", blk).group(1)) == 568 + staged_in
    assert int(re.search(r"\.private_segment_fixed_size: (\d+)", blk).group(1)) == 0
    assert int(re.search(r"\.group_segment_fixed_size: (\d+)", blk).group(1)) == 0
    assert re.search(r"\.uses_dynamic_stack: (\w+)", blk).group(1) == "false"
    assert [int(v) for v in re.findall(r"\.size:\s+(\d+)", blk)] == [568, staged_in]
