"""Kernel resource table from the build log of tools/build.sh / __graft_entry__.build() (-Rpass-analysis=kernel-resource-usage):
one line per kernel: VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD), static LDS.  tests/test_build_resources.py reads the same log.

    python tools/resource_table.py [/tmp/srbdqp_build.log] [--spills]
"""
import re
import subprocess
import sys


def parse(path):
    log = open(path).read()
    rows = []
    for blk in re.split(r"(?=remark: [^\n]*Function Name:)", log):
        m = re.search(r"Function Name: (\S+)", blk)
        if not m:
            continue

        def g(k):
            mm = re.search(k + r": (\d+)", blk)
            return int(mm.group(1)) if mm else -1
        rows.append(dict(mangled=m.group(1), vgprs=g("VGPRs"), agprs=g("AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                         occupancy=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]"), sgprs=g("SGPRs")))
    names = subprocess.run(["c++filt"], input="\n".join(r["mangled"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, d in zip(rows, names):
        d = re.sub(r"^void srbdqp::", "", d)
        r["name"] = re.sub(r"\((srbdqp::)?KArgs.*$", "", d)
    return rows


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = parse(args[0] if args else "/tmp/srbdqp_build.log")
    if "--spills" in sys.argv:
        rows = [r for r in rows if r["scratch"] > 0]
    for r in rows:
        print(f"{r['name'][:120]:120s} V{r['vgprs']:4d} A{r['agprs']:3d} scratch{r['scratch']:5d} occ{r['occupancy']:2d} lds{r['lds']:7d}")
    print(f"{len(rows)} kernels, {sum(r['scratch'] > 0 for r in rows)} with scratch")
