#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel in a built library, read from the gfx950 code object's
metadata notes (what the hardware allocates; rocprofv3's VGPR_Count column is something else).

  python tools/kernel_regs.py [simple-path-tracer_amd/lib/libspt_hip.so] [--md]
waves / SIMD = min(8, 512 // (ceil(vgpr / 8) * 8))   (MI355X_MICROARCH.md, register files)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table(lib):
    with tempfile.TemporaryDirectory() as tmp:
        tmp_lib = os.path.join(tmp, os.path.basename(lib))
        os.symlink(os.path.abspath(lib), tmp_lib)     # llvm-objdump writes the bundles next to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp_lib], check=True, capture_output=True)
        co = sorted(f for f in os.listdir(tmp) if "gfx950" in f)   # one bundle per translation unit of the library
        assert co, "no gfx950 code object in " + lib
        notes = "".join(subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True,
                                       capture_output=True, text=True).stdout for f in co)
    rows = []
    for k in re.split(r"\n\s+- ", notes):
        m = re.search(r"\.name:\s+(\S+)", k)
        if not m or ".vgpr_count" not in k:
            continue
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, k).group(1)) if re.search(r"\.%s:\s+(\d+)" % key, k) else 0
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))
        v = g("vgpr_count")
        alloc = (v + 7) // 8 * 8
        rows.append({"kernel": name, "vgpr": v, "agpr": g("agpr_count"), "sgpr": g("sgpr_count"), "vgpr_spill": g("vgpr_spill_count"),
                     "sgpr_spill": g("sgpr_spill_count"), "scratch": g("private_segment_fixed_size"), "lds_static": g("group_segment_fixed_size"),
                     "waves_per_simd": min(8, 512 // max(alloc, 8))})
    return sorted(rows, key=lambda r: r["kernel"])


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "simple-path-tracer_amd", "lib", "libspt_hip.so")
    rows = kernel_table(lib)
    if "--md" in sys.argv:
        print("| kernel | VGPR | waves / SIMD | SGPR | spilled SGPR | spilled VGPR | scratch B / lane |\n|---|---|---|---|---|---|---|")
        for r in rows:
            print("| `%s` | %d | %d | %d | %d | %d | %d |" % (r["kernel"], r["vgpr"], r["waves_per_simd"], r["sgpr"], r["sgpr_spill"], r["vgpr_spill"], r["scratch"]))
    else:
        for r in rows:
            print("%-64s vgpr %3d (%d waves/SIMD)  sgpr %3d  spill s%3d v%3d  scratch %5d" %
                  (r["kernel"][:64], r["vgpr"], r["waves_per_simd"], r["sgpr"], r["sgpr_spill"], r["vgpr_spill"], r["scratch"]))


if __name__ == "__main__":
    main()
