#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel, priced with the issue costs measured on MI355X (tools/issue_rate_bench*.hip,
profiles/r03_issue_rate.md): what one wave-instruction of this kernel costs a SIMD on average, hence the instruction rate at
which the kernel's OWN mix saturates the vector ALU.  Static counts over the whole kernel body stand in for dynamic ones.

  python tools/isa_mix.py <lib.so> <kernel-name-regex> [...]  [--json out.json]

Measured classes (cycles of SIMD issue per wave64 instruction, at the ~2.0-2.1 GHz the chip holds under VALU load):
  2   VOP1 / VOP2 encodings with at most two VGPR sources: v_add / v_sub / v_mul / v_min / v_max / v_and / v_or / v_xor / v_mov /
      v_cvt / v_fmac / v_fmaak / v_fmamk  (900 - 1000 G wave-instr/s chip-wide)
  4   everything VOP3-encoded with three sources or an SGPR / carry / compare result: v_fma (three VGPRs: 2.4 - 3.7 depending on
      register banks), v_max3 / v_min3 / v_add3 / v_lshl_add, v_cmp_* (VOPC), v_cndmask, v_mul_lo / v_mul_hi / v_mad_u64_u32,
      v_div_scale / v_div_fmas / v_div_fixup, shifts, v_pk_* (two floats per lane), v_readlane / v_writelane  (540 - 580 G/s)
  8   transcendental unit: v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos  (300 G/s)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
FOUR = ("v_cmp", "v_cndmask", "v_mul_lo", "v_mul_hi", "v_mad_", "v_div_", "v_lshl", "v_lshr", "v_ashr", "v_pk_", "v_readlane", "v_writelane",
        "v_readfirstlane", "v_max3", "v_min3", "v_med3", "v_add3", "v_fma_", "v_bfe", "v_bfi", "v_perm", "v_alignbit", "v_add_co", "v_addc",
        "v_sub_co", "v_subb", "v_mbcnt", "v_ldexp", "v_frexp", "v_cvt_pk", "v_and_or", "v_or3", "v_xad", "v_lerp", "v_sad", "v_mov_b64", "v_add_lshl", "v_lshl_or")


def cost(op):
    if op.startswith(TRANS):
        return 8
    if op.startswith(FOUR) or op.endswith("_e64") or op.endswith("_dpp") or op.endswith("_sdwa"):
        return 4
    return 2


def disassemble(lib):
    with tempfile.TemporaryDirectory() as tmp:
        tmp_lib = os.path.join(tmp, os.path.basename(lib))
        os.symlink(os.path.abspath(lib), tmp_lib)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp_lib], check=True, capture_output=True)
        text = ""
        for f in sorted(os.listdir(tmp)):
            if "gfx950" in f:
                text += subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True,
                                       capture_output=True, text=True).stdout
        return text


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], check=True, capture_output=True, text=True).stdout.strip()
    except Exception:
        return name


def main():
    args = [a for a in sys.argv[1:] if a != "--json"]
    out = None
    if "--json" in sys.argv:
        out = sys.argv[sys.argv.index("--json") + 1]
        args.remove(out)
    lib, pats = args[0], args[1:]
    text = disassemble(lib)
    res = {}
    for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", text, re.S | re.M):
        name = demangle(m.group(1))
        if not any(re.search(p, name) for p in pats):
            continue
        ops = re.findall(r"^\s+([vsd][a-z0-9_]+)", m.group(2), re.M)
        valu = [o for o in ops if o.startswith("v_")]
        n = {2: 0, 4: 0, 8: 0}
        for o in valu:
            n[cost(o)] += 1
        tot = max(len(valu), 1)
        avg = (2 * n[2] + 4 * n[4] + 8 * n[8]) / tot
        # the rate at which this mix saturates the vector ALU, from the MEASURED chip-wide rates of the three classes (G wave-instr/s:
        # 950 / 560 / 300, profiles/r03_issue_rate.md) - no clock assumption: 1 / rate = sum over classes of share / class rate
        sat = tot / (n[2] / 950.0 + n[4] / 560.0 + n[8] / 300.0)
        res[name] = {"valu_instructions_static": len(valu), "class_2_cycles": n[2], "class_4_cycles": n[4], "class_8_cycles": n[8],
                     "avg_issue_cycles_per_wave_instr": round(avg, 3), "saturation_rate_G_wave_instr_per_s": round(sat, 1),
                     "salu_static": sum(o.startswith("s_") for o in ops), "lds_static": sum(o.startswith("ds_") for o in ops)}
        print("%-90s valu %5d  (2 cyc %5d, 4 cyc %5d, 8 cyc %4d)  avg %.2f cycles / wave-instr, saturates at %.0f G wave-instr/s" % (name[:90], len(valu), n[2], n[4], n[8], avg, sat))
    if out:
        json.dump({"lib": os.path.basename(lib), "kernels": res}, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
