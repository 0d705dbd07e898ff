#!/usr/bin/env python3
"""Static mix of vector-instruction ISSUE CLASSES per kernel (r5).  gfx950 issues some wave64 VALU opcodes at ~1.8 x
the rate of the others once several waves share a SIMD (profiles/r5_valu_peak.txt, r5_valu_census.txt: ~920 - 1050 G wave-instr/s chip-wide
against 535 - 575 G; one wave alone: the same 5 cycles either way).  FAST, as measured: 32-bit add / sub / and / or / xor / right shifts /
mov, fp32 add / sub / mul / fma, and the non-packed 16-bit VOP2 arithmetic (add, sub, min, max, mul_lo, shifts).  SLOW: everything else the
census tried -- 32-bit min / max / min3 / max3, v_lshlrev_b32, multiplies and mads, bfe / perm / alignbit / bfi, dots, sads, compares,
cndmask, conversions, every packed (v_pk_*) op, and every *_sdwa form whatever its opcode.  An opcode the census did not try counts as slow.

Compiles the kernel sources to ISA here (no GPU needed) and counts the opcodes of each kernel symbol: a STATIC mix (not weighted by
execution), written to profiles/<round>_isa_class_mix.json (argument: round tag, default r6) for bench.py's roofline_valu (the newest one is read)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_add_f32",
        "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u16", "v_sub_u16", "v_subrev_u16", "v_min_u16", "v_max_u16",
        "v_min_i16", "v_max_i16", "v_mul_lo_u16", "v_lshrrev_b16", "v_add_f16", "v_sub_f16", "v_max_f16", "v_min_f16"}
NOT_ALU = ("v_readlane", "v_readfirstlane", "v_writelane", "v_nop", "v_mfma", "v_accvgpr")


def mix_of(src, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", src, "-o", "-"] + list(extra)
    asm = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stdout
    out = {}
    for m in re.finditer(r"\n(_Z[\w]+):\s*;[^\n]*\n(.*?)\n\.Lfunc_end", asm, re.S):
        sym, body = m.group(1), m.group(2)
        name = re.sub(r"^_ZN5orbfe\d+", "", sym)
        name = re.match(r"[a-z_0-9]+", name).group(0) if re.match(r"[a-z_0-9]+", name) else sym
        ops = collections.Counter(re.sub(r"_e(32|64)$|_dpp$", "", x.group(1)) for x in re.finditer(r"^\s+(v_[a-z0-9_]+)", body, re.M))
        ops = {k: v for k, v in ops.items() if not k.startswith(NOT_ALU)}
        tot, fast = sum(ops.values()), sum(v for k, v in ops.items() if k in FAST)
        e = out.setdefault(name, {"valu_static": 0, "fast_static": 0, "instances": 0, "top": collections.Counter()})
        e["valu_static"] += tot
        e["fast_static"] += fast
        e["instances"] += 1
        e["top"].update(ops)
    for e in out.values():
        e["frac_fast"] = round(e["fast_static"] / max(e["valu_static"], 1), 4)
        e["top"] = dict(e["top"].most_common(12))
    return out


def main():
    res = {"what": __doc__.split("\n\n")[0], "fast_opcodes": sorted(FAST), "kernels": {}}
    csrc = os.path.join(ROOT, "orb_slam2_ros2_amd", "csrc")
    for f, extra in (("k_fast.hip", ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]), ("k_pyramid.hip", []), ("k_blur_mfma.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"]), ("k_quadtree.hip", []), ("k_brief.hip", []),
                     ("k_match.hip", [])):
        for k, v in mix_of(os.path.join(csrc, f), extra).items():
            if k.startswith("k_"):
                res["kernels"][k] = v
    out = os.path.join(ROOT, "profiles", (sys.argv[1] if len(sys.argv) > 1 else "r6") + "_isa_class_mix.json")
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res["kernels"].items()):
        print(f"{k:22s} VALU {v['valu_static']:6d}  fast {v['frac_fast']:.2f}  ({v['instances']} instance(s))")


if __name__ == "__main__":
    main()
