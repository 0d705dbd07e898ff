"""Readers of the committed profiles (profiles/): what bench.py adds to its line from the rocprofv3 counter passes of the same command --
the vector-issue roofline of the dominant kernel and of the whole step (SQ_* pass) and the dominant kernel's memory-side traffic
(FETCH_SIZE / WRITE_SIZE passes)."""
from __future__ import annotations

import json
import os

from .common import ROOT

KERN_OF = {"fast": "k_fast", "blur": "k_blur_mfma", "quadtree": "k_quadtree", "stereo": "k_stereo", "resize": "k_resize_regions"}
VALU_PEAK = 256 * 4 * 2.4e9 / 4          # wave-instructions per second: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction
VALU_PEAK_MEASURED = 555e9               # ... at the clock the chip holds under such a load: 535 - 575 G measured chip-wide (profiles/r5_valu_peak.txt)


def add_valu_roofline(line, dom, dom_ms, step_s, B):
    # The same kernel against its VALU ISSUE ceiling: FAST is integer work on bytes and sits far below the HBM roofline because it is
    # instruction-bound, so the HBM fraction alone says little about it.  wave-instructions per second = waves per launch x VALU
    # instructions per wave (committed rocprofv3 SQ_* counter pass, profiles/) / the launch duration measured live above; peak = 256 CUs
    # x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction.
    import re as _re
    def _round_key(f):  # r2_v10 after r2_v9
        return [int(x) for x in _re.findall(r"\d+", f)]
    sq_files = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_sq_counters.json")), key=_round_key)
    kern_of = KERN_OF
    if sq_files and dom in kern_of:
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", sq_files[-1])))
            ent = sq["kernels"].get(kern_of[dom])
            if ent and sq.get("pairs_per_step") == B and dom_ms > 0:
                peak = VALU_PEAK
                ach = ent["waves_per_step"] * ent["valu_per_wave"] / (dom_ms * 1e-3)
                line["roofline_valu"] = {"kernel": dom, "bound": "valu", "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instr/s",
                                         "frac": ach / peak, "valu_per_wave": ent["valu_per_wave"], "waves_per_launch": ent["waves_per_step"],
                                         "source": "profiles/" + sq_files[-1]}
                # the whole step against the same ceiling: sum over the kernels of waves x vector instructions per wave / the step time
                tot_valu = sum(k["waves_per_step"] * k["valu_per_wave"] for k in sq["kernels"].values())
                line["roofline_valu"]["pipeline_frac"] = tot_valu / step_s / peak
                line["roofline_valu"]["pipeline_frac_at_measured_clock"] = tot_valu / step_s / VALU_PEAK_MEASURED
                line["roofline_valu"]["pipeline_valu_wave_instr"] = tot_valu
                line["roofline_valu"]["pipeline_valu_issue_ms"] = tot_valu / peak * 1e3
                line["roofline_valu"]["pipeline_what"] = ("vector instructions of ALL kernels of a step (SQ_INSTS_VALU x waves, committed counter pass) / "
                                                          "step time / 614.4 G wave-instructions per second: the share of the step that is vector issue")
                # r5: the issue rate depends on the opcode -- profiles/r5_valu_peak.txt, r5_valu_census.txt: most of what
                # these kernels execute (32-bit min / max / min3, mads, dots, compares, cndmask, perms) issues once per 4 cycles per SIMD
                # (535 - 575 G wave-instr/s measured chip-wide = 4 cycles at the ~2.2 GHz the chip holds under such a load; 614.4 is 4 cycles at the
                # nominal 2.4 GHz), while 32-bit add / sub / logic / right shifts, fp32 add / mul / fma and the non-packed 16-bit arithmetic
                # reach ~1.8 x that with 8 waves per SIMD.  `frac` above prices every instruction at 4 cycles; frac_class_weighted prices the
                # kernel's STATIC opcode mix (tools/isa_class_mix.py), fast ones at 4 / 1.8 cycles -- the lower, more honest figure
                mix_files = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_isa_class_mix.json")), key=_round_key)
                mix_path = os.path.join(ROOT, "profiles", mix_files[-1]) if mix_files else ""
                if mix_path:
                    mix = json.load(open(mix_path))["kernels"].get(kern_of[dom])
                    if mix:
                        ff = mix["frac_fast"]
                        peak_w = peak / ((1.0 - ff) + ff / 1.8)
                        line["roofline_valu"]["issue_classes"] = {
                            "source": "profiles/r5_valu_peak.txt, profiles/r5_valu_census.txt, profiles/" + mix_files[-1],
                            "slow_class_cycles": 4.0, "fast_class_speedup_at_8_waves_per_simd": 1.8, "static_frac_fast": ff,
                            "peak_class_weighted": peak_w / 1e9, "frac_class_weighted": ach / peak_w}
        except Exception:
            pass



def add_traffic(line, dom, stage_bytes, B, images_per_launch):
    # HBM traffic of the dominant kernel from a committed rocprofv3 PMC pass of this same command (profiles/), if present
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get("kernels", {}).get(dom)
            if ent and tj.get("pairs_per_step") == B and tj.get("images_per_launch") == images_per_launch:
                line["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = tj.get("source", "profiles/pmc_traffic.json")
                pk = tj.get("per_kernel", {}).get(KERN_OF.get(dom, ""), {})
                line["roofline"]["traffic_factor"] = pk.get("read_factor", 2.0)   # raw FETCH_SIZE -> bytes, for this kernel's load shape
                line["roofline"]["traffic_factor_source"] = tj.get("read_factor_source", "MI355X_MICROARCH.md (x2)")
                line["roofline"]["traffic_what"] = tj.get("what", "memory-side request bytes (Infinity-Cache hits included): an upper bound of the HBM bytes")
                line["roofline"]["traffic_over_algorithmic"] = ent["hbm_bytes_per_launch"] / stage_bytes[dom]
        except Exception:
            pass

