#!/usr/bin/env python3
"""pmc_summary.py output of tools/profile_r06.sh -> the per-launch summary tracked as profiles/r06_pmc_summary.json.

    python tools/pmc_r06.py gpurun_out/r06_profiles/pmc_raw.json profiles/r06_pmc_summary.json

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: the gfx950 correction of the micro-architecture guide's
HBM / rocprofv3 section (FETCH_SIZE counts 64-byte units on this part while rocprofv3 scales it as 32-byte ones)."""
import json, sys

raw = json.load(open(sys.argv[1]))
old_note = None
try:
    old_note = json.load(open(sys.argv[2])).get("_note")
except Exception:
    pass
out = {}
for M in (256, 32):
    for fam in ("skinny_gemm_f32", "conv_gemm_f32"):
        g = lambda p, c: raw[f"pmc_M{M}_{p}"][fam][c]["mean_per_launch"]
        n = raw[f"pmc_M{M}_FETCH_SIZE"][fam]["FETCH_SIZE"]["launches"]
        f, w = g("FETCH_SIZE", "FETCH_SIZE"), g("WRITE_SIZE", "WRITE_SIZE")
        hit, miss = g("TCC", "TCC_HIT_sum"), g("TCC", "TCC_MISS_sum")
        wc = g("SQ", "SQ_WAVE_CYCLES")
        out[f"{fam}_M{M}"] = {
            "launches_profiled": n,
            "FETCH_SIZE_KiB_per_launch_raw": f,
            "WRITE_SIZE_KiB_per_launch_raw": w,
            "hbm_bytes_per_launch": (2 * f + w) * 1024,
            "L2_hit_rate": hit / (hit + miss),
            "TCP_TCC_READ_REQ_per_launch": g("TCC", "TCP_TCC_READ_REQ_sum"),
            "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": g("SQ", "SQ_VALU_MFMA_BUSY_CYCLES"),
            "SQ_WAVES_per_launch": g("SQ", "SQ_WAVES"),
            "SQ_WAVE_CYCLES_per_launch": wc,
            "wave_cycles_share": {"WAIT_ANY (memory / barrier)": g("SQ", "SQ_WAIT_ANY") / wc,
                                  "WAIT_INST_ANY (issue stall)": g("SQ", "SQ_WAIT_INST_ANY") / wc,
                                  "ACTIVE_INST_ANY": g("SQ", "SQ_ACTIVE_INST_ANY") / wc},
            "GRBM_GUI_ACTIVE_per_launch": g("SQ", "GRBM_GUI_ACTIVE"),
            "SQ_BUSY_CYCLES_per_launch": g("SQ", "SQ_BUSY_CYCLES"),
        }
# per kernel name (round 3+: the 64 x 64 wide kernel apart from the split-K kernels), M = 256
per = {}
for name in raw.get("pmc_M256_FETCH_SIZE", {}):
    if name in ("skinny_gemm_f32", "conv_gemm_f32") or "skinny" not in name:
        continue
    try:
        g = lambda p, c: raw[f"pmc_M256_{p}"][name][c]["mean_per_launch"]
        f, w = g("FETCH_SIZE", "FETCH_SIZE"), g("WRITE_SIZE", "WRITE_SIZE")
        hit, miss = g("TCC", "TCC_HIT_sum"), g("TCC", "TCC_MISS_sum")
        wc = g("SQ", "SQ_WAVE_CYCLES")
        per[name] = {"launches_profiled": raw["pmc_M256_FETCH_SIZE"][name]["FETCH_SIZE"]["launches"],
                     "hbm_bytes_per_launch": (2 * f + w) * 1024, "L2_hit_rate": hit / (hit + miss),
                     "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": g("SQ", "SQ_VALU_MFMA_BUSY_CYCLES"),
                     "wave_cycles_share": {"WAIT_ANY": g("SQ", "SQ_WAIT_ANY") / wc, "WAIT_INST_ANY": g("SQ", "SQ_WAIT_INST_ANY") / wc,
                                           "ACTIVE_INST_ANY": g("SQ", "SQ_ACTIVE_INST_ANY") / wc}}
    except KeyError:
        pass
out["per_kernel_M256"] = per
if len(sys.argv) > 3:
    for k in out:
        if isinstance(out[k], dict) and k != "per_kernel_M256":
            out[k]["commit"] = sys.argv[3]
if old_note:
    out["_note"] = old_note
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    if k not in ("_note", "per_kernel_M256"):
        print(k, f'{v["hbm_bytes_per_launch"]/1e6:.2f} MB/launch, L2 hit {v["L2_hit_rate"]:.3f}, MFMA busy cyc {v["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"]:.0f}, shares',
              {a: round(b, 3) for a, b in v["wave_cycles_share"].items()})
