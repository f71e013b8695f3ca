#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the tracked summaries under profiles/.

    python tools/profile_summary.py stats <rocprof_dir> <out.csv>          # --kernel-trace --stats run
    python tools/profile_summary.py pmc <fetch_dir> <write_dir> <out.json>  # two --pmc passes (FETCH_SIZE, WRITE_SIZE)

Kernel families: skinny_gemm_f32 = the PixelCNN chain kernels (csrc/skinny_gemm.hip), conv_gemm_f32 = the implicit-GEMM
conv kernel (csrc/conv_gemm.hip); everything else is listed under its own (shortened) name.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def family(name):
    if "skinny" in name:
        return "skinny_gemm_f32"
    if "conv_gemm_kernel" in name or "conv_gemm_banded_kernel" in name or "conv_ring_kernel" in name:
        return "conv_gemm_f32"
    m = re.match(r"(?:void )?(?:ts::)?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def find(d, pat):
    r = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not r:
        raise SystemExit(f"no {pat} under {d}")
    return r


def stats(d, out):
    fam = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    var = defaultdict(lambda: [0, 0.0])
    for f in find(d, "*kernel_stats.csv"):
        for row in csv.DictReader(open(f)):
            n, calls, tot = row["Name"], int(row["Calls"]), float(row["TotalDurationNs"])
            a = fam[family(n)]
            a[0] += calls; a[1] += tot; a[2] = min(a[2], float(row["MinNs"])); a[3] = max(a[3], float(row["MaxNs"]))
            v = var[n if len(n) < 110 else n[:110]]
            v[0] += calls; v[1] += tot
    total = sum(a[1] for a in fam.values())
    with open(out, "w") as fo:
        fo.write("# rocprofv3 --kernel-trace --stats, aggregated by kernel family (tools/profile_summary.py)\n")
        fo.write("family,calls,total_ms,avg_us,min_us,max_us,percent\n")
        for k, a in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            fo.write(f"{k},{a[0]},{a[1]/1e6:.3f},{a[1]/a[0]/1e3:.3f},{a[2]/1e3:.3f},{a[3]/1e3:.3f},{100*a[1]/total:.2f}\n")
        fo.write("# per kernel instantiation\nkernel,calls,total_ms,avg_us\n")
        for k, v in sorted(var.items(), key=lambda kv: -kv[1][1]):
            fo.write(f"\"{k}\",{v[0]},{v[1]/1e6:.3f},{v[1]/v[0]/1e3:.3f}\n")
    print(open(out).read())


def pmc_pass(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in find(d, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = acc[family(row["Kernel_Name"])]
            a[0] += 1; a[1] += float(row["Counter_Value"])
    return acc


def pmc(dfetch, dwrite, out):
    fe, wr = pmc_pass(dfetch, "FETCH_SIZE"), pmc_pass(dwrite, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        n = fe[k][0] or wr[k][0]
        f_kib = fe[k][1] / max(fe[k][0], 1)
        w_kib = wr[k][1] / max(wr[k][0], 1)
        res[k] = {"launches": n, "FETCH_SIZE_KiB_per_launch_raw": f_kib, "WRITE_SIZE_KiB_per_launch_raw": w_kib,
                  "hbm_bytes_per_launch": (2 * f_kib + w_kib) * 1024}
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (each with --kernel-trace only) over "
                    "`bench.py --steps 1 --warmup 1 --streams 1 --no-face --no-roofline --no-cpu-baseline` (MI355X, ROCm 7.2); "
                    "values in KiB as reported; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE tallies "
                    "128-B requests at 64 B for wide (16 B/lane) reads, so it is doubled per MI355X_MICROARCH.md §HBM; WRITE_SIZE is "
                    "uncalibrated. Infinity-Cache hits are counted as fetches, so this is L2-miss (fabric) traffic, an upper bound "
                    "on HBM bytes.")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k in ("skinny_gemm_f32", "conv_gemm_f32")}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
