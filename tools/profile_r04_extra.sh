#!/bin/bash
# Round-4 extra profiles: PMC passes of the face generator (fused attention kernel) and kernel stats + PMC of the audio front-end (FFT kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_profiles_extra
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fe -- python $R/tools/frontend_pass.py > $O/stats_fe.log 2>&1; tail -1 $O/stats_fe.log
cp "$(find $O/stats_fe -name '*kernel_stats.csv' | head -1)" $O/stats_frontend.csv; rm -rf $O/stats_fe
pmc() { tag=$1; script=$2; shift; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$tag -- python $R/tools/$script --passes 1 > $O/$tag.log 2>&1; tail -1 $O/$tag.log; }
for t in face:face_pass.py fe:frontend_pass.py; do
  n=${t%%:*}; sc=${t##*:}
  pmc pmc_${n}_FETCH $sc FETCH_SIZE
  pmc pmc_${n}_WRITE $sc WRITE_SIZE
  pmc pmc_${n}_SQ $sc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
done
python - $O <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
O = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(O, "pmc_*"))):
    if not os.path.isdir(d): continue
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("ts::", "")
            a = acc[n][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
    out[os.path.basename(d)] = {k: {c: {"launches": v[0], "mean_per_launch": v[1] / v[0]} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open(os.path.join(O, "pmc_extra_raw.json"), "w"), indent=1)
def g(p, k, c): return out.get(p, {}).get(k, {}).get(c, {}).get("mean_per_launch")
for tag, kern in (("face", "attention_kernel"), ("fe", "stft_power_kernel")):
    f, w = g(f"pmc_{tag}_FETCH", kern, "FETCH_SIZE"), g(f"pmc_{tag}_WRITE", kern, "WRITE_SIZE")
    sq = out.get(f"pmc_{tag}_SQ", {}).get(kern, {})
    wc = sq.get("SQ_WAVE_CYCLES", {}).get("mean_per_launch", 0) or 1
    print(kern, "HBM-side bytes per launch (2 x FETCH + WRITE):", None if f is None else (2 * f + w) * 1024,
          {c: round(v["mean_per_launch"] / wc, 3) for c, v in sq.items() if c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
          "MFMA busy cycles", sq.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean_per_launch"),
          "LDS bank-conflict share", (sq.get("SQ_LDS_BANK_CONFLICT", {}).get("mean_per_launch", 0) / max(sq.get("SQ_LDS_IDX_ACTIVE", {}).get("mean_per_launch", 1), 1)))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
