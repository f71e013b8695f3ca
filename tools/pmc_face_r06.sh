#!/bin/bash
# Fabric traffic of the face generator's conv launches (BASELINE configs[2], batch 64) without (TS_CONV_SK=0) and with (default) the ring
# engine's stream-K band: rocprofv3 --kernel-trace --pmc, one counter set per run (no --stats / sys-trace with
# --pmc), FETCH_SIZE doubled as the micro-architecture guide's HBM section prescribes for gfx950.  Summary -> gpurun_out/r06_pmc_face/summary.json
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_pmc_face
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
for deal in 0 1; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=sk${deal}_$(echo $set | cut -d' ' -f1)
    TS_CONV_SK=$deal timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python $R/tools/face_pass.py --passes 1 > $O/$tag.log 2>&1
    tail -1 $O/$tag.log
  done
done
python - $O <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
O = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(O, "sk*"))):
    if not os.path.isdir(d):
        continue
    deal = os.path.basename(d).split("_")[0]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("ts::", "")
            if not ("conv" in n or "attention" in n):
                continue
            for key in (n, "all conv launches" if "conv_" in n and "conv0" not in n else None):
                if key:
                    a = acc[key][row["Counter_Name"]]
                    a[0] += 1; a[1] += float(row["Counter_Value"])
    for k, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(deal, {}).setdefault(k, {})[c] = {"launches": v[0], "sum": v[1], "mean_per_launch": v[1] / v[0]}
summ = {}
for deal, ks in out.items():
    for k, cs in ks.items():
        if "FETCH_SIZE" not in cs:
            continue
        n = cs["FETCH_SIZE"]["launches"]
        f, w = cs["FETCH_SIZE"]["sum"], cs.get("WRITE_SIZE", {"sum": 0})["sum"]
        hit, miss = cs.get("TCC_HIT_sum", {"sum": 0})["sum"], cs.get("TCC_MISS_sum", {"sum": 1})["sum"]
        # two calls of the generator per run (one warm-up + one timed): per call = / 2
        summ.setdefault(deal, {})[k] = {"launches_profiled": n, "fabric_read_GB_per_face_call": 2 * f * 1024 / 2 / 1e9, "fabric_write_GB_per_face_call": w * 1024 / 2 / 1e9,
                                        "fabric_MB_per_launch": (2 * f + w) * 1024 / n / 1e6, "L2_hit_rate": hit / max(hit + miss, 1)}
json.dump(summ, open(os.path.join(O, "summary.json"), "w"), indent=1)
for deal, ks in summ.items():
    for k, v in sorted(ks.items()):
        print(deal, f"{k[:60]:60s}", {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
