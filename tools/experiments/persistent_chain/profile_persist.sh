#!/bin/bash
# rocprofv3 evidence for the persistent chain kernel (opt-in): kernel stats of two 256-clip passes, then the SQ counter set
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/persist_prof
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
export TS_CHAIN_PERSIST=${TS_CHAIN_PERSIST:-3}
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/tools/chain_pass.py --batch 256 --passes 2 > $O/stats.log 2>&1
cp "$(find $O/st -name '*kernel_stats.csv' | head -1)" $O/persist_kernel_stats.csv; rm -rf $O/st
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -- python $R/tools/chain_pass.py --batch 256 --passes 2 > $O/pmc.log 2>&1
python - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
O = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(O, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "chain_persist" in row["Kernel_Name"]:
            a = acc["chain_persist_kernel"][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
with open(os.path.join(O, "persist_pmc.txt"), "w") as out:
    for k, cs in acc.items():
        wc = cs["SQ_WAVE_CYCLES"][1] / max(cs["SQ_WAVE_CYCLES"][0], 1)
        for c, (n, v) in sorted(cs.items()):
            out.write(f"{k} {c}: {v / n:.0f} per launch ({n} launches)" + (f"  = {v / n / wc:.3f} of the wave-cycles" if c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") else "") + "\n")
print(open(os.path.join(O, "persist_pmc.txt")).read())
PY
rm -rf $O/pmc; head -5 $O/persist_kernel_stats.csv | cut -c1-160; tail -1 $O/stats.log
