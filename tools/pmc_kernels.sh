#!/bin/bash
# rocprofv3 PMC passes over the chain at one operating point, summarised PER KERNEL NAME (mean counter value per launch).
# usage: pmc_kernels.sh <batch> <tag> [ENV=VAL ...]      (one counter set per pass; no --stats together with --pmc)
B=${1:-256}; TAG=${2:-M$B}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
run() { name=$1; shift; env "${EXTRA[@]}" timeout 240 rocprofv3 --kernel-trace "$@" --output-format csv -d $O/$name -- python $R/tools/chain_pass.py --batch $B --passes 2 > $O/$name.log 2>&1; tail -1 $O/$name.log; }
EXTRA=("$@"); [ ${#EXTRA[@]} -eq 0 ] && EXTRA=(TS_NOOP=1)
run sq   --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run tcc  --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
python - $O <<'PY' > $R/gpurun_out/pmc_$TAG.json
import csv, glob, json, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("ts::", "")
        a = acc[n][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
out = {k: dict(launches=max(v[0] for v in cs.values()), **{c: v[1] / v[0] for c, v in cs.items()}) for k, cs in acc.items() if "skinny" in k or "conv" in k}
json.dump(out, sys.stdout, indent=1)
PY
find $O -name "*.csv" -delete
cat $R/gpurun_out/pmc_$TAG.json
