#!/bin/bash
# PMC passes over the face generator with the opt-in bf16x3 plan (and fp32 beside it): what bounds conv_gemm_split?
# Issue mix, waits, L2 traffic and the latency of an L1 miss, one counter set per run (no --stats with --pmc).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_split
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
run() { tag=$1; arith=$2; shift; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$tag -- python $R/tools/face_pass.py --passes 1 --arith $arith > $O/$tag.log 2>&1; }
for a in 3 0; do
run s1_$a $a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA
run s3_$a $a SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD
run s4_$a $a TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
run s5_$a $a TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
run s6_$a $a FETCH_SIZE GRBM_GUI_ACTIVE
done
python - $O <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
O = sys.argv[1]
for d in sorted(glob.glob(os.path.join(O, "s*"))):
    if not os.path.isdir(d): continue
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("ts::", "")
            if not ("conv_gemm_split_kernel<128" in n or "conv_gemm_banded" in n): continue
            acc[n][row["Counter_Name"]] += float(row["Counter_Value"])
    for n, c in acc.items():
        print(os.path.basename(d), n, {k: float("%.6g" % v) for k, v in c.items()})
PY
find $O -name "*.csv" -delete
