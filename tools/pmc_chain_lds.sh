#!/bin/bash
# LDS bank conflicts / issue mix of the chain kernels by kernel name (one PMC pass over two 256-clip chain passes)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_pmc_chain_lds
cd /tmp && export TMPDIR=/tmp; rm -rf $O; mkdir -p $O
timeout 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA --output-format csv -d $O/p -- python $R/tools/chain_pass.py --batch 256 --passes 2 > $O/log.txt 2>&1
python - $O <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
O = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(O, "p", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("ts::", "")
        a = acc[n][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
with open(os.path.join(O, "summary.txt"), "w") as out:
    for k, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 0])[1]):
        m = {c: v[1] / v[0] for c, v in cs.items()}
        line = (f"{k[:48]:48s} launches {int(cs['SQ_WAVE_CYCLES'][0]):5d}  LDS conflict share {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}  "
                f"per launch: LDS insts {m.get('SQ_INSTS_LDS', 0):.0f} VALU {m.get('SQ_INSTS_VALU', 0):.0f} SALU {m.get('SQ_INSTS_SALU', 0):.0f} VMEM_RD {m.get('SQ_INSTS_VMEM_RD', 0):.0f} MFMA {m.get('SQ_INSTS_MFMA', 0):.0f}")
        print(line); out.write(line + "\n")
PY
rm -rf $O/p
