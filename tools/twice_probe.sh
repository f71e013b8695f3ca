#!/bin/bash
# experiment: every wide launch issued twice back to back; per-launch durations of the 1st (cold) vs 2nd (operands in L2) copies
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/twice
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
for x in 0 1; do
TS_SKINNY_WIDE_TWICE=1 TS_SKINNY_WIDE_XCD=$x timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/x$x -- python $R/tools/chain_pass.py --batch 256 --passes 1 > $O/run$x.log 2>&1
python - $O/x$x $x <<'PY'
import csv, glob, sys, numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wide" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]) / 1e3
g = np.array([int(r["Workgroup_Size_X"]) * 0 + int(r["Grid_Size_X"]) // 512 for r in rows])
n = len(d) // 2 * 2
a, b = d[-n::2], d[-n + 1::2]
print(f"xcd={sys.argv[2]}: {len(d)} wide launches; first copies median {np.median(a):.2f} us, second copies median {np.median(b):.2f} us")
for k in np.unique(g[-n::2]):
    m = g[-n::2] == k
    print(f"   {k} workgroups: {np.median(a[m]):.2f} -> {np.median(b[m]):.2f} us  ({m.sum()} pairs)")
PY
done
rm -rf $O/x0 $O/x1
