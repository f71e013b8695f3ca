#!/bin/bash
# per-launch kernel trace of the chain at a few operating points (GPU box): bash tools/trace_chain.sh "256 192 128"
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/trace_chain
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
for B in $1; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/M$B -- python $R/tools/chain_pass.py --batch $B --passes 1 > $O/M$B.log 2>&1
  f=$(find $O/M$B -name "*kernel_trace.csv" | head -1)
  python - "$f" "$O/M$B.trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as fo:
    fo.write("kernel,start_ns,dur_ns,grid,wg\n")
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        n = r["Kernel_Name"]
        short = "fast%s" % n.split("<")[1].split(">")[0].replace(" ", "") if "skinny16_fast" in n else ("generic" if "skinny" in n else ("sample" if "sample" in n else ("conv" if "conv_gemm" in n else "other")))
        fo.write(f'{short},{int(r["Start_Timestamp"]) - t0},{int(r["End_Timestamp"]) - int(r["Start_Timestamp"])},{r["Grid_Size_X"]},{r["Workgroup_Size_X"]}\n')
PY
  rm -rf $O/M$B
done
ls -la $O
