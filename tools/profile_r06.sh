#!/bin/bash
# Round-6 profile set (GPU box): rocprofv3 kernel stats of the bench command in its three execution modes, then the PMC
# passes (one counter set per run, --kernel-trace only: no --stats / sys-trace together with --pmc) over the PixelCNN chain +
# conv stacks at M = 256 and M = 32 clips per stage.  Raw output under gpurun_out/r06_profiles; summaries are copied to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_profiles
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
stats() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python $R/bench.py "$@" --no-face --no-cpu-baseline --no-modes --no-roofline > $O/$tag.log 2>&1; tail -1 $O/$tag.log | cut -c1-160; f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); cp "$f" $O/$tag.csv; rm -rf $O/$tag; }
stats stats_default --steps 20 --warmup 5
stats stats_single_stream --steps 16 --warmup 8 --coalesce 8 --streams 1
stats stats_one_batch --steps 8 --warmup 2 --coalesce 1 --streams 1
# the face generator alone (configs[2], batch 64): kernel stats with the fused attention kernel
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_face -- python $R/tools/face_pass.py > $O/stats_face.log 2>&1; tail -1 $O/stats_face.log | cut -c1-160
cp "$(find $O/stats_face -name '*kernel_stats.csv' | head -1)" $O/stats_face.csv; rm -rf $O/stats_face
pmc() { tag=$1; B=$2; shift; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$tag -- python $R/tools/chain_pass.py --batch $B --passes 2 --convs > $O/$tag.log 2>&1; tail -1 $O/$tag.log; find $O/$tag -name "*kernel_trace.csv" -delete; find $O/$tag -name "*agent_info.csv" -delete; }
for B in 256 32; do
  pmc pmc_M${B}_FETCH_SIZE $B FETCH_SIZE
  pmc pmc_M${B}_WRITE_SIZE $B WRITE_SIZE
  pmc pmc_M${B}_SQ $B SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  pmc pmc_M${B}_TCC $B TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
done
python - $O <<'PY'
# per KERNEL NAME means (the wide kernel apart from the split-K ones) + the family summary bench.py quotes
import csv, glob, json, os, re, sys
from collections import defaultdict
O = sys.argv[1]
raw = {}
for d in sorted(glob.glob(os.path.join(O, "pmc_M*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("ts::", "")
            for key in (n, "skinny_gemm_f32" if "skinny" in n else ("conv_gemm_f32" if ("conv_gemm" in n or "conv_ring" in n) else None)):
                if key:
                    a = acc[key][row["Counter_Name"]]
                    a[0] += 1; a[1] += float(row["Counter_Value"])
    raw[os.path.basename(d)] = {k: {c: {"launches": v[0], "mean_per_launch": v[1] / v[0]} for c, v in cs.items()} for k, cs in acc.items()
                                if "skinny" in k or "conv_gemm" in k or "conv_ring" in k}
json.dump(raw, open(os.path.join(O, "pmc_raw.json"), "w"), indent=1)
PY
find $O -name "*counter_collection.csv" -delete
du -sh $O; ls $O
