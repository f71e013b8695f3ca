#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats for the body bench and the face generator, then the two PMC passes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_body $O/prof_face $O/pmc_fetch $O/pmc_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_body -- python $R/bench.py --steps 8 --warmup 4 --no-face --no-cpu-baseline --no-roofline > $O/prof_body.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_face -- python -c "import sys; sys.path.insert(0,'$R'); import bench, json; print(json.dumps(bench.face_block(0)))" > $O/prof_face.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --streams 1 --no-face --no-roofline --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --streams 1 --no-face --no-roofline --no-cpu-baseline > $O/pmc_write.log 2>&1
# keep only the small csv files (gpurun merges <= 64 MiB back)
find $O/prof_body $O/prof_face -name "*kernel_trace.csv" -delete
for d in pmc_fetch pmc_write; do find $O/$d -name "*kernel_trace.csv" -delete; done
du -sh $O/prof_body $O/prof_face $O/pmc_fetch $O/pmc_write
tail -1 $O/prof_body.log | cut -c1-300
