#!/bin/bash
# round 4, GPU session 1: full GPU test-suite, chain || chain probe, FETCH_SIZE of the persistent chain kernel, one bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_s1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > $O/tests.log
cat $O/tests.log | tail -25
timeout 200 python tools/chain_corun.py > $O/chain_corun.txt 2>&1
TS_SKINNY_WIDE_MIN=0 timeout 200 python tools/chain_corun.py >> $O/chain_corun.txt 2>&1
cat $O/chain_corun.txt | grep chain
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json | head -c 3000; tail -3 $O/bench.err
# the persistent kernel's fabric traffic (weight-traffic floor of the clip-per-XCD split): FETCH_SIZE / WRITE_SIZE in their own passes
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  TS_CHAIN_PERSIST=3 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/tools/chain_pass.py --batch 256 --passes 2 > $O/pmc_$c.log 2>&1
done
python - $O <<'PY'
import csv, glob, os, sys
O = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(O, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "chain_persist" in row["Kernel_Name"] and row["Counter_Name"] == c:
                tot += float(row["Counter_Value"]); n += 1
    print(f"chain_persist_kernel {c}: {tot / max(n, 1):.0f} (rocprofv3 units, KiB) per launch over {n} launches")
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
