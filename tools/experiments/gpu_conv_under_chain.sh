#!/bin/bash
# round 5, session 13: can the conv stacks of one pass run UNDER the chain of another?  conv_gemm capped at one workgroup per CU
# (TS_CONV_PAD_LDS), the chain on its split-K kernels only (TS_SKINNY_WIDE_MIN=0: <= 118 VGPRs, <= 64 KB of LDS: they fit beside a conv
# workgroup), with and without s_setprio 3 in the chain kernels (tools/lib_prio3.so).  tools/corun_probe.py: one stream of chains, one
# of conv stacks, alone / together; then the bench line for the combinations.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s13}; mkdir -p $O
cd $R
cp talkshow_amd/lib/libtalkshow_hip.so /tmp/lib_keep.so
run() {  # lib, env...
  lib=$1; shift
  cp tools/lib_$lib.so talkshow_amd/lib/libtalkshow_hip.so
  echo "== lib=$lib $*" | tee -a $O/corun.txt
  env "$@" TS_N=4 timeout 200 python tools/corun_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/corun.txt
}
run base TS_X=0
run base TS_SKINNY_WIDE_MIN=0
run base TS_SKINNY_WIDE_MIN=0 TS_CONV_PAD_LDS=20480
run prio3 TS_SKINNY_WIDE_MIN=0 TS_CONV_PAD_LDS=20480
run prio3 TS_SKINNY_WIDE_MIN=0
run prio3 TS_X=0
bench() {
  lib=$1; shift
  cp tools/lib_$lib.so talkshow_amd/lib/libtalkshow_hip.so
  line=$(env "$@" TS_BENCH_WATCHDOG=150 timeout 200 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-face --no-modes 2>/dev/null | tail -1)
  python - "$lib $*" "$line" <<'PY' | tee -a $O/bench.txt
import json, sys
d = json.loads(sys.argv[2])
print(f'{sys.argv[1]:60s} value {d["value"]/1e6:.3f} M  ms/step {d["ms_per_step"]:.3f} chain256 {d["roofline"]["chain_ms_per_pass"]:.2f} ms selfcheck {d.get("selfcheck")}')
PY
}
bench base TS_X=0
bench base TS_SKINNY_WIDE_MIN=0 TS_CONV_PAD_LDS=20480
bench prio3 TS_SKINNY_WIDE_MIN=0 TS_CONV_PAD_LDS=20480
bench prio3 TS_X=0
bench base TS_X=0
cp /tmp/lib_keep.so talkshow_amd/lib/libtalkshow_hip.so
