#!/bin/bash
# AddressSanitizer pass over the GPU suite — for a machine that allows xnack+ code objects and HSA_XNACK=1.
# The GPU pool of this project refuses both (`gpurun` rejects the command), so there this script only documents the recipe; the
# memory-safety evidence that runs everywhere is tests/test_gpu_canary.py.
#   bash tools/asan_run.sh [pytest args]        (default: the parity + operating-point + canary files)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
make -C $R/talkshow_amd/csrc asan -j8
RT=$(/opt/rocm/bin/hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libasan -print-file-name=libclang_rt.asan-x86_64.so)
export HSA_XNACK=1 TS_LIB_PATH=$R/talkshow_amd/lib_asan/libtalkshow_hip_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0
cd $R
python -m pytest ${@:-tests/test_gpu_parity.py tests/test_gpu_operating_points.py tests/test_gpu_canary.py} -m gpu -x -q
