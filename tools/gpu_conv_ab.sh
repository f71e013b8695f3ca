#!/bin/bash
# conv_gemm A/B: parity of the conv operators / modules with the current build, then same-box A/B of library builds on the conv stacks
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-conv_ab}; mkdir -p $O; shift
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "op_conv1d or conv_tile or conv_banded or vqvae_golden or audioenc_golden or face_golden or wrapper_body_vq_e2e" 2>&1 | tail -4 | tee $O/tests.log
bash tools/conv_lib_ab.sh "$@" 2>&1 | tee $O/conv_ab.txt
