#!/bin/bash
# round 5, session 8: ring engine (8 waves) as the production engine for the 128 x 128 layers: parity of the whole suite, then
# same-box A/B against the register-staged engine (TS_CONV_RING=0) on the conv stacks of a 256-clip pass, the face batch, the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s8}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1
tail -4 $O/tests.log
bash tools/conv_mix_ab.sh "TS_CONV_RING=0" "TS_CONV_RING=9" "TS_CONV_RING=1" 2>&1 | tee $O/conv_stacks_ab.txt
for v in 0 9 0 9; do
TS_CONV_RING=$v timeout 300 python - <<'PY' 2>&1 | tail -1 | tee -a $O/face_ab.txt
import json, os, sys
sys.path.insert(0, '.')
import bench, torch
torch.cuda.set_device(0)
f = bench.face_block(0)
print("TS_CONV_RING=" + os.environ["TS_CONV_RING"], json.dumps({k: f[k] for k in ('frames_per_s', 'ms_per_batch', 'conv_gemm_f32', 'other_kernels_ms')}))
PY
done
for v in 0 9 0 9; do
echo "TS_CONV_RING=$v" >> $O/bench_ab.txt
TS_CONV_RING=$v TS_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-face --no-modes 2>> $O/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.3f M ms/step %.3f chain %.2f conv frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['chain_ms_per_pass'], d['roofline_conv_gemm']['frac']))" | tee -a $O/bench_ab.txt
done
