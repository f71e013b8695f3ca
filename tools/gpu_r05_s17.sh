#!/bin/bash
# round 5, session 17: LDS-staged codebook search: parity (op test incl. the kernel-order recompute, goldens, both forms), time
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s17}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_operating_points.py -m gpu -q -x -k "op_vq_argmin or vqvae_golden or wrapper_body_vq or vq_encode or golden_counts or per_thread_vq" 2>&1 | tail -4 | tee $O/tests.log
for v in 1 0 1 0; do
TS_VQ_LDS=$v timeout 120 python - <<'PY' | tee -a $O/vq_time.txt
import os, sys, time, ctypes as C
sys.path.insert(0, '.')
import torch, numpy as np
from talkshow_amd import _lib, synth
from talkshow_amd.modules import VQVAE
lib = _lib.load(); ctx = _lib.context(0)
vb = VQVAE(39, 64, 2048, 1024, 2).cuda(); vb.load_state_dict(synth.to_torch(synth.vqvae_state_dict(seed=7, in_dim=39)))
vh = VQVAE(90, 64, 2048, 1024, 2).cuda(); vh.load_state_dict(synth.to_torch(synth.vqvae_state_dict(seed=7, in_dim=90, salt=1)))
for n in (256, 32):
    poses = torch.from_numpy(synth.gt_poses(11, n, 300)).cuda()
    codes = torch.empty((n, 75, 2), dtype=torch.int64, device="cuda")
    run = lambda: _lib.check(lib.ts_body_vq_infer(vb.handle(), vh.handle(), _lib.dptr(poses), n, 300, _lib.dptr(codes), None, _lib.stream_ptr()))
    run(); torch.cuda.synchronize()
    _lib.check(lib.ts_prof_enable(ctx, 1)); run(); torch.cuda.synchronize()
    ms, k, fl = (C.c_double * 4)(), (C.c_int64 * 4)(), (C.c_double * 4)()
    _lib.check(lib.ts_prof_read_n(ctx, 4, ms, k, fl, 1)); _lib.check(lib.ts_prof_enable(ctx, 0))
    print(f"TS_VQ_LDS={os.environ['TS_VQ_LDS']} n={n}: misc family (code search + gathers) {ms[2]*1e3:.1f} us in {k[2]} launches; conv {ms[0]:.3f} ms; codes checksum {int(codes.sum())}")
PY
done
