import os, sys, ctypes as C
os.environ["TS_PROF_LOG"]="1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import torch, bench
from talkshow_amd import _lib, synth
lib=_lib.load(); w,_=bench.build_models(0)
B,T=int(os.environ.get("TS_B","32")),300
mfcc=torch.from_numpy(synth.mfcc_features(1000,B,T)).cuda(); gt=torch.from_numpy(synth.gt_poses(2000,B,T)).cuda()*float(os.environ.get("TS_GT_SCALE","1")); ids=torch.from_numpy(synth.speaker_ids(B)).cuda()
codes=torch.empty((B,75,2),dtype=torch.int64,device="cuda")
def step():
    _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(gt), B, T, _lib.dptr(codes), None, _lib.stream_ptr()))
    w.generate_batch(mfcc, ids, mode=_lib.TS_SAMPLE_GREEDY)
step(); torch.cuda.synchronize()
ctx=_lib.context(0)
_lib.check(lib.ts_prof_enable(ctx,1)); step(); step(); torch.cuda.synchronize()
ms,n,fl=(C.c_double*3)(),(C.c_int64*3)(),(C.c_double*3)()
_lib.check(lib.ts_prof_read(ctx,ms,n,fl,1))
print("conv total ms", ms[0], "launches", n[0])
