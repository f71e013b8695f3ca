"""Is the decoder's 8-10 % deficit against the encoder (same conv shapes) a property of the decoder or of what ran before it?
Sequences of single-network launches (groups = 1) with per-launch HIP-event timing (TS_PROF_LOG): dec dec enc enc dec enc."""
import os, sys, ctypes as C
os.environ["TS_PROF_LOG"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from talkshow_amd import _lib, synth
lib = _lib.load(); w, _ = bench.build_models(0)
B, T = int(os.environ.get("TS_B", "256")), 300
gt = torch.from_numpy(synth.gt_poses(2000, B, T)).cuda()
body = gt[..., :39].contiguous()
lat = torch.randint(0, 2048, (B, T // 4), dtype=torch.int64, device="cuda")
vq = w.g_body
def enc(): vq.encode_nlc(body)
def dec(): vq.decode_nlc(lat)
enc(); dec(); torch.cuda.synchronize()
ctx = _lib.context(0)
_lib.check(lib.ts_prof_enable(ctx, 1))
for f in os.environ.get("TS_SEQ", "dec dec enc enc dec enc").split():
    print("==", f, flush=True)
    {"enc": enc, "dec": dec}[f]()
torch.cuda.synchronize()
ms, n, fl = (C.c_double * 3)(), (C.c_int64 * 3)(), (C.c_double * 3)()
_lib.check(lib.ts_prof_read(ctx, ms, n, fl, 1))
