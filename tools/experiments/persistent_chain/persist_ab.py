#!/usr/bin/env python3
"""Persistent chain kernel (skinny_persist.hip) against the hipGraph of launches: the codes of a pass must be bit-identical in every
sampling mode, and the pass time.  Each path runs in its own process (the switch is read once).

    python tools/persist_ab.py [clips ...]
"""
import os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from talkshow_amd import _lib, synth
B, out = int(sys.argv[1]), sys.argv[2]
w, _ = bench.build_models(0)
T = 300
mf = torch.from_numpy(synth.mfcc_features(1, B, T)).cuda(); ids = torch.from_numpy(synth.speaker_ids(B)).cuda()
feat = w.audioencoder.forward_nlc(mf)
res = {}
for name, kw in (("greedy", dict(mode=_lib.TS_SAMPLE_GREEDY)), ("philox", dict(mode=_lib.TS_SAMPLE_PHILOX, seed=1234)),
                 ("uniforms", dict(mode=_lib.TS_SAMPLE_UNIFORMS, uniforms=torch.rand((B, 75, 2), generator=torch.Generator().manual_seed(5)).cuda()))):
    r = w.generator.run(ids, feat, **kw)
    torch.cuda.synchronize()
    res[name] = (r[0] if isinstance(r, tuple) else r).cpu().numpy()
w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8): w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)
torch.cuda.synchronize()
res["ms"] = np.array((time.perf_counter() - t0) / 8 * 1e3)
np.savez(out, **res)
''' % REPO
import numpy as np
for B in [int(x) for x in sys.argv[1:]] or [256, 128]:
    outs = {}
    for p in os.environ.get("PERSIST_MODES", "0 1").split():
        f = tempfile.mktemp(suffix=".npz")
        r = subprocess.run([sys.executable, "-c", CHILD, str(B), f], env=dict(os.environ, TS_CHAIN_PERSIST=p, TS_CHAIN_PERSIST_DEBUG="1"),
                           capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            print(f"clips {B} TS_CHAIN_PERSIST={p}: FAILED\n" + r.stdout[-1500:] + r.stderr[-2500:]); continue
        msg = [l for l in r.stderr.splitlines() if l.startswith("[ts]")]
        outs[p] = np.load(f)
        print(f"clips {B} TS_CHAIN_PERSIST={p}: {float(outs[p]['ms']):.2f} ms per pass " + " ".join(msg[:2]))
    base = outs.get("0")
    for p, o in outs.items():
        if p == "0" or base is None: continue
        for k in ("greedy", "philox", "uniforms"):
            d = int((o[k] != base[k]).sum())
            print(f"   {k:9s}: {d} of {o[k].size} codes differ from the launch graph" + ("" if d else "  (bit-identical)"))
