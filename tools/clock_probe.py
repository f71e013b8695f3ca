#!/usr/bin/env python3
"""What clock does the GPU sustain under the fp32 MFMA conv kernel?  Runs the 128x128 tile on a 4096^3 GEMM for a few
seconds while sampling rocm-smi, and prints achieved TFLOP/s next to the reported sclk (nominal peak assumes 2.4 GHz)."""
import ctypes as C, os, subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from talkshow_amd import _lib
lib = _lib.load(); ctx = _lib.context(0)
B, L, Cin, Cout, K = 32, 128, 4096, 4096, 1
x = torch.randn(B, L, Cin, device="cuda"); w = torch.randn(Cout, K * Cin, device="cuda") / 64; b = torch.randn(Cout, device="cuda")
out = torch.empty(B, L, Cout, device="cuda")
samples = []
def smi():
    for _ in range(6):
        time.sleep(0.7)
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        samples.append([l.strip() for l in r.splitlines() if "sclk" in l or "Power" in l or "mclk" in l][:3])
t = threading.Thread(target=smi); t.start()
for tile in (1, 2):
    ms = C.c_float()
    t0 = time.time()
    while time.time() - t0 < 2.5:
        _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(x), B, L, Cin, _lib.dptr(w), _lib.dptr(b), Cout, K, tile, 200, _lib.dptr(out), C.byref(ms), None))
    print(f"tile {tile}: {2.0*B*L*Cin*Cout/(ms.value*1e-3)/1e12:.1f} TFLOP/s sustained")
t.join()
for s_ in samples: print(s_)
