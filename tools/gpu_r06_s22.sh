#!/bin/bash
# r06 session 22: the face pass per layer at the final code (in situ: HIP events around every launch), twice
mkdir -p gpurun_out/r06_s22
for i in 1 2; do timeout 300 python tools/face_layers.py 2>gpurun_out/r06_s22/layers_$i.err | tail -1; done
python - <<'PY'
import re, collections
for i in (1, 2):
    d = collections.OrderedDict()
    for l in open(f"gpurun_out/r06_s22/layers_{i}.err"):
        m = re.search(r"conv M=(\d+) N=(\d+) K=(\d+) groups=(\d+) z=(\d+) stride=(\d+).*?\s([\d.]+) us\s+([\d.]+) TF", l)
        if m:
            d.setdefault(tuple(int(m[j]) for j in range(1, 7)), []).append(float(m[7]))
    tg_t = tg_f = 0.0
    for k, v in d.items():
        fl = 2.0 * k[0] * k[1] * k[2] * k[3]
        t = sum(v) / len(v)
        tag = ""
        if k[0] == 19200 and (k[1], k[2]) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
            tg_t += sum(v); tg_f += fl * len(v); tag = "  <- transformer GEMM"
        print(i, k, f"x{len(v)} {t:9.1f} us {fl / t / 1e6:6.1f} TF{tag}")
    print(i, f"transformer GEMMs: {tg_t / 1e3:.2f} ms = {tg_f / tg_t / 1e6:.1f} TFLOP/s")
PY
