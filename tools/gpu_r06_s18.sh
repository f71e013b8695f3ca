#!/bin/bash
# r06 session 18: per-layer times of the conv launches of one 256-clip body pass (VQ encode + audio encoder + audio terms + VQ decode)
mkdir -p gpurun_out/r06_s18
TS_B=256 timeout 300 python tools/conv_layers.py 2>gpurun_out/r06_s18/layers.err | tail -1
python - <<'PY'
import re, collections
d=collections.OrderedDict()
for l in open("gpurun_out/r06_s18/layers.err"):
    m=re.search(r"conv M=(\d+) N=(\d+) K=(\d+) groups=(\d+) z=(\d+) stride=(\d+) Lout=(\d+)\s+([\d.]+) us\s+([\d.]+) TF", l)
    if m:
        k=tuple(int(m[i]) for i in range(1,8)); d.setdefault(k,[]).append(float(m[8]))
tot=sum(sum(v) for v in d.values())
print("total us", tot/2, "(per step)")
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    fl=2.0*k[0]*k[1]*k[2]*k[3]
    t=sum(v)/len(v)
    print(k, "x%d"%(len(v)//2), f"{t:8.1f} us  {fl/t/1e6:6.1f} TF  share {sum(v)/tot*100:4.1f} %")
PY
