#!/bin/bash
# r06 session 9: in-situ per-layer times of the face pass with the stream-K band forced wherever a plan exists (TS_CONV_SK=2) vs off (0), twice
mkdir -p gpurun_out/r06_s9
O=gpurun_out/r06_s9
for round in 1 2; do
for sk in 0 2; do
  TS_CONV_SK=$sk timeout 300 python tools/face_layers.py 2>$O/face_layers_sk${sk}_$round.err | tail -1
done; done
python - <<'PY'
import re, collections
def load(f):
    d=collections.OrderedDict()
    for l in open(f):
        m=re.search(r"conv M=(\d+) N=(\d+) K=(\d+) groups=(\d+) z=(\d+) stride=(\d+).*?\s([\d.]+) us\s+([\d.]+) TF", l)
        if m:
            d.setdefault(tuple(int(m[i]) for i in range(1,7)),[]).append(float(m[7]))
    return d
O="gpurun_out/r06_s9/"
for rnd in (1,2):
    a,b=load(O+f"face_layers_sk0_{rnd}.err"),load(O+f"face_layers_sk2_{rnd}.err")
    for k in a:
        ta,tb=sum(a[k])/len(a[k]),sum(b[k])/len(b[k])
        if abs(ta-tb)/ta>0.004: print(rnd,k,len(a[k]),f"off {ta:9.1f} us | forced band {tb:9.1f} us  ({(tb/ta-1)*100:+.1f} %)")
PY
