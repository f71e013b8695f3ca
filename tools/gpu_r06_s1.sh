#!/bin/bash
# r06 session 1: which third-party packages the GPU box has, host description, three consecutive default bench lines
mkdir -p gpurun_out/r06_s1
O=gpurun_out/r06_s1
python - > $O/packages.json 2>$O/packages.err <<'PY'
import importlib, json, platform, os
out = {}
for n in ["torchaudio", "librosa", "resampy", "smplx", "soxr", "python_speech_features", "soundfile", "scipy", "numpy", "transformers", "torch"]:
    try:
        m = importlib.import_module(n)
        out[n] = getattr(m, "__version__", "present")
    except Exception as e:
        out[n] = f"ABSENT ({type(e).__name__}: {str(e)[:80]})"
out["cpu_count"] = os.cpu_count()
out["platform"] = platform.platform()
try:
    out["cpu_model"] = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception as e:
    out["cpu_model"] = str(e)
print(json.dumps(out, indent=1))
PY
cat $O/packages.json
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>>$O/bench.err | tail -1 >> $O/bench_lines.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_s1/bench_lines.jsonl"):
    d = json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("selfcheck"))
PY
