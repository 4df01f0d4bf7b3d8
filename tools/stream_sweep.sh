#!/bin/bash
for n in 1 2 3 4; do
  DGMI_BENCH_STREAMS=$n python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/streams_$n.json
  python - <<PY
import json
d = json.load(open("gpurun_out/streams_$n.json"))
print("streams $n: ms/step", round(d["ms_per_step"], 3), "Gedge/s", round(d["value"] / 1e9, 2), "dominant pair avg ms", round(d["roofline"]["avg_launch_ms"], 4))
PY
done
