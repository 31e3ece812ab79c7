#!/bin/bash
# frames/s of the headline model for batch sizes x row blocks per workgroup (DRNMF_RB); run from the repo root
for b in 128 256 512 1024; do for rb in 1 2; do
  DRNMF_RB=$rb python bench.py --batch $b --frames 200 --no-cpu-baseline --no-ista --no-train --steps 2 --warmup 1 2>/dev/null > /tmp/o.json
  python - <<PY
import json
d=json.load(open('/tmp/o.json'))
print(d["config"]["B_per_gpu"], $rb, round(d["value"]), round(d["roofline"]["launch_us"],2), round(d["roofline"]["frac"],3))
PY
done; done
