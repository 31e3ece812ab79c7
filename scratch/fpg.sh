#!/bin/bash
for f in 1 2 4 8 20; do
  DRNMF_FPG=$f python bench.py --no-cpu-baseline --no-ista --no-train --no-slab --steps 2 --warmup 1 2>/dev/null > /tmp/o.json
  python - <<PY
import json
d=json.load(open('/tmp/o.json'))
print($f, round(d["value"]), round(d["roofline"]["launch_us"],3), d["finite_positive_masks"])
PY
done
