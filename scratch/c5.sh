#!/bin/bash
DRNMF_RB=${RB:-1} python bench.py --bins 1025 --r 4000 --layers 50 --frames 100 --batch 64 --no-cpu-baseline --no-ista --no-train --no-slab --steps 2 --warmup 1 2>/tmp/err.txt > /tmp/o.json || tail -5 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/o.json'))
print(d["config"], round(d["value"]), round(d["roofline"]["launch_us"],2), round(d["roofline"]["frac"],3), d["finite_positive_masks"])
PY
