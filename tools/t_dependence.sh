#!/bin/bash
# per-launch time of the training chains against the sequence length (row stride of the saved copies)
for T in "$@"; do
  python bench.py --frames $T --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null > gpurun_out/tdep.json
  python - $T <<'PY'
import sys, json
T = int(sys.argv[1])
d = json.loads(open('gpurun_out/tdep.json').read().strip().splitlines()[-1]); b = d["step_breakdown_ms"]
n = T * 49
print(T, "fwd us/launch %.3f  bptt %.3f  batched ms/frame %.4f  inference fwd %.3f" % (
    b["cell_forward_chain"] * 1e3 / n, b["bptt_sequential_pass"] * 1e3 / (n + T),
    b["bptt_time_batched_weight_gradients"] / T, d["forward"]["roofline"]["launch_us"]))
PY
done
