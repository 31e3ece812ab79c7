"""Per-API HIP call counts of N steady-state optimiser steps = (trace with N steps) - (trace with 0 steps):
    python tools/hip_api_delta.py base_hip_api_stats.csv with_steps_hip_api_stats.csv N
(both from `rocprofv3 --hip-trace --stats -- python3 tools/step_api_trace.py B T F r K <steps>`)."""
import csv
import sys

base = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(sys.argv[1]))}
full = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(sys.argv[2]))}
n = int(sys.argv[3])
print("HIP API calls made by %d steady-state train_on_batch steps (with-steps minus zero-steps run):" % n)
sync = 0
for name in sorted(set(base) | set(full)):
    d = full.get(name, 0) - base.get(name, 0)
    if d:
        print("  %-44s %+7d  (%.2f per step)" % (name, d, d / n))
    if any(k in name for k in ("Synchronize", "hipMemcpyWithStream", "hipMemcpyDtoH", "hipMemcpy ")) and d > 0:
        sync += d
print("synchronising calls attributable to the steps: %d" % sync)
