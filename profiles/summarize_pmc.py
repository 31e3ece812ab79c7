#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes.

usage: summarize_pmc.py OUT.json PASS_DIR [PASS_DIR ...]

Each PASS_DIR is the `-d` directory of ONE `rocprofv3 --kernel-trace --pmc <counters> -- python3
bench.py ...` run (counters are collected in separate passes: FETCH_SIZE and WRITE_SIZE do not fit
one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output: {kernel: {counter: {n, mean}}} plus
the derived per-launch HBM-side traffic with the guide's gfx950 correction (FETCH_SIZE is in KB
and under-reports wide coalesced reads by 2x: bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE in KB,
uncalibrated, taken at face value).
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    """'void (anonymous namespace)::cell_b_kernel<8, 8>((anonymous namespace)::CellBArgs)' ->
    'cell_b_kernel<8,8>'; long library kernel names are cut at 80 characters."""
    n = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void\s+)?([A-Za-z_0-9:]+)(<[^(]*>)?", n)
    if not m:
        return n[:80]
    base = m.group(1).split("::")[-1]
    targs = (m.group(2) or "").replace(" ", "")
    return (base + targs)[:80]


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    c = row["Counter_Name"]
                    v = float(row["Counter_Value"])
                    s = acc.setdefault(k, {}).setdefault(c, [0, 0.0])
                    s[0] += 1
                    s[1] += v
    res = {}
    for k, cs in sorted(acc.items()):
        res[k] = {c: {"n": n, "mean": tot / n} for c, (n, tot) in sorted(cs.items())}
        fs = res[k].get("FETCH_SIZE")
        ws = res[k].get("WRITE_SIZE")
        if fs:
            res[k]["hbm_side_read_bytes_per_launch"] = fs["mean"] * 1024.0 * 2.0
        if ws:
            res[k]["hbm_side_write_bytes_per_launch"] = ws["mean"] * 1024.0
    # launch-weighted mean over the two cell kernels (what bench.py's roofline.traffic reports)
    tot_n, tot_b = 0, 0.0
    for k, v in res.items():
        if k.startswith("cell_a_kernel") or k.startswith("cell_b_kernel"):
            if "hbm_side_read_bytes_per_launch" in v:
                n = v["FETCH_SIZE"]["n"]
                b = v["hbm_side_read_bytes_per_launch"] + v.get("hbm_side_write_bytes_per_launch", 0.0)
                tot_n += n
                tot_b += n * b
    if tot_n:
        res["_cell_launch_mean_traffic_bytes"] = tot_b / tot_n
    # which library the counters belong to: content hash of libdrnmf's sources and flags (the same
    # one build.py stamps the .so with).  bench.py reports the figure as `roofline.traffic` only while
    # the library it runs is built from exactly these sources.
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dr-nmf_amd"))
        import build as _b
        res["_lib_src_sha16"] = _b._src_hash()[:16]
    except Exception as e:       # noqa: BLE001
        res["_lib_src_sha16"] = None
        res["_lib_src_sha16_error"] = repr(e)[:200]
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k.startswith("_")}))


if __name__ == "__main__":
    main()
