"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md's HBM
section prescribes: separate passes, values in KiB, FETCH_SIZE doubled for the gfx950 wide-read under-count,
WRITE_SIZE exact for 16-byte-per-lane stores.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --proofs 0
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --proofs 0
    python tools/pmc_traffic.py out/fetch out/write BATCH N > profiles/rNN/x_pmc_traffic.json
"""
import collections
import csv
import glob
import json
import os
import sys


def read(directory, counter):
    per = collections.defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per[name].append((int(r["Grid_Size"]) if r.get("Grid_Size") else 0, float(r["Counter_Value"])))
    return per


def main():
    fetch_dir, write_dir, batch, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    fetch, write = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 2 --warmup 1 "
                      "--no-cpu-baseline --no-profile --proofs 0   (two separate passes; batch %d, n = %d)" % (batch, n),
           "units": "KiB; FETCH_SIZE doubled for the gfx950 wide-read under-count (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact "
                    "for 16-byte-per-lane stores",
           "batch": batch, "n": n, "raw": {"FETCH_SIZE": {}, "WRITE_SIZE": {}}}
    for label, per in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        for name, vals in sorted(per.items()):
            out["raw"][label][name] = {"launches": len(vals), "sum_KiB": sum(v for _, v in vals), "max_KiB": max(v for _, v in vals)}
    # per kernel, template variants of one kernel (k_mle_multifold_mfma<5>, <3>, ...) pooled: the mean is over the same
    # launch mix bench.py's algorithmic bytes per launch are averaged over
    pooled = collections.defaultdict(lambda: {"launches": 0, "bytes": 0.0, "largest": 0.0, "variants": {}})
    for name in sorted(set(fetch) & set(write)):
        if len(fetch[name]) != len(write[name]):
            continue
        per_launch = [(2.0 * f[1] + w[1]) * 1024.0 for f, w in zip(fetch[name], write[name])]
        p = pooled[name.replace("gkr::", "").split("<")[0]]
        p["launches"] += len(per_launch)
        p["bytes"] += sum(per_launch)
        p["largest"] = max(p["largest"], max(per_launch))
        p["variants"][name] = {"launches": len(per_launch), "per_launch_mean_bytes": sum(per_launch) / len(per_launch)}
    for base, p in sorted(pooled.items()):
        out[base] = {"launches": p["launches"], "per_launch_mean_bytes": p["bytes"] / p["launches"],
                     "largest_launch_bytes": p["largest"], "variants": p["variants"]}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
