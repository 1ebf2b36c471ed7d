"""Per-placement PMC counters of the fold pass: reads the counter_collection CSVs of
    rocprofv3 --pmc <counters> --output-format csv -d DIR -- python3 tools/alloc_mode_probe.py 256
(one DIR per counter set) and prints, per trial of the probe (a burst of k_fill_table dispatches starts a trial),
the mean counter values of the 2^20 -> 2^15 fold launches and of the pass-0 launches, next to the speeds the probe
itself printed.

    python tools/pmc_modes.py probe_stdout.txt DIR [DIR ...]
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def trials_of(directory):
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    per_dispatch = collections.OrderedDict()
    for r in rows:
        d = per_dispatch.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r.get("Grid_Size") or 0), "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    trials, cur, in_fill = [], None, False
    for d in per_dispatch.values():
        fill = "k_fill_table" in d["name"]
        if fill and not in_fill:
            cur = []
            trials.append(cur)
        in_fill = fill
        if cur is not None and not fill:
            cur.append(d)
    return trials


def main():
    speeds = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"fold (\d+) GB/s  first pass (\d+) GB/s", open(sys.argv[1]).read())]
    out = []
    per_trial = collections.defaultdict(dict)
    for directory in sys.argv[2:]:
        for t, disp in enumerate(trials_of(directory)):
            for label, pick in (("fold5", lambda d: "k_mle_multifold_mfma<5>" in d["name"]), ("pass0", lambda d: "k_mle_sub_sums" in d["name"])):
                sel = [d for d in disp if pick(d)]
                if not sel:
                    continue
                big = max(d["grid"] for d in sel)
                sel = [d for d in sel if d["grid"] == big]
                for cname in sel[0]["c"]:
                    per_trial[t]["%s.%s" % (label, cname)] = sum(d["c"][cname] for d in sel) / len(sel)
    for t in sorted(per_trial):
        row = {"trial": t}
        if t < len(speeds):
            row["fold_GBps"], row["pass0_GBps"] = speeds[t]
        row.update({k: round(v, 1) for k, v in sorted(per_trial[t].items())})
        out.append(row)
        print(json.dumps(row))


if __name__ == "__main__":
    main()
