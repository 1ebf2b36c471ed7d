"""GPU idle time between kernels, from a rocprofv3 --kernel-trace csv: which hand-offs leave the device waiting.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 bench.py --steps 3 --warmup 1 ...
    python tools/trace_gaps.py /tmp/prof/*/*kernel_trace.csv
"""
import collections
import csv
import sys


def short(name):
    name = name.split("(")[0]
    for p in ("void ", "gkr::"):
        name = name.replace(p, "")
    return name[:34]


def main(path, skip_prefix="k_fill_table"):
    rows = [r for r in csv.DictReader(open(path))]
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_Y"]) if "Grid_Size_Y" in r else 0)
                 for r in rows), key=lambda e: e[0])
    # drop the table generation at the front
    while ev and ev[0][2].startswith(skip_prefix):
        ev.pop(0)
    ev = [e for e in ev if not e[2].startswith(skip_prefix)]
    busy_until = ev[0][0]
    last = None
    gaps = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    t_begin = ev[0][0]
    for s, e, name, gy in ev:
        if s > busy_until:
            g = (s - busy_until) / 1e3
            if g < 2000:   # ignore the pauses between steps / phases of the benchmark
                k = (last, name)
                gaps[k][0] += 1
                gaps[k][1] += g
            busy_until = s
        if e > busy_until:
            busy += (e - max(s, busy_until)) / 1e3
            busy_until = e
            last = name
    total = (busy_until - t_begin) / 1e3
    print("span %.1f us, busy %.1f us" % (total, busy))
    for k, (n, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print("%-34s -> %-34s n=%4d  idle %9.1f us  (avg %6.1f)" % (k[0], k[1], n, g, g / n))


if __name__ == "__main__":
    main(sys.argv[1])
