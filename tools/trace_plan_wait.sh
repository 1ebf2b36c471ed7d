# Does a matrix-core fold pass ever WAIT for its plan (k_mle_fold_plan<J>, built on the side stream from the weights the host
# just wrote)?  Kernel trace of the default bench command, then per fold launch: the end of its plan against the end of the
# kernel before it on its own stream -- the fold could not have started before either.   bash tools/trace_plan_wait.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trpw
rocprofv3 --kernel-trace --output-format csv -d /tmp/trpw -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-verify --proofs 0 > /tmp/trpw.json 2>/dev/null
F=$(ls /tmp/trpw/*/*kernel_trace.csv | head -1)
python3 - "$F" <<'P'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
stream_key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
ev = []
for r in rows:
    n = r["Kernel_Name"]
    m = re.search(r"k_mle_(fold_plan|multifold_mfma)<(\d)>", n)
    kind = (m.group(1), int(m.group(2))) if m else ("other", 0)
    ev.append(dict(s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]), kind=kind, q=r[stream_key], name=n.split("(")[0][-40:],
                   gy=int(r.get("Grid_Size_Y", r.get("Grid_Size_y", "0")) or 0)))
ev.sort(key=lambda x: x["s"])
by_stream = collections.defaultdict(list)
for x in ev:
    by_stream[x["q"]].append(x)
print("streams (%s):" % stream_key, {q: len(v) for q, v in by_stream.items()})
for J in (5, 3):
    plans = [x for x in ev if x["kind"] == ("fold_plan", J)]
    folds = [x for x in ev if x["kind"] == ("multifold_mfma", J)]
    n = min(len(plans), len(folds))
    waited = []
    slack = []
    for p, f in zip(plans[:n], folds[:n]):
        qs = by_stream[f["q"]]
        i = qs.index(f)
        prev_end = qs[i - 1]["e"] if i > 0 else f["s"]
        # the earliest the fold could start: after its stream's previous kernel AND after its plan
        w = max(0, p["e"] - max(prev_end, 0)) if p["e"] > prev_end else 0
        waited.append(w / 1e3)
        slack.append((f["s"] - p["e"]) / 1e3)
    late = [w for w in waited if w > 0]
    dur_p = sorted((p["e"] - p["s"]) / 1e3 for p in plans)
    print("J = %d: %d folds; plan duration us min/median/max %.1f / %.1f / %.1f" % (J, n, dur_p[0], dur_p[len(dur_p) // 2], dur_p[-1]))
    print("   folds whose plan ended AFTER the previous kernel of the fold's stream (the plan was what the fold waited for): %d of %d, "
          "%.1f us in all, worst %.1f us" % (len(late), n, sum(late), max(late) if late else 0.0))
    ss = sorted(slack)
    print("   fold start minus plan end, us: min %.1f, median %.1f, max %.1f" % (ss[0], ss[len(ss) // 2], ss[-1]))
span = (max(x["e"] for x in ev if x["kind"][0] != "other") - min(x["s"] for x in ev if x["kind"][0] != "other")) / 1e3
print("span of the plain-sumcheck kernels in the trace: %.1f us (6 steps)" % span)
P
