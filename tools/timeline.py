"""Condense a rocprofv3 kernel trace (csv) into a per-queue timeline: runs of the same kernel merged, times in ms."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
tend = max(int(r["End_Timestamp"]) for r in rows)
by_q = {}
for r in rows:
    by_q.setdefault(r.get("Queue_Id", "?"), []).append(r)
for q, rs in sorted(by_q.items()):
    rs.sort(key=lambda r: int(r["Start_Timestamp"]))
    print("== queue", q, "launches", len(rs))
    run = None
    prev_end = None
    for r in rs:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("jxlhip::", "")[:40]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (tend - s) / 1e6 > tail_ms:
            continue
        if run and run[0] == name and s - run[2] < 200000:
            run[2] = e; run[3] += 1; run[4] += e - s
        else:
            if run:
                print("  %9.2f -> %9.2f  (%7.2f ms, busy %7.2f, gap before %6.2f) x%-4d %s" % ((run[1] - t0) / 1e6, (run[2] - t0) / 1e6, (run[2] - run[1]) / 1e6, run[4] / 1e6, run[5], run[3], run[0]))
            run = [name, s, e, 1, e - s, ((s - prev_end) / 1e6 if prev_end else 0.0)]
        prev_end = e
    if run:
        print("  %9.2f -> %9.2f  (%7.2f ms, busy %7.2f, gap before %6.2f) x%-4d %s" % ((run[1] - t0) / 1e6, (run[2] - t0) / 1e6, (run[2] - run[1]) / 1e6, run[4] / 1e6, run[5], run[3], run[0]))
