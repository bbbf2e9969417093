"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table.
usage: python scripts/prof_summary.py gpurun_out/prof_x/bench_results.db [steps_in_trace] [tail_ms] [marker]
tail_ms > 0 keeps only the kernels that start within the last tail_ms of the trace (the timed steps, leaving out
the warm-up with its one-off autotune trial launches).  With a marker (a kernel-name substring that occurs once per
step, e.g. divide_multi) the window is exactly the last `steps` steps: from the end of the (steps+1)-th last marker
kernel to the end of the last one (tail_ms is then ignored)."""
import sqlite3
import sys

db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tail_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
marker = sys.argv[4] if len(sys.argv) > 4 else None
c = sqlite3.connect(db)
where = ""
if marker:
    ends = [r[0] for r in c.execute("select end from kernels where name like ? order by start", ("%" + marker + "%",))]
    n = int(steps)
    if len(ends) < n + 1:
        raise SystemExit("marker '%s' occurs %d times, need %d" % (marker, len(ends), n + 1))
    where = " where start >= %d and end <= %d" % (ends[-n - 1], ends[-1])
elif tail_ms > 0:
    t_end = list(c.execute("select max(end) from kernels"))[0][0]
    where = " where start >= %d" % (t_end - int(tail_ms * 1e6))
rows = list(c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                      "max(end-start)/1e3 from kernels" + where + " group by name order by 3 desc"))
span = list(c.execute("select (max(end)-min(start))/1e6 from kernels" + where))[0][0]
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace summary of %s%s" % (db, (" (last %.0f ms)" % tail_ms) if tail_ms > 0 else ""))
print("# total kernel time %.3f ms over %g step(s) -> %.3f ms/step; wall span %.3f ms -> %.3f ms/step" %
      (tot, steps, tot / steps, span, span / steps))
# how much of the wall span has at least one kernel running, and how much has two or more (lanes overlapping)
ev = []
for st, en in c.execute("select start, end from kernels" + where):
    ev.append((st, 1))
    ev.append((en, -1))
ev.sort()
depth, last, busy1, busy2 = 0, None, 0, 0
for t, d in ev:
    if last is not None:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
    depth += d
    last = t
print("# device busy (>= 1 kernel running) %.3f ms/step = %.1f %% of the span; >= 2 kernels running %.3f ms/step" %
      (busy1 / 1e6 / steps, 100.0 * busy1 / 1e6 / max(span, 1e-9), busy2 / 1e6 / steps))
print("%-72s %7s %11s %6s %10s %10s %10s" % ("kernel", "calls", "total_ms", "%", "avg_us", "min_us", "max_us"))
for r in rows:
    name = r[0].replace("(anonymous namespace)::", "")
    print("%-72s %7d %11.3f %6.1f %10.1f %10.1f %10.1f" % (name[:72], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5]))
