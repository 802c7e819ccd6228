"""Per-kernel table from a rocprofv3 results database (the default output format of `rocprofv3 --kernel-trace --stats`):
python profiles/topk.py <results.db> [steps]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
rows = list(db.execute("select name, total_calls, total_duration, average from top_kernels"))
tot = sum(r[2] for r in rows)
print("# GPU busy per step: %.3f ms (%d steps)" % (tot / steps / 1000.0, steps))
for name, calls, dur, avg in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print("%5.2f%%  %7.1f calls/step  %8.1f us  %7.3f ms/step  %s" % (100.0 * dur / tot, calls / steps, avg, dur / steps / 1000.0, name[:110]))
