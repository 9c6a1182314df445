"""Sum of kernel time and launch count of a rocprofv3 kernel trace (rocpd sqlite): python tools/ksum.py results.db ITERATIONS"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
its = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n, tot, t0, t1 = db.execute("select count(*), sum(end - start), min(start), max(end) from kernels").fetchone()
print(f"launches {n} ({n / its:.1f} per iteration), kernel time {tot / 1e6:.3f} ms ({tot / 1e6 / its:.3f} per iteration), "
      f"span {(t1 - t0) / 1e6:.3f} ms")
