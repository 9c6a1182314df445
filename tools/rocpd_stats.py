"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total, average.
usage: python tools/rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\[clone.*", "", name)
    name = name.replace("marl::", "").replace("void ", "")
    m = re.match(r"([\w:]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or ""))[:90] if m else name[:90]


def main() -> None:
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {namecol}, (end - start) from kernels").fetchall()
    agg = {}
    for n, d in rows:
        k = short(n)
        a = agg.setdefault(k, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'"{k}",{a[0]},{a[1] / 1e6:.3f},{a[1] / a[0] / 1e3:.2f},{a[2] / 1e3:.2f},'
                     f"{a[3] / 1e3:.2f},{100 * a[1] / tot:.2f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()


def gaps(path: str) -> None:
    """busy time vs wall span of the trace's last third (steady state)"""
    db = sqlite3.connect(path)
    rows = db.execute("select start, end from kernels order by start").fetchall()
    rows = rows[len(rows) * 2 // 3:]
    busy = sum(e - s for s, e in rows)
    span = rows[-1][1] - rows[0][0]
    gap = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
    big = sorted(gap)[-5:]
    print(f"# last third: {len(rows)} kernels, busy {busy / 1e6:.3f} ms, span {span / 1e6:.3f} ms, "
          f"idle {100 * (1 - busy / span):.1f} %, mean gap {sum(gap) / len(gap) / 1e3:.2f} us, "
          f"largest gaps {[round(g / 1e3, 1) for g in big]} us")


if __name__ == "__main__" and len(sys.argv) > 1:
    gaps(sys.argv[1])


def by_grid(path: str, pattern: str) -> None:
    """per (kernel, grid, workgroup) breakdown for kernels whose name contains `pattern`"""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    gcols = [c for c in ("grid_size_x", "grid_size_y", "grid_size_z", "workgroup_size_x", "grid_x",
                         "grid_y", "grid_z", "workgroup_x") if c in cols]
    rows = db.execute(f"select {namecol}, {', '.join(gcols)}, (end - start) from kernels").fetchall()
    agg = {}
    for r in rows:
        if pattern not in r[0]:
            continue
        k = (short(r[0]),) + tuple(r[1:-1])
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += r[-1]
    print("# by grid:", gcols)
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k} calls {a[0]} total_ms {a[1] / 1e6:.3f} avg_us {a[1] / a[0] / 1e3:.2f}")


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] != "--seq":
    by_grid(sys.argv[1], sys.argv[3])


def sequence(path: str, pattern: str, last: int) -> None:
    """durations (us) of the last `last` launches whose kernel name contains `pattern`, in launch order"""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    rows = [(short(n), (e - s) / 1e3) for n, s, e in rows if pattern in n][-last:]
    for n, d in rows:
        print(f"{d:9.2f}  {n}")


if __name__ == "__main__" and len(sys.argv) > 4 and sys.argv[3] == "--seq":
    sequence(sys.argv[1], sys.argv[4], int(sys.argv[5]))
