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
