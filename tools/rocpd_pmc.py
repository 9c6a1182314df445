"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite file.
usage: python tools/rocpd_pmc.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\[clone.*", "", name).replace("marl::", "").replace("void ", "")
    m = re.match(r"([\w:]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or ""))[:70] if m else name[:70]


def main() -> None:
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select kernel_name, counter_name, count(*), avg(value), avg(duration) "
        "from counters_collection group by kernel_name, counter_name").fetchall()
    table, counters = {}, []
    for k, c, n, v, d in rows:
        e = table.setdefault(short(k), {"calls": n, "dur_us": (d or 0) / 1e3})
        e[c] = v
        if c not in counters:
            counters.append(c)
    lines = ["kernel,calls,avg_us," + ",".join(counters)]
    for k, e in sorted(table.items(), key=lambda kv: -kv[1]["calls"] * kv[1]["dur_us"]):
        lines.append(f'"{k}",{e["calls"]},{e["dur_us"]:.2f},' + ",".join(f"{e.get(c, 0):.0f}" for c in counters))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
