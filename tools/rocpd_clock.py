"""Effective shader clock and MFMA-busy at that clock, per (kernel, grid size), from a rocprofv3 --pmc pass that
collected GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES.
  clock [GHz]      = GRBM_GUI_ACTIVE / 8 XCDs / duration       (the counter is summed over the XCDs)
  busy at clock    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
  busy at 2.4 GHz  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * 2.4e3 * duration_us)
usage: python tools/rocpd_clock.py results.db [out.csv] [min_us]"""
import sqlite3
import sys

from rocpd_pmc import short


def main() -> None:
    db = sqlite3.connect(sys.argv[1])
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    rows = db.cursor().execute(
        "select kernel_name, grid_size, workgroup_size, counter_name, count(*), avg(value), avg(duration), min(start) "
        "from counters_collection group by kernel_name, grid_size, workgroup_size, counter_name").fetchall()
    table = {}
    for k, g, w, c, n, v, d, t0 in rows:
        e = table.setdefault((short(k), g // max(w, 1)), {"calls": n, "us": (d or 0) / 1e3, "t0": t0})
        e[c] = v
    lines = ["kernel,workgroups,calls,avg_us,clock_ghz,mfma_busy_at_clock,mfma_busy_at_2p4,GRBM_GUI_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES"]
    for (k, wg), e in sorted(table.items(), key=lambda kv: kv[1]["t0"]):
        if e["us"] < min_us:
            continue
        grbm, busy = e.get("GRBM_GUI_ACTIVE", 0.0), e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        clk = grbm / 8.0 / (e["us"] * 1e3) if e["us"] > 0 else 0.0
        b_clk = busy / (1024.0 * grbm / 8.0) if grbm > 0 else 0.0
        b_24 = busy / (1024.0 * 2.4e3 * e["us"]) if e["us"] > 0 else 0.0
        lines.append(f'"{k}",{wg},{e["calls"]},{e["us"]:.2f},{clk:.3f},{b_clk:.3f},{b_24:.3f},{grbm:.0f},{busy:.0f}')
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
