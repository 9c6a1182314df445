"""Every kernel of a hipcc -S listing: waits INSIDE loops - per loop the counts of full drains (vmcnt(0), lgkmcnt(0))
next to its MFMA / VMEM / DS counts.  A `vmcnt(0)` inside a pipelined loop is what to look for (perf debugging aid:
round 6 found hipcc draining an LDS-DMA ring this way).    python tools/isa_loop_waits.py file.s [name-filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):", s, re.M):
    name = m.group(1)
    if flt not in name:
        continue
    i = m.start()
    j = s.find(".Lfunc_end", i)
    loops = {}
    cur = None
    for l in s[i:j].split("\n"):
        t = l.strip()
        mm = re.match(r"^\.LBB\d+_\d+:(.*)", t)
        if mm:
            h = re.search(r"Header[:=]\s*(BB\d+_\d+)?", mm.group(1))
            lp = re.search(r"Loop Header|in Loop: Header=(BB\d+_\d+)", mm.group(1))
            if "Loop" in mm.group(1):
                hm = re.search(r"Header=(BB\d+_\d+)", mm.group(1))
                cur = hm.group(1) if hm else t.split(":")[0].lstrip(".L")
            else:
                cur = None
            continue
        if cur is None or not t or t.startswith((";", ".")):
            continue
        d = loops.setdefault(cur, dict(mfma=0, vmem=0, ds=0, vm0=0, lgkm0=0, bar=0, n=0))
        d["n"] += 1
        op = t.split()[0]
        if op.startswith("v_mfma"):
            d["mfma"] += 1
        elif op.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "scratch_")):
            d["vmem"] += 1
        elif op.startswith("ds_"):
            d["ds"] += 1
        elif op == "s_barrier":
            d["bar"] += 1
        elif op == "s_waitcnt":
            if "vmcnt(0)" in t:
                d["vm0"] += 1
            if "lgkmcnt(0)" in t:
                d["lgkm0"] += 1
    hot = {k: v for k, v in loops.items() if v["vm0"] and (v["mfma"] or v["vmem"])}
    if hot:
        print(name[:110])
        for k, v in hot.items():
            print("   loop", k, v)
