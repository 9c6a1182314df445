"""Control-flow skeleton of one kernel in a hipcc -S listing: per basic block the MFMA / ds_read / LDS-DMA / barrier /
waitcnt counts and the branch that ends it; only blocks inside loops (or with MFMAs) are shown.
usage: python tools/isa_skel.py file.s mangled_name_substring [all]"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"^(_Z\w*" + re.escape(sys.argv[2]) + r"\w*):", s, re.M)
i = m.start()
j = s.index(".Lfunc_end", i)
show_all = len(sys.argv) > 3
cur, inloop = "entry", False
cnt = {}


def flush(br=""):
    global cnt
    if cnt and (show_all or inloop or cnt.get("mfma")):
        print(f"{cur:11s} " + " ".join(f"{k}={v}" for k, v in cnt.items()) + (" -> " + br if br else ""))
    cnt = {}


print(m.group(1)[:100])
for l in s[i:j].split("\n"):
    t = l.strip()
    mm = re.match(r"^(\.LBB\d+_\d+):(.*)", t)
    if mm:
        flush()
        cur, inloop = mm.group(1), "Loop" in mm.group(2)
        continue
    if not t or t.startswith((";", ".")):
        continue
    op = t.split()[0]
    key = ("mfma" if op.startswith("v_mfma") else "ds_read" if op.startswith("ds_read") else "ds_write" if op.startswith("ds_write")
           else "dma" if op.startswith("buffer_load") and "lds" in t else "vmem" if op.startswith(("global_", "buffer_", "scratch_"))
           else "bar" if op == "s_barrier" else None)
    if key:
        cnt[key] = cnt.get(key, 0) + 1
    if op == "s_waitcnt":
        cnt.setdefault("waits", [])
        cnt["waits"].append(t.replace("s_waitcnt ", ""))
    if op.startswith(("s_cbranch", "s_branch")):
        flush(t)
flush()
