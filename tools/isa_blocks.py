"""Basic blocks of one kernel in a hipcc -S listing with their MFMA / DS / VMEM counts, and the compact opcode stream
of the blocks that hold at least `min_mfma` MFMAs (perf debugging aid).
usage: python tools/isa_blocks.py file.s mangled_name_substring [min_mfma]"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = re.search(r"^(_Z\w*" + re.escape(pat) + r"\w*):", s, re.M)
i = m.start()
j = s.index("s_endpgm", i)
blocks, cur, name = [], [], "entry"
for l in s[i:j].split("\n"):
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append((name, cur))
        cur, name = [], t
    elif t and not t.startswith((";", ".", "_")):
        w = t.split()
        cur.append(w[0] + (" " + " ".join(w[1:]) if w[0] in ("s_waitcnt", "s_barrier", "s_setprio") else ""))
blocks.append((name, cur))
print(m.group(1)[:90])
for n, b in blocks:
    mm = sum(1 for x in b if x.startswith("v_mfma"))
    if mm >= 6:
        print(n, "instrs", len(b), "mfma", mm, "ds_read", sum(1 for x in b if x.startswith("ds_read")), "lds-dma",
              sum(1 for x in b if x.startswith("buffer_load")), "scratch", sum(1 for x in b if x.startswith("scratch")))
for n, b in blocks:
    if sum(1 for x in b if x.startswith("v_mfma")) >= min_mfma:
        out, prev, c = [], None, 0
        for o in b + [None]:
            if o == prev:
                c += 1
            else:
                if prev:
                    out.append(prev + (f"x{c}" if c > 1 else ""))
                prev, c = o, 1
        print("\n" + n + "\n" + " | ".join(out))
        break
