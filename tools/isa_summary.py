"""Opcode-sequence summary of one kernel in a hipcc -S listing (perf debugging aid)."""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else None
for m in re.finditer(r"^(_Z\w+):", s, re.M):
    name = m.group(1)
    i = m.start()
    j = s.index("s_endpgm", i)
    body = s[i:j]
    print(name[:70], "saveexec:", body.count("s_and_saveexec"), "gload:", body.count("global_load_dwordx4"),
          "scratch:", body.count("scratch_"), "mfma:", body.count("v_mfma"))
    if pat and pat in name:
        ops = [l.strip().split()[0] for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".", "_"))]
        out, prev, cnt = [], None, 0
        for o in ops:
            if o == prev:
                cnt += 1
            else:
                if prev:
                    out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
                prev, cnt = o, 1
        txt = " ".join(out)
        k = txt.find("v_mfma")
        print(txt[max(0, k - 900): k + 1100])
