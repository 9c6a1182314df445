"""The bf16x6 kernels keep requested tiles in the accumulation registers a[128:255], addressed by
number from inline assembly only (gemm_split.hip).  This scans the compiler's ISA for any access
to that range OUTSIDE the asm statements - the compiler must not place anything of its own there.
usage: python tools/check_staging_regs.py gemm_split.s   (exit code 1 on a violation)"""
import re
import sys


def violations(path):
    out = []
    kernel, in_asm = None, False
    for n, line in enumerate(open(path), 1):
        if re.match(r"^_ZN4marl\S*split_kernel\S*:", line):
            kernel = line.strip().rstrip(":")
        if "s_endpgm" in line:
            kernel = None
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        if kernel is None or in_asm:
            continue
        code = line.split(";")[0]
        for m in re.finditer(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", code):
            hi = int(m.group(2)) if m.group(2) else int(m.group(3))
            if hi >= 128:
                out.append((kernel, n, line.strip()))
    return out


if __name__ == "__main__":
    v = violations(sys.argv[1])
    for k, n, l in v[:20]:
        print(f"{k[:60]}: line {n}: {l}")
    print("violations:", len(v))
    sys.exit(1 if v else 0)
