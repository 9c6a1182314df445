"""LDS bank model of MI355X_MICROARCH.md (LDS table) applied to this library's layouts: cycles of one
wave-instruction = sum over its lane groups of the most-loaded bank's distinct addresses.
  python tools/lds_conflicts.py gemm     staging stores / fragment reads of the bf16x6 kernels
  python tools/lds_conflicts.py cnn      searches the row / patch pads of the cnn_fwd2 images
  python tools/lds_conflicts.py cnn3     the same for the cnn_fwd3 (AidCnn) images
The pads found here are the RP1.. / PP template arguments of Fwd2Net (csrc/cnn.hip)."""
import sys

R128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
        [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
R128 += [[l + 32 for l in g] for g in R128]
GROUPS = {"r128": (R128, 64), "r32": ([list(range(32)), list(range(32, 64))], 32),
          "w64": ([list(range(16 * g, 16 * g + 16)) for g in range(4)], 32),
          "w128": ([list(range(8 * g, 8 * g + 8)) for g in range(8)], 32)}


def cycles(addrs, width, kind):
    """addrs: byte address of each of the 64 lanes; width: bytes per lane."""
    groups, nb = GROUPS[kind]
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            for d in range(width // 4):
                dw = addrs[l] // 4 + d
                banks.setdefault(dw % nb, set()).add(dw)
        tot += max(len(s) for s in banks.values())
    return tot


# ---- bf16x6 GEMM staging (gemm_split.hip) ----------------------------------------------------------
def gemm():
    srow = 80
    old = [(l // 8) * srow + (l % 8) * 8 for l in range(64)]
    new = [((l >> 4) + 4 * ((l >> 3) & 1)) * srow + (l % 8) * 8 for l in range(64)]
    print("fp32 tile store (ds_write_b64): rows r, r+1 per group", cycles(old, 8, "w64"),
          "| rows r, r+4 (stage_row)", cycles(new, 8, "w64"), "| ideal 4")

    def image_unit(c, rw):
        wave, u = (c & 255) >> 6, ((c & 63) >> 2) + 16 * (c >> 8)
        v = u >> 1
        rb, plane = v % (rw // 2), v // (rw // 2)
        return wave * rw + (rb >> 2) * 8 + (rb & 3) + 4 * (u & 1), plane

    for bn, ch in ((128, 6), (64, 3)):
        bpl, t_old, t_new = bn * srow, 0, 0
        for w in range(4):
            for i in range(ch):
                a_old, a_new = [], []
                for l in range(64):
                    c = w * 64 + l + 256 * i
                    a_old.append((c % 12 // 4) * bpl + (c // 12) * srow + (c % 4) * 16)
                    row, pl = image_unit(c, bn // 4)
                    a_new.append(pl * bpl + row * srow + (c & 3) * 16)
                t_old += cycles(a_old, 16, "w128")
                t_new += cycles(a_new, 16, "w128")
        print(f"weight-image tile copy BN={bn} (ds_write_b128): linear", t_old, "| image_unit", t_new,
              "| ideal", 4 * ch * 8)
    print("fragment read (ds_read_b128, 80-byte rows):",
          cycles([(l % 32) * srow + (l >> 5) * 16 for l in range(64)], 16, "r128"), "| ideal 4")


# ---- cnn_fwd2 images (cnn.hip, Fwd2Net) ------------------------------------------------------------
class Net:
    def __init__(self, f, chs, rp=(0, 0, 0, 0), pp=0, cp1=4, ldw_pad=8):
        self.F, self.ch, self.L, self.rp, self.pp, self.cp1, self.ldw_pad = f, chs, len(chs) - 1, rp, pp, cp1, ldw_pad

    def hin(self, l):
        h = self.F
        for _ in range(l):
            h = (h - 1) // 2 + 1
        return h

    def hout(self, l): return (self.hin(l) - 1) // 2 + 1
    def cin(self, l): return self.ch[l]
    def cout(self, l): return self.ch[l + 1]
    def hp(self, l): return self.hin(l) + 2
    def cs(self, l): return self.cin(0) if l == 0 else self.cin(l) + (self.cp1 if l == 1 else 4)
    def rs(self, l): return self.hp(l) * self.cs(l) + (self.rp[l] if l > 0 else 0)
    def in_per(self, l): return (self.hp(l) * self.rs(l) + 3) & ~3
    def steps(self, l): return (9 * self.cin(l) + 15) // 16
    def ldw(self, l): return self.steps(l) * 16 + self.ldw_pad
    def per_patch(self): return self.in_per(0) + max([self.in_per(l) for l in range(1, self.L)] + [0]) + self.pp


def b_frag(n, l, nt, kk):
    return cycles([4 * ((nt * 16 + (lane & 15)) * n.ldw(l) + kk * 16 + 4 * (lane >> 4)) for lane in range(64)],
                  16, "r128")


def layer_wave(n, l):
    """a wave owns a patch: lanes = positions of one patch"""
    P, hout, cs, rs, cin, K = n.hout(l) ** 2, n.hout(l), n.cs(l), n.rs(l), n.cin(l), 9 * n.cin(l)
    tot = ideal = 0
    for kk in range(n.steps(l)):
        for mt in range((P + 15) // 16):
            for j in range(1 if l > 0 else 4):
                ad = []
                for lane in range(64):
                    quad, m = lane >> 4, mt * 16 + (lane & 15)
                    m = m if m < P else 0
                    rbase = 2 * (m // hout) * rs + 2 * (m % hout) * cs
                    if l > 0:
                        k0 = kk * 16 + 4 * quad
                        tap, ci = min(k0 // cin, 8), k0 % cin
                    else:
                        k = min(kk * 16 + 4 * quad + j, K - 1)
                        tap, ci = k // cin, k % cin
                    ad.append(4 * (rbase + (tap // 3) * rs + (tap % 3) * cs + ci))
                tot += cycles(ad, 16 if l > 0 else 4, "r128" if l > 0 else "r32")
                ideal += 4 if l > 0 else 2
        for nt in range((n.cout(l) + 15) // 16):
            tot += b_frag(n, l, nt, kk)
            ideal += 4
    return tot, ideal


def layer_rows4(n, l, patch_stride):
    """16-row tiles = 4 patches x 4 positions (last layer, P == 4)"""
    hout, cs, rs, cin = n.hout(l), n.cs(l), n.rs(l), n.cin(l)
    tot = ideal = 0
    for kk in range(n.steps(l)):
        ad = []
        for lane in range(64):
            quad, l16 = lane >> 4, lane & 15
            lr, pos = l16 >> 2, l16 & 3
            k0 = kk * 16 + 4 * quad
            tap, ci = min(k0 // cin, 8), k0 % cin
            ad.append(4 * (lr * patch_stride + 2 * (pos // hout) * rs + 2 * (pos % hout) * cs
                           + (tap // 3) * rs + (tap % 3) * cs + ci))
        tot += cycles(ad, 16, "r128") + b_frag(n, l, 0, kk)
        ideal += 8
    return tot, ideal


def search(name, f, chs, rows4_last, cp1s=(4,), stride_of=None):
    L = len(chs) - 1
    best = None
    for cp1 in cp1s:
        for rp1 in range(0, 64, 4):
            for rp2 in (range(0, 64, 4) if L > 2 else [0]):
                for rp3 in (range(0, 64, 4) if L > 3 else [0]):
                    for pp in (range(0, 64, 4) if rows4_last else [0]):
                        n = Net(f, chs, (0, rp1, rp2, rp3), pp, cp1)
                        stride = stride_of(n) if stride_of else n.per_patch()
                        res = [layer_rows4(n, l, stride) if rows4_last and l == L - 1 else layer_wave(n, l)
                               for l in range(L)]
                        key = (sum(r[0] for r in res), rp1 + rp2 + rp3 + pp + cp1)
                        if best is None or key < best[0]:
                            best = (key, dict(rp=(rp1, rp2, rp3), pp=pp, cp1=cp1, per_layer=res, stride=stride))
    n0 = Net(f, chs, ldw_pad=4)
    base = [layer_rows4(n0, l, n0.per_patch()) if rows4_last and l == L - 1 else layer_wave(n0, l) for l in range(L)]
    print(name, "| unpadded (ldw + 4):", base, "| best:", best[1])


def cnn():
    search("Resisc f=12", 12, [3, 16, 32, 64], True)
    search("Mnist f=6", 6, [1, 8, 16], False)
    search("Mnist f=12", 12, [1, 8, 16], False)




def search3(name, f, chs, cp1s=(0, 4)):
    """cnn_fwd3 (workgroup-owned patch groups): layers 1..L-2 independently, then the packed last layer."""
    L = len(chs) - 1
    for cp1 in cp1s:
        rp = [0] * 4
        tot = 0
        for l in range(1, L - 1):
            cand = []
            for r in range(0, 64, 4):
                rr = list(rp)
                rr[l] = r
                cand.append((layer_wave(Net(f, chs, tuple(rr), 0, cp1), l)[0], r))
            c, rp[l] = min(cand)
            tot += c
        cand = []
        for r3 in range(0, 64, 4):
            for pp in range(0, 64, 4):
                rr = list(rp)
                rr[L - 1] = r3
                n = Net(f, chs, tuple(rr), pp, cp1)
                ev = max(n.in_per(l) for l in range(0, L, 2))
                od = max(n.in_per(l) for l in range(1, L, 2))
                cand.append((layer_rows4(n, L - 1, ev + od + pp)[0], r3 + pp, r3, pp, ev + od + pp))
        c, _, rp[L - 1], pp, stride = min(cand)
        n = Net(f, chs, tuple(rp), pp, cp1)
        lds = 4 * stride * 4 + 4 * (16 * n.ldw(0) + 48)
        print(name, "cp1", cp1, "rp", rp, "pp", pp, "cycles", tot + c, "patch floats", stride, "LDS bytes", lds)


def cnn3():
    search3("Aid f=24", 24, [3, 16, 32, 64, 128])
    search3("Aid f=32", 32, [3, 16, 32, 64, 128])


if __name__ == "__main__":
    {"gemm": gemm, "cnn": cnn, "cnn3": cnn3}[sys.argv[1] if len(sys.argv) > 1 else "gemm"]()
