"""LDS bank-conflict model for the A-fragment gathers of the conv stack kernels (ds_read_b128).

Lane groups and banking as measured on MI355X: a wave-wide ds_read_b128 is served in four groups of 16 lanes,
{0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}; bank = dword address mod 64;
a group costs one LDS cycle per distinct dword address on its busiest bank.  Prints LDS cycles per instruction
(4 = conflict-free) for candidate pixel pitches."""
import itertools, sys

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def instr_cycles(addr_of_lane):
    """addr_of_lane: 64 dword addresses (start of 4 consecutive dwords)."""
    cyc = 0
    for g in GROUPS:
        banks = {}
        for l in g:
            for d in range(4):
                a = addr_of_lane[l] + d
                banks.setdefault(a % 64, set()).add(a)
        cyc += max(len(v) for v in banks.values())
    return cyc


def layer(name, nrows, out_w, in_w, stride, taps, pitch_dw, cg_off, halves=1, pad_row=None):
    """rows m -> input pixel (stride*oy + ky)*in_w + stride*ox + kx; lane (l15, g4) reads channel group g4 (+4*half)."""
    tot, n = 0, 0
    for rb in range((nrows + 15) // 16):
        for (ky, kx) in taps:
            for half in range(halves):
                addrs = []
                for lane in range(64):
                    l15, g4 = lane & 15, lane >> 4
                    m = rb * 16 + l15
                    if m >= nrows:
                        m = 0
                    oy, ox = divmod(m, out_w)
                    pix = (stride * oy + ky) * in_w + stride * ox + kx
                    addrs.append(pix * pitch_dw + cg_off[g4 + 4 * half])
                tot += instr_cycles(addrs)
                n += 1
    return tot / n


if __name__ == "__main__":
    taps4 = [(a, b) for a in range(4) for b in range(4)]
    taps3 = [(a, b) for a in range(3) for b in range(3)]
    print("conv2 reads of a1 (20x20x32, stride 2): pitch dwords -> cycles/instr")
    for pitch in range(16, 41, 4):
        print("  pitch %2d: %.2f" % (pitch, layer("c2", 81, 9, 20, 2, taps4, pitch, [0, 4, 8, 12])))
    print("conv3 reads of a2 (9x9x64, stride 1): pitch dwords -> cycles/instr")
    for pitch in range(32, 57, 4):
        print("  pitch %2d: %.2f" % (pitch, layer("c3", 49, 7, 9, 1, taps3, pitch, [0, 4, 8, 12, 16, 20, 24, 28], halves=2)))


def gather(nrows, row_w, src_of, taps, halves, RP, CP):
    """generic: rows m -> (y, x) = divmod(m, row_w); src_of(y, x, tap) -> (sy, sx) in the stored grid;
    lane (l15, g4) reads dword address sy*RP + sx*CP + 4*(g4 + 4*half)."""
    tot = n = 0
    for rb in range((nrows + 15) // 16):
        for tap in taps:
            for half in range(halves):
                addrs = []
                for lane in range(64):
                    l15, g4 = lane & 15, lane >> 4
                    m = rb * 16 + l15
                    if m >= nrows:
                        m = 0
                    y, x = divmod(m, row_w)
                    sy, sx = src_of(y, x, tap)
                    addrs.append(sy * RP + sx * CP + 4 * (g4 + 4 * half))
                tot += instr_cycles(addrs)
                n += 1
    return tot / n


def search_dgrad():
    """Backward (data-gradient) gathers over zero-bordered 11x11 grids of 64-channel pixels."""
    taps3 = [(a, b) for a in range(3) for b in range(3)]
    taps2 = [(a, b) for a in range(2) for b in range(2)]
    res3, res2 = [], []
    for CP in range(32, 49, 4):
        for RP in range(11 * CP, 11 * CP + 68, 4):
            c3 = gather(81, 9, lambda y, x, t: (y + 2 - t[0], x + 2 - t[1]), taps3, 2, RP, CP)
            c2 = gather(100, 10, lambda y, x, t: (y + 1 - t[0], x + 1 - t[1]), taps2, 2, RP, CP)
            res3.append((c3, CP, RP))
            res2.append((c2, CP, RP))
    res3.sort(); res2.sort()
    print("conv3^T gathers of da3 (cycles, pixel pitch, row pitch in dwords):", res3[:6])
    print("conv2^T gathers of da2:", res2[:6])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "dgrad":
    search_dgrad()
