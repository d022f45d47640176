#!/usr/bin/env python3
"""LDS bank model of conv3x3_halo.hip's activation-fragment reads (ds_read_b128; lane groups and banking from MI355X_MICROARCH.md, LDS table):
extra cycles / all cycles of those reads for every pixel pitch P (16-byte slots), per chunks-per-pixel CPT (= Cin / 8) and stride.
`--old-tail`: lanes past the ninth tap read per-lane addresses (round 3) instead of one broadcast address."""
import sys
G0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
G1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
GROUPS = [G0, G1, [l + 32 for l in G0], [l + 32 for l in G1]]
OLD_TAIL = "--old-tail" in sys.argv


def conflicts(CPT, S, P):
    HC = 15 * S + 3
    nk = (9 * CPT * 8 + 63) // 64
    extra = total = 0
    for t in range(2 * nk):
        for G in GROUPS:
            banks = {}
            for lane in G:
                fr, fq = lane & 15, lane >> 4
                c = 4 * t + fq
                if c >= 9 * CPT:
                    slot = (fr * S) * P if OLD_TAIL else -1
                else:
                    tap, cc = divmod(c, CPT)
                    ty, dx = divmod(tap, 3)
                    slot = (fr * S + ty * HC + dx) * P + cc
                banks.setdefault(slot % 16, set()).add(slot)
            m = max(len(v) for v in banks.values())
            extra += m - 1
            total += m
    return extra / total


for S in (1, 2):
    for CPT in (1, 2, 3, 4, 5, 7, 8, 10, 16):
        odd = CPT if CPT % 2 else CPT + 1
        new = CPT
        if S == 1 and CPT > 1:
            while new % 4 != 2:
                new += 1
        else:
            new = odd
        print(f"stride {S} Cin {8 * CPT:3d}: odd pitch {odd:2d} -> {conflicts(CPT, S, odd):.3f}   round-4 pitch {new:2d} -> {conflicts(CPT, S, new):.3f}")
