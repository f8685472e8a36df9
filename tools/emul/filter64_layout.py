#!/usr/bin/env python3
"""
Lane-level model of the LDS placement of `score_filter64_kernel` (csrc/topk.hip): where the
`global_load_lds_dwordx4` instructions put the user panel and an item slab (LDS side linear in the
lane, swizzle chosen on the global side) and where the MFMA operand fetches read them back.

    python tools/emul/filter64_layout.py      # checks every (row, feature) and prints bank loads
"""
from collections import Counter

import numpy as np


def place_users(U):
    "U: [128, 64] -> LDS image [8192]; instruction n = 8 wave + q moves rows 4 n .. 4 n + 3"
    lds = np.full(128 * 64, -1, dtype=U.dtype)
    for n in range(32):
        for lane in range(64):
            row = n * 4 + (lane >> 4)
            c = (lane & 15) ^ (row & 15)  # chunk of the row this lane names
            lds[n * 256 + lane * 4 : n * 256 + lane * 4 + 4] = U[row, c * 4 : c * 4 + 4]
    return lds


def place_slab(I, s):
    """
    I: [rows, KP], features 16 s .. 16 s + 15 -> LDS image [rows * 16]; instruction n moves rows
    16 n ..  (256 item rows; also the 128 user rows of score_filter_slab_kernel, -DLK_TOPK_DMA=2)
    """
    lds = np.full(I.shape[0] * 16, -1, dtype=I.dtype)
    for n in range(I.shape[0] // 16):
        for lane in range(64):
            row = n * 16 + (lane >> 2)
            c = (lane & 3) ^ ((lane >> 4) & 3)  # == (row >> 2) & 3
            lds[n * 256 + lane * 4 : n * 256 + lane * 4 + 4] = I[row, 16 * s + c * 4 : 16 * s + c * 4 + 4]
    return lds


def a_address(wave, lane, ut, s, kk):
    "float address of A[i = lane & 31][k = lane >> 5] for step kk of slab s"
    wu, r, h = (wave & 1) * 64, lane & 31, lane >> 5
    return (wu + r) * 64 + (((4 * s + (kk >> 2)) ^ (r & 15)) << 2) + h + ut * 32 * 64 + (kk & 2)


def a_slab_address(wave, lane, ut, kk):
    "score_filter_slab_kernel: the user slab is laid out like the item slab"
    wu, r, h = (wave & 1) * 64, lane & 31, lane >> 5
    return (wu + r) * 16 + (((kk >> 2) ^ ((r >> 2) & 3)) << 2) + h + ut * 32 * 16 + (kk & 2)


def b_address(wave, lane, t, kk):
    wi, r, h = (wave >> 1) * 128, lane & 31, lane >> 5
    return (wi + r) * 16 + (((kk >> 2) ^ ((r >> 2) & 3)) << 2) + h + t * 32 * 16 + (kk & 2)


def check():
    U = np.arange(128 * 64, dtype=np.int64).reshape(128, 64)
    I = 100000 + np.arange(256 * 64, dtype=np.int64).reshape(256, 64)
    lu = place_users(U)
    assert (lu >= 0).all()
    for wave in range(4):
        for lane in range(64):
            for ut in range(2):
                for s in range(4):
                    for kk in range(0, 16, 2):
                        row = (wave & 1) * 64 + ut * 32 + (lane & 31)
                        assert lu[a_address(wave, lane, ut, s, kk)] == U[row, 16 * s + kk + (lane >> 5)]
    for s in range(4):
        li = place_slab(I, s)
        assert (li >= 0).all()
        for wave in range(4):
            for lane in range(64):
                for t in range(4):
                    for kk in range(0, 16, 2):
                        row = (wave >> 1) * 128 + t * 32 + (lane & 31)
                        assert li[b_address(wave, lane, t, kk)] == I[row, 16 * s + kk + (lane >> 5)]
    U2 = np.arange(128 * 256, dtype=np.int64).reshape(128, 256)  # KP = 256: 16 slabs
    for s in (0, 7, 15):
        lus = place_slab(U2, s)
        assert (lus >= 0).all()
        for wave in range(4):
            for lane in range(64):
                for ut in range(2):
                    for kk in range(0, 16, 2):
                        row = (wave & 1) * 64 + ut * 32 + (lane & 31)
                        assert lus[a_slab_address(wave, lane, ut, kk)] == U2[row, 16 * s + kk + (lane >> 5)]
    # worst bank multiplicity of one operand fetch (64 banks of 4 bytes)
    worst_a = max(max(Counter(a_address(0, l, 0, s, kk) % 64 for l in range(64)).values())
                  for s in range(4) for kk in range(0, 16, 2))
    worst_b = max(max(Counter(b_address(0, l, 0, kk) % 64 for l in range(64)).values())
                  for kk in range(0, 16, 2))
    return worst_a, worst_b


if __name__ == "__main__":
    print("placement and reads agree; worst bank load of an A / B fetch: %d / %d" % check())
