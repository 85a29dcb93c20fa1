"""LDS bank-conflict check of the MFMA fragment reads, on the CPU (no GPU, no simulator).

MI355X_MICROARCH.md (LDS table): a wave64 `ds_read_b128` is served in four 16-lane groups -- lanes {0-3, 12-15, 20-27},
{4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63} -- over 64 four-byte banks; lanes of one group that touch
the same bank at different addresses serialise.  Every GEMM-shaped kernel of the engine reads its operand fragments as "lane
(r = l & 15, g = l >> 4) takes 16 bytes at word 4g (+16 for the second half) of LDS row r", so the row stride decides whether
that is conflict free.  Round 1 padded rows by 4 words (36 / 132 / 260) -- 2-way conflicts on every read, measured as -10 % on
the fused feed-forward block and -3 % end to end (profiles/r02_lds_stride_ab.json); the strides in the sources are now = 8 mod
16 words.  This test reads the stride constants from the kernel sources and brute-forces the access pattern, so that a later
edit of a stride (or of the fragment addressing) that reintroduces conflicts fails here rather than in a profile."""
import os
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "motion-latent-diffusion_amd", "csrc", "kernels")
GROUPS_B128 = [[*range(0, 4), *range(12, 16), *range(20, 28)], [*range(4, 12), *range(16, 20), *range(28, 32)],
               [*range(32, 36), *range(44, 48), *range(52, 60)], [*range(36, 44), *range(48, 52), *range(60, 64)]]


def lds_cycles_b128(stride_words, word_of_g, base=0):
    """LDS-array cycles of one wave64 ds_read_b128 where lane (r, g) reads words base + r * stride + word_of_g(g) .. + 3 (4 = conflict free)."""
    total = 0
    for grp in GROUPS_B128:
        banks = {}
        for lane in grp:
            g, r = lane >> 4, lane & 15
            a = base + r * stride_words + word_of_g(g)
            for w in range(4):
                banks.setdefault((a + w) % 64, set()).add(a + w)
        total += max(len(v) for v in banks.values())
    return total


def constant(fname, name):
    src = open(os.path.join(CSRC, fname)).read()
    m = re.search(r"constexpr\s+int\s+(?:\w+\s*=\s*\d+\s*,\s*)*" + name + r"\s*=\s*(\d+)\s*[;,]", src)
    assert m, (fname, name)
    return int(m.group(1))


def test_the_check_sees_the_round1_conflicts():
    # the strides round 1 used: two rows of a lane group share banks -> 8 cycles instead of 4
    for stride in (36, 132, 260):
        assert lds_cycles_b128(stride, lambda g: 4 * g) == 8
    # the former fp32 fragment (words 8g and 8g + 4) is at least 2-way at ANY 16-byte-aligned stride
    assert all(lds_cycles_b128(s, lambda g: 8 * g) >= 8 for s in range(32, 300, 4))


@pytest.mark.parametrize("fname,name", [("gemm.hpp", "kGemmLdsStride"), ("strip.hpp", "kStripWStride"), ("tile32.hpp", "kT32Stride"),
                                        ("attention.hpp", "kAttnX3KStride"), ("ffn_strip.hpp", "kFsXs"), ("ffn_strip.hpp", "kFsHs"),
                                        ("loop_fused.hpp", "kLfXs"), ("loop_fused.hpp", "kLfHs")])
def test_fragment_reads_are_conflict_free(fname, name):
    stride = constant(fname, name)
    assert stride % 4 == 0, "rows must stay 16-byte aligned"
    for base in (0, 16):                       # first / second 16-byte piece of a fragment (bf16 high | low halves, fp32 k-slots 4g | 16 + 4g)
        assert lds_cycles_b128(stride, lambda g: 4 * g, base) == 4, (name, stride, base)


def test_strip_a_strip_strides_are_conflict_free():
    # strip.hpp: the A strip holds K = 256 or 512 floats per row, stride K + 8 (read from the source: "ST = K + 8")
    src = open(os.path.join(CSRC, "strip.hpp")).read()
    m = re.search(r"ST\s*=\s*K\s*\+\s*(\d+)", src)
    assert m
    pad = int(m.group(1))
    for K in (256, 512):
        for kc in range(K // 32):
            assert lds_cycles_b128(K + pad, lambda g: 4 * g, kc * 32) == 4
            assert lds_cycles_b128(K + pad, lambda g: 4 * g, kc * 32 + 16) == 4


def test_sources_read_fp32_fragments_at_the_conflict_free_slots():
    # the fp32 fragment addressing itself: words 4g and 16 + 4g of the chunk in all three fp32 kernels
    strip = open(os.path.join(CSRC, "strip.hpp")).read()
    assert "kc * 32 + g * 4)" in strip and "kc * 32 + 16 + g * 4)" in strip and "g * 8" not in strip.split("strip_mma")[1].split("PREC_BF16")[0]
    t32 = open(os.path.join(CSRC, "tile32.hpp")).read()
    assert "kc * 32 + g * 4)" in t32 and "kc * 32 + 16 + g * 4)" in t32
    gemm = open(os.path.join(CSRC, "gemm.hpp")).read()
    lf = gemm.split("auto lfrags = [&](int buf)")[1].split("};")[0]
    assert "+ g * 4;" in lf and "+ 16)" in lf and "g * 8" not in lf


def lds_cycles_write_b64(stride_words, word_of_lane):
    """LDS-array cycles of one wave64 ds_write_b64 (MI355X_MICROARCH.md: served per 16 CONTIGUOUS lanes, 32 banks): lane (r, g) stores
    two words at r * stride + word_of_lane(r, g).  4 = conflict free (one cycle per lane group)."""
    total = 0
    for grp in range(4):
        banks = {}
        for lane in range(grp * 16, grp * 16 + 16):
            g, r = lane >> 4, lane & 15
            a = r * stride_words + word_of_lane(r, g)
            for w in range(2):
                banks.setdefault((a + w) % 32, set()).add(a + w)
        total += max(len(v) for v in banks.values())
    return total


def lds_cycles_b128_lane(stride_words, word_of_lane, base=0):
    """as lds_cycles_b128 with a lane-dependent word offset inside the row (swizzled images)"""
    total = 0
    for grp in GROUPS_B128:
        banks = {}
        for lane in grp:
            g, r = lane >> 4, lane & 15
            a = base + r * stride_words + word_of_lane(r, g)
            for w in range(4):
                banks.setdefault((a + w) % 64, set()).add(a + w)
        total += max(len(v) for v in banks.values())
    return total


def test_row_swizzled_operand_images_of_the_persistent_loop():
    """loop_fused.hpp SWZ ("fused_swz"): lane (r, g) of wave w stores its four columns of row 16 t + r as 8 bytes at word
    32 (w >> 1) + 8 (w & 1) + 2 g of the chunk's plane.  Sixteen rows 264 (136) = 8 mod 32 words apart: every bank pair is hit four
    times per 16-lane group; with the word offset XORed by 4 (r >> 2), twice -- and the fragment reads (group g ^ (r >> 2) of the half
    chunk) stay conflict free for every chunk and both halves."""
    src = open(os.path.join(CSRC, "loop_fused.hpp")).read()
    assert "^ swz4" in src and "((g ^ (r >> 2)) << 2)" in src and "((r >> 2) << 2)" in src
    for name in ("kLfXs", "kLfHs"):
        stride = constant("loop_fused.hpp", name)
        for wave in range(8):
            plain = lambda r, g: (wave >> 1) * 32 + (wave & 1) * 8 + 2 * g
            swz = lambda r, g: plain(r, g) ^ ((r >> 2) << 2)
            for plane in (0, 16):
                assert lds_cycles_write_b64(stride, lambda r, g: plain(r, g) + plane) == 16      # 4-way
                assert lds_cycles_write_b64(stride, lambda r, g: swz(r, g) + plane) == 8         # 2-way
        # (no 8-byte store of one column group can do better at a stride = 8 mod 16 words: 16 rows reach only 8 distinct bank pairs)
        for c in range(8 if name == "kLfXs" else 4):
            for half in (0, 16):
                assert lds_cycles_b128_lane(stride, lambda r, g: (g ^ (r >> 2)) << 2, c * 32 + half) == 4


def test_pipelined_gemm_tile_map_and_lds_rows():
    """gemm_pipe.hpp: (a) the XCD-aware 1-D grid -- id lands on XCD id % 8; XCD x walks row tiles x, x + 8, ... and inside a row tile all
    column tiles back to back -- visits every (row tile, column tile) exactly once, the padded workgroups fall outside, and the column
    tiles of a row panel are consecutive dispatches of ONE XCD; (b) its fragment reads (ds_read_b128 at word 4 g of row r, 40-word rows,
    high plane at + 0, low plane at + 16) are conflict free, as are the head-dim-128 attention's (72-word rows, four chunks)."""
    src = open(os.path.join(CSRC, "gemm_pipe.hpp")).read()
    assert "blockIdx.x & 7" in src and "(kk / nt) * 8 + xcd" in src and "kk % nt" in src
    for mt, nt in ((196, 2), (196, 6), (196, 4), (1, 6), (9, 2), (16, 1)):
        grid = (mt + 7) // 8 * 8 * nt
        seen = {}
        for wg in range(grid):
            xcd, kk = wg & 7, wg >> 3
            tm, tn = (kk // nt) * 8 + xcd, kk % nt
            if tm < mt:
                assert (tm, tn) not in seen
                seen[(tm, tn)] = wg
        assert len(seen) == mt * nt
        for tm in range(mt):
            ids = [seen[(tm, tn)] for tn in range(nt)]
            assert len({i & 7 for i in ids}) == 1 and [i >> 3 for i in ids] == list(range(ids[0] >> 3, (ids[0] >> 3) + nt))
    stride = constant("gemm.hpp", "kGemmLdsStride")
    assert stride % 16 == 8
    for plane in (0, 16):
        assert lds_cycles_b128_lane(stride, lambda r, g: g * 4, plane) == 4
    kst = constant("attention.hpp", "kFlash128KStride")
    assert kst % 16 == 8
    for c in range(4):
        assert lds_cycles_b128_lane(kst, lambda r, g: g * 4, c * 16) == 4
