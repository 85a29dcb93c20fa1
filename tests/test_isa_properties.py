"""Static checks of the gfx950 machine code (no GPU): tools/isa_report.py disassembles libmldhip.so and this test pins the properties
of the hot kernels that earlier rounds paid for on the profiler -- a ring pointer that keeps its address space (no flat loads), no
scratch where there was none, the V operand through the transpose read, the expected number of matrix instructions."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import isa_report  # noqa: E402


@pytest.fixture(scope="module")
def rep():
    if not os.path.exists(isa_report.DEFAULT_LIB) or not os.path.exists(os.path.join(isa_report.LLVM, "llvm-objdump")):
        pytest.skip("libmldhip.so / the LLVM binary tools are not here")
    return isa_report.report()


def one(rep, part):
    hits = [v for k, v in rep.items() if part in k]
    assert len(hits) == 1, (part, [k for k in rep if part in k])
    return hits[0]


def test_persistent_loop_code(rep):
    k = one(rep, "den_loop_kernel<true, 0>")
    assert k["flat"] == 0                      # DESIGN.md point 34: an opaque POINTER turned the ring into flat loads (+8 %)
    assert k["mfma"] == 1224 and k["ds_write_b16"] == 0
    # round 4: the skip linears' accumulators are pinned behind their matrix instructions (loop_fused.hpp pin_acc) -- the A fragments / ring
    # slots of five chunks are no longer live at once: no scratch at all (rounds 3-4: 176-216 B per lane at the 256-register cap), 8-deep ring
    assert k["scratch"] == 0 and k["vgpr"] <= 232
    assert one(rep, "den_loop_kernel<false, 0>")["scratch"] == 0
    assert k["ds_bpermute"] <= 8               # round 4: row statistics on v_permlane16/32_swap (the 8 left are the once-per-step CFG row exchange, xor 8)
    assert one(rep, "den_loop_kernel<false, 0>")["flat"] == 0
    assert not [n for n in rep if "den_loop_kernel" in n and n.split("<")[1].split(">")[0].split(",")[1].strip() in ("1", "2", "3", "4")]   # measurement builds live in tools/loopbench


def test_pipelined_gemm_tile_code(rep):
    """kernels/gemm_pipe.hpp: 16 / 32 K chunks x 48 matrix instructions per wave, no scratch, and -- what the placement pins are for --
    the global prefetch ring is never drained inside the chunk loop (a `s_waitcnt vmcnt(0)` there = the split of a staged row floated up
    to its load)."""
    for kcs in (16, 32):
        k = one(rep, f"gemm_pipe_x3_kernel<2, 4, 4, 4, {kcs}, 2>")
        assert k["mfma"] == kcs * 48 and k["scratch"] == 0 and k["flat"] == 0 and k["vgpr"] <= 240, k
        assert k["vmcnt0"] <= 3, k                 # prologue (two chunks) only
        assert k["barriers"] == kcs + 3, k         # one per chunk, prologue, two around the output tile


def test_key_blocked_attention_reads_v_through_the_transpose_read(rep):
    tr = one(rep, "attn_flash_x3_kernel")
    assert tr["ds_read_tr"] == 16 and tr["ds_write_b16"] == 0 and tr["scratch"] == 0 and tr["vgpr"] <= 128 and tr["mfma"] == 48


def test_row_strip_kernels_have_no_scratch_and_no_flat_accesses(rep):
    for part in ("strip_gemm_x3_kernel<6, 1, false, true, true>", "strip_gemm_x3_kernel<4, 2, false, false, false>",
                 "ffn_strip_x3_kernel<3, true, true>", "final_strip_x3_kernel"):
        k = one(rep, part)
        assert k["scratch"] == 0 and k["flat"] == 0, (part, k)
    assert one(rep, "final_strip_x3_kernel")["mfma"] == 8 * 3 * 3 * 3          # chunks x column blocks x row tiles x split products
    assert not [n for n in rep if "ffn_x3_kernel" in n]                        # round 2's LDS-staged feed-forward kernel is gone


def test_cluster_loop_code(rep):
    """kernels/loop_cluster.hpp: the first GPU build kept the loop-invariant addresses of every phase live (256 registers + 544 B of scratch per lane: 9.9 ms against
    7.9); lane indices are laundered per phase since, and the ring stays at 4 fragments (6 / 8 spill).  Matrix instructions per wave and layer path: Ph1 96 (Q, K) or
    72 (V) + 12 (out-projection partial), linear1 + linear2 48 + 48 (4 column groups) or 24 + 24 (8), skip linear 24 or 12."""
    for wt in ("true", "false"):
        k4, k8 = one(rep, f"den_cluster_kernel<{wt}, 4>"), one(rep, f"den_cluster_kernel<{wt}, 8>")
        for k in (k4, k8):
            assert k["scratch"] == 0 and k["vgpr_spills"] == 0 and k["flat"] == 0 and k["vgpr"] <= 200, k
            assert k["ds_write_b16"] == 0
        assert k4["mfma"] == 96 + 72 + 12 + 48 + 48 + 24 and k8["mfma"] == 96 + 72 + 12 + 24 + 24 + 12
    assert one(rep, "clear_cluster_flags_kernel")["global_store"] == 1
