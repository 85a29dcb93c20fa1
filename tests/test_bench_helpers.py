"""CPU checks of the measurement code in bench.py that needs no GPU: the shape-keyed PMC summary is refused when it was collected on other
code, and the decoder_roofline block is reproducible from the committed rocprofv3 summary of the driver's run."""
import csv
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


@pytest.fixture(scope="module")
def bench():
    import bench as b
    return b


def test_algorithmic_flops_match_the_survey(bench):
    total, den, dec = bench.algorithmic_gflop(64, 196)
    assert abs(den - 5.849) < 2e-3 and abs(dec - 215.2) < 0.1 and abs(total - 507.7) < 0.2      # SURVEY.md App. C / DESIGN.md section 3


def test_pmc_summary_is_keyed_by_shape_and_refused_when_stale(bench):
    pmc = json.load(open(bench.PMC_FILE))
    assert set(pmc["shapes"]) >= {"1", "20", "32"}                  # one bs-64 request (the cluster loop), the driver's call, the chip-filling call
    for shape, ent in pmc["shapes"].items():
        assert ent["requests_per_call"] == int(shape)
        loop = ent["kernels"]["den_cluster" if shape == "1" else "den_loop"]
        assert 1.5e10 < loop["traffic_bytes_per_launch"] < 3.5e10
        assert loop["traffic_bytes_per_launch"] == int(2 * loop["FETCH_SIZE"] * 1024 + loop["WRITE_SIZE"] * 1024)   # gfx950 wide-read correction
        if shape != "1":
            assert 0.3 < ent["sq"]["den_loop"]["mfma_busy_frac"] < 0.9
    good = pmc["shapes"]["20"]["loop_kernel_code_hash"]
    ent, why = bench.pmc_summary(20, good)
    assert ent is not None and "20 requests per call" in why
    ent, why = bench.pmc_summary(7, good)
    assert ent is None and "no entry" in why
    ent, why = bench.pmc_summary(20, "0123456789abcdef")            # other machine code AND (almost surely) other sources
    if pmc["shapes"]["20"]["source_hash"] != bench.source_hash():
        assert ent is None and "stale" in why


def test_decoder_roofline_from_the_committed_rocprof_summary(bench):
    rows = list(csv.DictReader(open(os.path.join(REPO, "profiles", "r05_kernel_stats_bench_child_s20.csv"))))
    stats = {r["Name"]: (float(r["AverageNs"]), int(r["Calls"]), float(r["TotalDurationNs"])) for r in rows}
    pmc = json.load(open(bench.PMC_FILE))["shapes"]["20"]
    d = bench.decoder_roofline(stats, 1280, pmc)
    k = d["kernels"]
    assert set(k) == {"in_projection", "self_attention", "decoder_tail", "skip_linear", "final_norm_linear"}
    assert d["rows_per_launch"] == 1280 * 196
    assert abs(k["in_projection"]["algorithmic_gb_per_launch"] - 1280 * 196 * 4096 / 1e9) < 1e-3
    for name, e in k.items():
        assert 0.05 < e["frac_of_copy_rate"] < 1.0 and 0.05 < e["frac_of_mfma_roof"] < 1.0, (name, e)
        assert e["bound"] in ("hbm", "mfma")
    assert k["decoder_tail"]["bound"] == "mfma" and k["in_projection"]["bound"] == "hbm"
    assert 0.3 < k["decoder_tail"]["mfma_busy_frac"] < 0.8
    line = json.load(open(os.path.join(REPO, "profiles", "r05_bench_s20.json")))
    assert abs(line["decoder_roofline"]["decode_ms_per_call_sum_of_these"] - d["decode_ms_per_call_sum_of_these"]) < 0.05     # the committed line was computed from this summary
    r = line["roofline"]
    # (the committed line read round 4's PMC pass of the same machine code: the two passes agree to 5e-6)
    assert abs(r["traffic"] - pmc["kernels"]["den_loop"]["traffic_bytes_per_launch"]) < 1e-3 * r["traffic"] and 0.3 < r["frac"] < 0.6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and abs(r["achieved"] - r["gflop_per_launch"] / r["avg_us_rocprof_dispatch"] * 1e3) < 0.5      # frac = the rocprofv3 figure
    assert abs(r["achieved_hip_events"] - r["gflop_per_launch"] / r["avg_us_hip_events_loop_only_call"] * 1e3) < 0.5 and abs(r["frac_hip_events"] - r["achieved_hip_events"] / r["peak"]) < 1e-3
