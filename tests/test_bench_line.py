"""CPU: the ONE line bench.py prints for the driver stays compact and parseable.

BENCH_r05's line had grown to 20 KB and the driver could not parse it; the line is now `bench.compact_record(full)`
(numbers only) and the prose lives in bench_full.json.  This test runs the formatter on a canned full record
(tests/golden/bench_full_sample.json = round 5's 20 KB line) and on degenerate ones."""
import json
import os

import bench
from _util import ROOT

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
HARD_LIMIT = 8192


def _sample():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_sample.json")))


def _line(full):
    return json.dumps(bench.compact_record(full), separators=(",", ":"))


def test_compact_line_is_small_and_round_trips():
    full = _sample()
    assert len(json.dumps(full)) > 15000                                # the canned record is the one that broke the driver
    line = _line(full)
    assert len(line) <= bench.COMPACT_LIMIT_BYTES < HARD_LIMIT, len(line)
    assert "\n" not in line
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    # the contract's numbers are carried over untouched
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert d[k] == full[k], k
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["frac"] - full["roofline"]["frac"]) < 1e-4
    assert r["algorithmic_bytes_per_launch"] == 4096 * full["config"]["blocks_per_gpu"]
    assert r["traffic"] is not None and r["traffic_measured_live"] is True and r["kernel_ms_per_launch"] <= d["ms_per_step"] * 1.0001
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] == 16 and c["unit"] == "blocks/s" and c["gpu_output_bit_exact_vs_cpu"] is True
    assert "BASELINE configs[1]" in d["config"]["workload"] and str(full["config"]["blocks_per_gpu"]) in d["config"]["workload"]


def test_compact_also_is_scalars_only():
    d = json.loads(_line(_sample()))
    assert d["checks_failed"] == []
    assert len(d["also"]) >= 30
    for k, v in d["also"].items():
        assert isinstance(v, (int, float)) and not isinstance(v, bool), (k, v)
    for leg in ("dct32_inv", "dct32_fwd_inv_fused", "satd8x8", "satd8x8_me_search", "per_ctu_one_launch", "stream8k",
                "intra32.predict", "dct32_scatter_gather"):
        assert leg in d["also"], leg
    assert 0.5 < d["also"]["dct32_fwd_inv_fused"] < 1.0                 # a roofline fraction where the leg has one
    assert d["secondary"]["metric"] == "satd8x8_blocks_per_s"
    for v in d.values():                                                # no prose anywhere: every string is short
        if isinstance(v, str):
            assert len(v) < 120


def test_compact_reports_a_failed_check_and_errors():
    full = _sample()
    full["also"]["stream8k"]["bit_exact_vs_single_device"] = False
    full["also"]["node_layer_error"] = "x" * 5000
    full["error"] = "GPU output differs from the CPU reference"
    d = json.loads(_line(full))
    assert d["checks_failed"] == ["stream8k.bit_exact_vs_single_device"]
    assert len(d["node_layer_error"]) == 200 and d["error"]
    assert len(_line(full)) < HARD_LIMIT


def test_compact_survives_a_headline_only_and_a_multi_rank_record():
    full = _sample()
    for k in ("also", "secondary"):
        del full[k]
    full["cpu_baseline"] = None
    full["n_gpus"] = 8
    full["roofline"]["frac_by_rank"] = [0.83] * 8
    d = json.loads(_line(full))
    assert d["cpu_baseline"] is None and "also" not in d and len(d["roofline"]["frac_by_rank"]) == 8
    assert len(_line(full)) < 2048


def test_rank_to_device_mapping_for_eight_ranks():
    """torchrun's LOCAL_RANK i -> HIP ordinal i on an 8-GPU node; a launcher that narrows each rank to one visible device -> ordinal 0;
    fewer devices than ranks without such narrowing is refused with a message (not a failed hipSetDevice deep in the library)."""
    import pytest
    assert [bench.pick_device(r, 8, False, {}) for r in range(8)] == list(range(8))
    assert [bench.pick_device(r, 1, False, {"HIP_VISIBLE_DEVICES": str(r)}) for r in range(8)] == [0] * 8
    assert [bench.pick_device(r, 1, False, {"ROCR_VISIBLE_DEVICES": str(r)}) for r in range(8)] == [0] * 8
    assert [bench.pick_device(r, 2, True, {}) for r in range(8)] == [0, 1] * 4         # the one-GPU test hook
    with pytest.raises(SystemExit, match="LOCAL_RANK 5 but only 4"):
        bench.pick_device(5, 4, False, {})
    with pytest.raises(SystemExit):
        bench.pick_device(1, 1, False, {})


def test_non_finite_numbers_never_reach_the_line():
    """a leg that divides by zero (inf) or by nothing (nan) becomes null in the line: NaN / Infinity are not JSON and would cost the driver the whole record"""
    full = _sample()
    full["also"]["dct32_inv"]["roofline"]["frac"] = float("nan")
    full["also"]["stream8k"]["frames_per_s"] = float("inf")
    full["roofline"]["frac_at_mean"] = float("-inf")
    line = json.dumps(bench._finite(bench.compact_record(full)), separators=(",", ":"), allow_nan=False)
    d = json.loads(line)
    assert d["also"]["dct32_inv"] is None and d["also"]["stream8k"] is None and d["roofline"]["frac_at_mean"] is None
    assert d["value"] == full["value"] and "NaN" not in line and "Infinity" not in line


def test_a_leg_that_threw_is_named_in_the_line():
    full = _sample()
    del full["also"]["intra32"]
    full["also"]["intra32_error"] = "X266Error: xIntra32PredictDev failed (-3): intra launch: out of memory"
    d = json.loads(_line(full))
    assert d["checks_failed"] == ["intra32_error"] and "intra32.predict" not in d["also"] and d["also"]["dct32_inv"] > 0
