"""The bench's FINAL stdout line is what the driver parses: BENCH_r05.parsed was null because the line had grown to
23.6 KB.  These tests assemble the line from recorded payloads (round 5's N = 1 line and the 2-rank rehearsal line) and
hold its shape: strict JSON, ASCII, small, every key of the contract, nothing but the compact line last on stdout."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PAYLOADS = ["profiles/r05_bench_n1.json", "profiles/r05_n2_rehearsal_gloo.json", "profiles/r06_bench_detail.json", "profiles/r06_n2_rehearsal_gloo.json"]


def strict(text):
    def no_constants(c):
        raise ValueError(f"non-strict JSON constant {c}")
    return json.loads(text, parse_constant=no_constants)


def load(rel):
    with open(os.path.join(ROOT, rel)) as f:
        lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


@pytest.mark.parametrize("rel", PAYLOADS)
def test_final_line_is_small_strict_ascii_and_complete(rel):
    out = load(rel)
    text = bench.compact_line(out)
    assert len(text.encode()) <= bench.FINAL_LINE_MAX < 8192
    assert text.isascii() and "\n" not in text
    d = strict(text)
    for k in bench.REQUIRED_KEYS:
        assert k in d, k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"):
        assert d[k] == out[k]
    assert isinstance(d["config"]["workload"], str) and len(d["config"]["workload"]) <= 300
    assert d["roofline"]["frac"] == out["roofline"]["frac"] and d["roofline"]["bound"] in ("hbm", "mfma")
    assert "note" not in d["roofline"] and "source" not in (d["roofline"].get("issue_model") or {})
    assert "stage_ms_per_step_rank0" in d
    assert "legs" not in d and "sub_records" not in d
    # nothing in the line is a list, and nothing is nested deeper than roofline.issue_model
    def depth(v):
        assert not isinstance(v, list)
        return 1 + max((depth(x) for x in v.values()), default=0) if isinstance(v, dict) else 0
    assert depth(d) <= 3


def test_cpu_baseline_keeps_its_scalars_not_its_lists():
    d = strict(bench.compact_line(load(PAYLOADS[0])))
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "parity_pose_max_abs_diff", "parity_inliers_equal", "parity_ok_equal",
              "parity_rmse_max_abs_diff"):
        assert k in cb, k
    assert all(not isinstance(v, (list, dict)) for v in cb.values())


def test_the_n_gt_1_line_keeps_its_scaling_scalars():
    d = strict(bench.compact_line(load(PAYLOADS[1])))
    for k in ("per_gpu_value", "rccl_ranks_seen", "knn_cfgE_q64_us", "knn_cfgE_q1_us"):
        assert k in d, k
    assert "collectives" in d["config"]


def test_nan_and_infinity_never_reach_the_line_and_oversize_groups_are_dropped():
    out = load(PAYLOADS[0])
    out["roofline"]["traffic"] = float("nan")
    out["lone_query_ms"] = float("inf")
    out["config"]["workload"] = "x" * 5000
    out["accuracy"] = {f"k{i}": float(i) for i in range(400)}          # a group that has outgrown the line: dropped, line still fits
    text = bench.compact_line(out)
    d = strict(text)
    assert len(text) <= bench.FINAL_LINE_MAX and d["roofline"]["traffic"] is None and d["lone_query_ms"] is None
    assert len(d["config"]["workload"]) == 300 and "accuracy" not in d
    for k in bench.REQUIRED_KEYS:
        assert k in d


def test_emit_prints_the_detail_first_and_the_compact_line_last(tmp_path):
    out = load(PAYLOADS[0])
    buf = io.StringIO()
    final = bench.emit(out, stream=buf, detail_path=str(tmp_path / "bench_detail.json"))
    lines = buf.getvalue().splitlines()
    assert len(lines) == 2 and lines[0].startswith("[bench-detail] {") and lines[1] == final
    detail = strict(lines[0][len("[bench-detail] "):])
    assert "legs" in detail and "sub_records" in detail
    assert strict(open(tmp_path / "bench_detail.json").read())["value"] == out["value"]
    assert strict(lines[-1])["detail"] == "bench_detail.json"
