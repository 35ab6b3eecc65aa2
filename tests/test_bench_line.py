"""The contract line of bench.py stays a line the driver can keep (a 20 KB line left `parsed: null`).

The canned record is the full line a plain `python bench.py` printed at the end of round 4
(profiles/r04_bench_default_all_legs.json: every leg with its chain, parity, notes); `contract_line` must turn it —
and the same record with every optional leg there is, two ranks and a small batch block — into one JSON line under
the budget that still carries the contract's fields, `roofline` and `cpu_baseline`."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default_all_legs.json")))


def check(text, full):
    assert "\n" not in text
    assert len(text) < bench.LINE_BUDGET, len(text)
    line = json.loads(text)
    for key in CONTRACT:
        assert key in line, key
        if key != "config":
            assert line[key] == full[key], key
    assert line["config"]["workload"] == full["config"]["workload"]
    assert "model" not in line["config"]
    roofline = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roofline, key
    assert roofline["bound"] == "hbm" and roofline["unit"] == "GB/s"
    assert abs(roofline["frac"] - roofline["achieved"] / roofline["peak"]) < 1e-4
    return line


def test_the_round_4_record_becomes_a_line_under_the_budget():
    full = canned()
    assert len(json.dumps(full)) > 4 * bench.LINE_BUDGET          # the record that did not parse
    line = check(bench.contract_line(full), full)
    baseline = line["cpu_baseline"]
    assert baseline["kind"] == "port" and baseline["cores"] == full["cpu_baseline"]["cores"] and baseline["unit"] == "Mreads/s"
    assert line["parity"]["reads"] == full["parity"]["reads"]
    assert set(line["shapes"]) == set(full["shapes"])
    for name, (rate, fraction) in line["shapes"].items():
        assert abs(rate * 1e3 - full["shapes"][name]["Mreads/s"]) <= 1e-3 * full["shapes"][name]["Mreads/s"]
        assert abs(fraction - full["shapes"][name]["step_frac"]) <= 1e-2 * full["shapes"][name]["step_frac"]
    assert line["literal"]["step_frac"] > 0
    assert set(line["staged"]) == {"pageable", "pinned", "packed", "packed_records_only"}
    assert "report_only" in line["fastq_to_sam"]


def test_every_optional_leg_at_once_still_fits():
    full = canned()
    for k in range(24):                                              # twice the shapes there are today, long names
        full["shapes"]["X%02d_pamld_estimated_priors_three_segments" % k] = copy.deepcopy(full["shapes"]["C3_pamld"])
    full["small_batches"] = {"C%d_%s" % (c, n): {"Mreads/s": 12345.678} for c in (2, 3, 4, 5) for n in ("64K", "1M", "4M", "16M")}
    full["fastq_to_sam"]["report_only_bgzf"] = {"Mreads/s": 41.2}
    full["fastq_to_sam"]["report_only_gz"] = {"Mreads/s": 5.2}
    full.update(n_gpus=8, ranks=8, backend="nccl")
    full["legs"] = "bench_legs.json"
    line = check(bench.contract_line(full), full)
    assert line["ranks"] == 8 and line["backend"] == "nccl"
    assert "cpu_baseline" in line and "roofline" in line             # what goes first when the budget is short is never these


def test_a_multi_rank_line_without_the_single_rank_legs():
    full = canned()
    for key in ("cpu_baseline", "parity", "parity_checked_reads", "staged", "fastq_to_sam", "literal", "literal_Mreads/s", "shapes"):
        full.pop(key, None)
    full.update(n_gpus=2, ranks=2, backend="gloo")
    line = check(bench.contract_line(full), full)
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo"
    assert "cpu_baseline" not in line


def test_emit_prints_the_line_last_and_keeps_every_leg_beside_the_script(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = canned()
    bench.emit(full)
    captured = capsys.readouterr()
    lines = captured.out.strip().split("\n")
    assert len(lines) == 1
    line = check(lines[0], full)
    assert line["legs"] == "bench_legs.json"
    side = json.load(open(tmp_path / "bench_legs.json"))
    assert side["shapes"]["C3_pamld"]["chain"] == full["shapes"]["C3_pamld"]["chain"]      # nothing is lost: it moved
    assert json.loads(captured.err.strip().split("\n")[-1])["shapes"].keys() == full["shapes"].keys()
