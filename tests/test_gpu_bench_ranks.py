"""bench.py's N > 1 branch, executed: `bench.py --gpus 2 --share-device ...` starts its own two ranks
(python -m torch.distributed.run), both on this box's one GPU with the gloo backend, shards nothing across them (weak
scaling: every rank decodes its own reads), merges the counter buffers with one SUM all-reduce at the end of the run
(reference transcode.cpp:1816-1818 -> selector.cpp:68-77) and rank 0 prints the contract line.  The child process is
started by tests/conftest.py before the pytest process touches the GPU.
"""
import json

import pytest

pytestmark = pytest.mark.gpu


def test_bench_with_two_ranks_prints_a_whole_job_line(bench_ranks_run):
    log = bench_ranks_run["log"]
    assert bench_ranks_run["returncode"] == 0, "bench.py %s failed:\n%s" % (" ".join(bench_ranks_run["command"]), log)
    lines = [text for text in bench_ranks_run["stdout"].splitlines() if text.startswith('{"metric"')]
    assert len(lines) == 1, log                      # rank 0 alone prints
    assert bench_ranks_run["stdout"].rstrip().splitlines()[-1] == lines[0]      # and the contract line is the LAST line of stdout
    assert len(lines[0]) < 4096, len(lines[0])       # a line the driver can keep whole (20 KB did not parse)
    line = json.loads(lines[0])
    assert line["ranks"] == 2 and line["backend"] == "gloo"      # what the process group itself reports
    reads, steps = 2000000, 3
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["merge"] == "once per run"
    assert line["counted_reads"] == 2 * reads * steps           # the merged counters hold every read of every rank and batch
    assert line["ms_per_step"] > 0
    assert abs(line["value"] - 2 * reads / (line["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * line["value"]      # whole job throughput
    assert line["config"]["ranks_share_device"] is True
    roofline = line["roofline"]                      # per rank: the dominant kernel of rank 0 over its own reads
    assert roofline["bound"] == "hbm" and 0 < roofline["frac"] < 1 and roofline["kernel_ms"] > 0
    # a scaling record can be read from the line alone: both ranks' clocks (the value is computed from the slowest one) and the
    # all-reduce of the counters on its own
    per_rank = line["per_rank"]
    assert 0 < per_rank["ms_per_step"][0] <= per_rank["ms_per_step"][1]
    assert abs(per_rank["ms_per_step"][1] - line["ms_per_step"]) < 1e-3 * line["ms_per_step"]
    assert 0 < per_rank["classify_ms_per_step"][0] <= per_rank["classify_ms_per_step"][1] and per_rank["merge_ms"][1] > 0
    assert line["merge_ms"] > 0
