"""The table shape -> kernel chain, read off the stage dispatcher itself without a GPU (phq_plan_chain: launch_stage walks its
predicates for every decoder with its launches switched off), over the shape grid of profiles/tools/shape_grid.py — barcode
length x barcodes x {single index, unique pairs, two and three segment combinatorial sets} x {uniform, estimated priors}.  Every
shape has a tier 0 kernel as the FIRST kernel of its chain or sits on the tool's allow list with its reason; no shape is refused
(barcodes past 32 nt take the general kernels, the reference is length agnostic: barcode.h:147-161); no shape of up to 32 nt has
the literal kernel on every read.  tests/test_gpu_shape_grid.py runs the same shapes on the device and holds what ran to this plan."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))

import shape_grid  # noqa: E402

SHAPES = [shape for k, shape in enumerate(shape_grid.shapes()) if shape[4] == "uniform" or k % 6 == 1]


def test_the_allow_list_names_real_shapes():
    for family, predicate, reason in shape_grid.ALLOW:
        assert any(f == family and shape_grid.allowed(f, L, N, p) == reason for f, L, N, _, p in shape_grid.shapes()), reason


def test_every_grid_shape_plans_a_tier_0_kernel_or_is_on_the_allow_list():
    without_tier0, long_barcodes = [], 0
    for family, L, N, per, priors in SHAPES:
        spec, barcodes = shape_grid.spec_of(family, L, N, per, priors)
        names = shape_grid.plan(spec)                                        # (never refused: PheniqsError would fail the test)
        assert names, (family, L, N, priors)
        nucleotides = sum(len(word) for word in next(iter(spec["sample"]["codec"].values()))["barcode"])
        if nucleotides > 32:
            assert names[0] == "pamld_literal_long_kernel (every read)" and names[1:] in ([], ["count_results_kernel"]), (family, L, N, names)
            long_barcodes += 1
            continue
        assert not any("every read" in name for name in names), (family, L, N, priors, names)
        assert names[-1] == "pamld_literal_kernel" or names[-1] == "count_results_kernel", names      # every chain ends in the reference's own sums
        if not any(names[0].startswith(name) for name in shape_grid.TIER0):
            assert shape_grid.allowed(family, L, N, priors) is not None, "first kernel %s is no tier 0 and the shape is not on the allow list (%s)" % (names[0], (family, L, N, priors, names))
            without_tier0.append((family, L, N, priors))
    assert long_barcodes > 0
    assert len(without_tier0) < len(SHAPES) // 2


def test_a_plan_needs_no_device_and_follows_the_mode():
    from pheniqs_amd import DecoderSet, capi, synthetic
    job, _ = synthetic.config_job("C5", "pamld")
    s = DecoderSet(job, device=capi.DEVICE_NONE)
    fast = s.plan_chain(1 << 20, capi.MODE_FAST)
    assert [c["kernel"] for c in fast if c["decoder"] == 1] == ["pamld_stream_kernel", "pamld_exact_lanes_kernel", "pamld_literal_kernel"]
    assert [c["kernel"] for c in fast if c["decoder"] == 4][0] == "pamld_few_kernel"
    literal = s.plan_chain(1 << 20, capi.MODE_LITERAL)
    assert [c["kernel"] for c in literal if c["decoder"] == 1] == ["pamld_literal_kernel (every read)"]
    assert s.plan_chain(0, capi.MODE_FAST) == []
    with pytest.raises(Exception):                                             # the set still computes nothing without a device
        s.stats_export()
    s.close()
