"""The shape grid of profiles/tools/shape_grid.py — barcode length x barcodes x {single index, unique pairs, two and three
segment combinatorial sets} x {uniform, estimated priors} — held to a tier 0 kernel: every shape either has one as the FIRST
kernel of its chain (phq_profile_chain: the names are the dispatcher's own) or sits on the tool's allow list with its reason.
A `*_supports()` predicate boundary that sends a shape to the product form over all barcodes — or, as round 5 found for every
codec past 512 packed words, to the literal kernel for every read — fails here instead of hiding behind thirteen named shapes.
Uniform priors walk the whole grid; estimated priors every third shape (the chains differ by the prior weighted kernels only)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))

import shape_grid  # noqa: E402

pytestmark = pytest.mark.gpu

SHAPES = [shape for k, shape in enumerate(shape_grid.shapes()) if shape[4] == "uniform" or k % 6 == 1]


def test_the_allow_list_names_real_shapes():
    """every entry of the allow list matches at least one shape of the grid"""
    for family, predicate, reason in shape_grid.ALLOW:
        assert any(f == family and shape_grid.allowed(f, L, N, p) == reason for f, L, N, _, p in shape_grid.shapes()), reason


@pytest.mark.parametrize("family,L,N,per,priors", SHAPES, ids=["%s-L%d-N%d-%s" % (f, L, N, p) for f, L, N, _, p in SHAPES])
def test_every_grid_shape_takes_a_tier_0_kernel_or_is_on_the_allow_list(family, L, N, per, priors):
    import torch
    from pheniqs_amd import PheniqsError
    spec, barcodes = shape_grid.spec_of(family, L, N, per, priors)
    reason = shape_grid.allowed(family, L, N, priors)
    try:
        rate, ms, chain, _ = shape_grid.run(torch, spec, 1 << 16, steps=1, warmup=0)
    except PheniqsError as error:
        assert reason is not None, "refused and not on the allow list: %s" % error
        return
    assert chain, "no kernel ran"
    first = chain[0][0]
    names = [name for name, _ in chain]
    long_barcode = sum(len(word) for word in next(iter(spec["sample"]["codec"].values()))["barcode"]) > 32
    # never again, allow list or not — but for barcodes past 32 nt, where the general kernel is what keeps the job from being refused
    assert long_barcode or not any("every read" in name for name in names), "the literal kernel took every read: %s" % names
    if not any(first.startswith(name) for name in shape_grid.TIER0):
        assert reason is not None, "first kernel %s is no tier 0 and the shape is not on the allow list (chain: %s)" % (first, names)
