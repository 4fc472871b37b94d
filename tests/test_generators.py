"""Synthetic-input generators (include/ansx.h ansx_generate_host / _dev): the reference's distributions
(src/generate_inputs.cpp:94-122, include/zipf_dist.hpp) as counter-based functions of (seed, index)."""
import json
import os

import numpy as np
import pytest

import ans_large_alphabet_amd as A

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_host_generators_follow_their_distributions():
    n = 2_000_000
    u = A.generate_host("uniform1-256", n, seed=5)
    assert u.min() == 1 and u.max() == 256
    c = np.bincount(u, minlength=257)[1:]
    assert abs(c / n - 1 / 256).max() < 4e-4
    assert A.generate_host("uniform12", 1000, seed=1).max() < 4096          # generate_inputs.cpp:96
    for p in (0.01, 0.4, 0.9):                                               # generate_inputs.cpp:103-118
        g = A.generate_host("geom%g" % p, n, seed=7)
        k = np.arange(6)
        assert abs(np.bincount(g, minlength=6)[:6] / n - p * (1 - p) ** k).max() < 2e-3, p
        assert abs(g.mean() - (1 - p) / p) < 0.02 * (1 - p) / p + 1e-3
    for lg, q in ((12, 1.0), (20, 1.0), (20, 1.2), (24, 1.2)):               # zipf_dist.hpp:49-59
        z = A.generate_host("zipf%ds%g" % (lg, q), n, seed=11)
        assert z.min() >= 1 and z.max() <= (1 << lg)
        w = 1.0 / np.arange(1, (1 << lg) + 1, dtype=np.float64) ** q
        w /= w.sum()
        got = np.bincount(z, minlength=65)[1:65] / n
        assert abs(got - w[:64]).max() < 1.5e-3, (lg, q)
        tail = (z > 1000).mean()
        assert abs(tail - w[1000:].sum()) < 2e-3, (lg, q)


def _check_zipf_map(n, q, values, draws, u01):
    """Every uniform the reference's rejection loop consumed must take the same turn here: rejected draws rejected,
    the accepted one accepted with the reference's value."""
    pos = 0
    for v, d in zip(values, draws):
        for j in range(d):
            k, acc = A.zipf_from_uniform(n, q, u01[pos])
            pos += 1
            last = j == d - 1
            assert acc == last, (n, q, v, j, d)
            if last:
                assert k == v, (n, q, v, k)
    assert pos == len(u01)


def test_zipf_map_equals_the_reference_class_on_its_own_uniforms():
    """include/zipf_dist.hpp:49-59 (compiled into oracle/_ref, driven by std::mt19937 as generate_inputs.cpp does)
    recorded the canonical uniforms of 10 500 values (tests/golden/zipf_trace.json, make_zipf_golden.py); the
    package's map uniform -> (candidate, accept) must reproduce every one of its decisions and values.  (The
    random STREAM is this build's own -- counter-based, so that a list can be drawn in shards on any number of GPUs;
    the distribution code is pinned here.)"""
    with open(os.path.join(GOLD, "zipf_trace.json")) as fh:
        doc = json.load(fh)
    total = 0
    for c in doc["cases"]:
        u = [float.fromhex(x) for x in c["u01"]]
        _check_zipf_map(c["n"], c["q"], c["values"], c["draws"], u)
        total += len(c["values"])
    assert total >= 10000


def test_zipf_map_equals_the_compiled_reference_live(oracle_built):
    """The same comparison on fresh draws when oracle/_ref is present (authoring container and GPU box)."""
    import oracle_lib as ol

    if not ol.have_ref() or not hasattr(ol.ref(), "ref_zipf_trace"):
        pytest.skip("oracle/_ref not built")
    for n, q, seed in ((1 << 20, 1.2, 77), (1 << 24, 1.2, 78), (1 << 16, 1.0, 79)):
        vals, nd, u = ol.ref_zipf_trace(n, q, seed, 20000)
        _check_zipf_map(n, q, [int(x) for x in vals], [int(x) for x in nd], [float(x) for x in u])


def test_generators_are_pure_functions_of_seed_and_index():
    a = A.generate_host("zipf20s1.2", 10000, seed=42)
    b = np.concatenate([A.generate_host("zipf20s1.2", 3000, seed=42),
                        A.generate_host("zipf20s1.2", 7000, seed=42, first_index=3000)])
    assert np.array_equal(a, b)                      # pieces of one list (multi-GPU shards) line up
    assert not np.array_equal(a, A.generate_host("zipf20s1.2", 10000, seed=43))
    with pytest.raises(A.AnsxError):
        A.generate_host("geom1.5", 10)
    with pytest.raises(A.AnsxError):
        A.generate_host("uniform9-3", 10)


@pytest.mark.gpu
def test_generators_device_equals_host():
    import torch

    ctx = A.Context(0)
    n = 300_007
    for spec in ("uniform1-256", "uniform20", "geom0.01", "geom0.6", "zipf12", "zipf20s1.2", "zipf24s1.0", "zipf24s1.2"):
        d = torch.empty(n, dtype=torch.int32, device="cuda")
        A.generate_dev(ctx, spec, d.data_ptr(), n, seed=99, first_index=12345)
        torch.cuda.synchronize()
        host = A.generate_host(spec, n, seed=99, first_index=12345)
        assert np.array_equal(d.cpu().numpy().view(np.uint32), host), spec
    ctx.close()
