"""Synthetic-input generators (include/ansx.h ansx_generate_host / _dev): the reference's distributions
(src/generate_inputs.cpp:94-122, include/zipf_dist.hpp) as counter-based functions of (seed, index)."""
import numpy as np
import pytest

import ans_large_alphabet_amd as A


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
