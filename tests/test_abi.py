"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/ansx.h declares; pure host functions behave; no compute call is made (no GPU here)."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def A():
    import ans_large_alphabet_amd as A_

    if not os.path.exists(os.path.join(ROOT, "ans_large_alphabet_amd", "libansx.so")):
        A_.build_library()
    return A_


def test_every_declared_symbol_is_exported(A):
    hdr = open(os.path.join(ROOT, "include", "ansx.h")).read()
    declared = sorted(set(re.findall(r"\b(ansx_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 16
    lib = C.CDLL(os.path.join(ROOT, "ans_large_alphabet_amd", "libansx.so"))
    for name in declared:
        assert hasattr(lib, name), name
    from ans_large_alphabet_amd import _lib

    assert sorted(_lib.EXPORTS) == declared


def test_names_follow_methods_hpp(A):
    # /root/reference/include/methods.hpp:530-533,550-553
    assert A.ANSfold(1).name() == "ANSfold-1"
    assert A.ANSfold(5).name() == "ANSfold-5"
    assert A.ANSrfold(3).name() == "ANSrfold-3"
    assert A.ANSmsb().name() == "ANSmsb"  # methods.hpp:500


def test_bound_and_argument_validation(A):
    L = A.lib()
    for kind in (A.FOLD, A.RFOLD):
        for f in (1, 3, 5, 6, 7):
            b = L.ansx_bound(kind, f, 1 << 20, None)
            assert b >= 7 * (1 << 20)
    assert L.ansx_bound(A.FOLD, 0, 100, None) == 0      # fidelity out of range
    assert L.ansx_bound(A.FOLD, 8, 100, None) == 0      # above ANSX_MAX_FIDELITY (unsound in the reference itself, include/ansx.h)
    assert L.ansx_bound(A.RFOLD, 4, 100, None) > 0
    assert L.ansx_bound(3, 1, 100, None) == 0           # unknown codec
    assert L.ansx_bound(A.MSB, 1, 100, None) == 0       # ANSmsb takes no fidelity
    assert L.ansx_bound(A.MSB, 0, 100, None) > 700
    assert L.ansx_bound(A.FOLD, 1, 0, None) == 0        # n == 0 (reference never terminates)
    o = A.make_opts(block_ints=1001)                    # blocks must be a multiple of 4 ints
    assert L.ansx_bound(A.FOLD, 1, 100, C.byref(o)) == 0
    o = A.make_opts(block_ints=A.SINGLE_STREAM)
    assert L.ansx_bound(A.FOLD, 1, 100, C.byref(o)) >= 700
    for st in range(8):
        assert L.ansx_strerror(st)


def test_container_info_rejects_garbage(A):
    from ans_large_alphabet_amd import _lib

    H = _lib.ContainerHeader()
    buf = np.zeros(128, dtype=np.uint8)
    assert A.lib().ansx_container_info(buf.ctypes.data, buf.size, C.byref(H)) == _lib.ERR_FORMAT
    assert A.lib().ansx_container_info(buf.ctypes.data, 10, C.byref(H)) == _lib.ERR_FORMAT


def test_container_v3_host_side(A, oracle_built):
    """Container v3 without a GPU: the Python builder's container is accepted by ansx_container_info in both
    restart-point formats, parse_container unpacks the 29-byte records to the states and cursors that went in
    (4 x 52-bit states, 24-bit cursor; extremes included), v2 is refused."""
    import container_py as cp
    import oracle_lib as ol

    rng = np.random.default_rng(3)
    st = rng.integers(0, 1 << 52, size=(37, 4), dtype=np.uint64)
    st[0] = (1 << 52) - 1
    st[1] = 0
    off = rng.integers(0, 1 << 24, size=37, dtype=np.uint32)
    off[0], off[1] = (1 << 24) - 1, 0
    raw = cp.pack_restart_points(st, off)
    assert raw.size == 37 * 29
    o2, s2 = A.codec.unpack_restart_points(raw)
    assert np.array_equal(o2, off) and np.array_equal(s2.reshape(-1, 4), st)
    data = ol.gen_inputs("zipf20s1.2", 20001, seed=2)
    for wide in (False, True):
        c = cp.build_container(ol.FOLD, 1, data, 4096, 1024, wide=wide)
        parts = A.parse_container(c)
        assert bool(parts["header"].kind & 0x200) == wide and parts["header"].nblocks == 5
        for b in range(5):
            s, info, est, eoff = ol.oracle_encode(ol.FOLD, 1, data[b * 4096:(b + 1) * 4096], ckpt_interval=1024)
            k = est.shape[0]
            assert np.array_equal(parts["streams"][b], s)
            assert np.array_equal(parts["ckpt_state"][b][:k], est) and np.array_equal(parts["ckpt_off"][b][:k], eoff)
    v2 = cp.build_container(ol.FOLD, 1, data, 4096, 1024).copy()
    v2[5] = ord("2")
    with pytest.raises(A.AnsxError):
        A.parse_container(v2)


def test_init_without_device_fails_loudly(A):
    """No GPU in the CPU container: the product must refuse, not fall back."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(A.AnsxError) as ei:
        A.Context()
    assert ei.value.status == 5  # ANSX_ERR_NO_DEVICE


def test_portable_log2_close_to_libm(A):
    """The device normaliser's log2 (same code on host) is within 1 ulp of libm and exact on
    powers of two (normalisation)."""
    L = A.lib()
    rng = np.random.default_rng(0)
    worst = 0.0
    xs = np.concatenate([rng.random(20000), rng.integers(1, 65535, 20000) / 65536.0,
                         rng.integers(1, 1 << 14, 20000) / float(1 << 14)])
    for x in xs:
        if x <= 0:
            continue
        a, b = L.ansx_host_log2(float(x)), math.log2(float(x))
        if b != 0:
            worst = max(worst, abs(a - b) / abs(b))
    assert worst <= 2.3e-16
    for k in range(-40, 20):
        assert L.ansx_host_log2(2.0 ** k) == float(k)


def test_product_does_not_import_oracle():
    """The shipped package must not reference the test oracle."""
    pkg = os.path.join(ROOT, "ans_large_alphabet_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle_lib" not in txt and "ans_oracle" not in txt and "libans_ref" not in txt, fn
