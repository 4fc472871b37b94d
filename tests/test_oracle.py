"""CPU tests (-m "not gpu"): pin the oracle (oracle/ans_oracle.c) against

* the canonical known answers of SURVEY.md section 8c,
* the committed golden fixtures (tests/golden/*.json, made from the real reference by
  tests/golden/make_golden.py),
* oracle/_ref/libans_ref.so itself when it is present (authoring container / prebuilt on the box),
* properties the reference's own tests state (src/test.cpp:54-64 fold/unfold identity;
  round trip as in table_efficiency.cpp:105-106).
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KIND = {"fold": ol.FOLD, "rfold": ol.RFOLD, "msb": ol.MSB}


@pytest.fixture(scope="module", autouse=True)
def _build(oracle_built):
    return oracle_built


def hexs(a):
    return a.tobytes().hex()


# ---------------------------------------------------------------- SURVEY 8c known answers

def test_known_answers_survey_8c():
    s, info, _, _ = ol.oracle_encode(ol.FOLD, 1, [1, 2, 3])
    assert len(s) == 38 and hexs(s) == "0305c8860000aa34" + "00" * 30
    assert info.log2_frame == 5 and info.interp_bits == 17

    s, info, _, _ = ol.oracle_encode(ol.FOLD, 1, [5, 300])
    assert len(s) == 40 and hexs(s) == "80020140abaa012c62" + "00" * 31
    assert info.max_sym == 256 and info.log2_frame == 1 and info.interp_bits == 25

    s, info, _, _ = ol.oracle_encode(ol.FOLD, 1, [1, 2, 3, 70000, 5])
    expect = ("ff0306" "71555555e596122e666666e601000000" "7011" "3565000000000000"
              "a40f000000000000" "970f000000000000" "8a0f000000000000")
    assert len(s) == 53 and hexs(s) == expect
    assert info.max_sym == 511 and info.log2_frame == 6 and info.interp_bits == 99

    r, _, _, _ = ol.oracle_encode(ol.RFOLD, 1, [1, 2, 3, 70000, 5])
    assert hexs(r) == "00000000" + expect

    s, info, _, _ = ol.oracle_encode(ol.FOLD, 3, [9] * 7 + [2])
    assert len(s) == 38 and hexs(s).startswith("0903d75d01001304")
    assert info.log2_frame == 3 and info.interp_bits == 17

    s, info, _, _ = ol.oracle_encode(ol.FOLD, 1, [5])
    assert len(s) == 42 and hexs(s) == "050ffefffefffd7f0000" + "00" * 32
    assert info.log2_frame == 15 and info.interp_bits == 48

    s, info, _, _ = ol.oracle_encode(ol.FOLD, 1, [7] * 1000)
    assert len(s) == 42 and info.log2_frame == 15 and info.interp_bits == 64


def test_interp_codeword_example():
    # nfreq={0,3,1,4}, M=8 -> inc={0,4,6,11}, u=13 -> one word 0x000001c8
    out = np.zeros(64, dtype=np.uint8)
    bits = C.c_uint32(0)
    nf = np.array([0, 3, 1, 4], dtype=np.uint32)
    nb = ol.oracle().ans_oracle_write_prelude(nf, 4, 8, out, C.byref(bits))
    assert nb == 2 + 4 and out[0] == 3 and out[1] == 3
    assert int.from_bytes(out[2:6].tobytes(), "little") == 0x1C8
    back = np.zeros(4, dtype=np.uint32)
    lg = C.c_uint32(0)
    assert ol.oracle().ans_oracle_read_prelude(out, back, C.byref(lg)) == 4
    assert lg.value == 3 and list(back) == [0, 3, 1, 4]


# ---------------------------------------------------------------- golden fixtures

def _load(name):
    with open(os.path.join(GOLD, name)) as fh:
        return json.load(fh)


def test_golden_small():
    gold = _load("small.json")
    assert len(gold) > 500
    assert sum(1 for e in gold if e["kind"] == "msb") > 60
    for e in gold:
        data = np.array(e["input"], dtype=np.uint32)
        s, info, _, _ = ol.oracle_encode(KIND[e["kind"]], e["f"], data)
        tag = (e["kind"], e["f"], e.get("family"), e["n"])
        assert len(s) == e["stream_len"], tag
        assert info.interp_bits == e["interp_bits"], tag
        assert info.max_sym == e["max_sym"] and info.log2_frame == e["log2_frame"], tag
        assert info.reorder_flag == e["reorder_flag"], tag
        assert hexs(ol.canonicalize(s, info)) == e["stream_hex"], tag
        # decode the *golden* bytes (reference-made) with the oracle decoder
        ref_stream = np.frombuffer(bytes.fromhex(e["stream_hex"]), dtype=np.uint8)
        assert np.array_equal(ol.oracle_decode(KIND[e["kind"]], e["f"], ref_stream, e["n"]), data), tag


@pytest.mark.parametrize("fixture", ["large.json", "f24.json", "f67.json"])
def test_golden_large(fixture):
    gold = _load(fixture)
    for e in gold:
        data = ol.gen_inputs(e["family"], e["n"], e["seed"])
        if e["kind"] == "rfold":
            data = data % np.uint32(1 << 21)
        assert hashlib.sha256(data.tobytes()).hexdigest() == e["input_sha256"], "generator drifted"
        s, info, _, _ = ol.oracle_encode(KIND[e["kind"]], e["f"], data)
        tag = (e["kind"], e["f"], e["family"], e["n"])
        assert len(s) == e["stream_len"], tag
        assert hashlib.sha256(ol.canonicalize(s, info).tobytes()).hexdigest() == e["stream_sha256"], tag
        assert np.array_equal(ol.oracle_decode(KIND[e["kind"]], e["f"], s, e["n"]), data), tag


# ---------------------------------------------------------------- against the real reference

needs_ref = pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")


@needs_ref
@pytest.mark.parametrize("f", [1, 2, 3, 5, 7])
def test_fold_unfold_vs_reference(f):
    rng = np.random.default_rng(f)
    xs = np.concatenate([np.arange(0, 70000, dtype=np.uint64),
                         rng.integers(0, 1 << 32, size=20000, dtype=np.uint64)]).astype(np.uint32)
    k = C.c_uint32(0)
    for x in xs[::7]:
        s = ol.oracle().ans_oracle_fold(f, int(x), C.byref(k))
        assert s == ol.ref().ref_fold_mapping(f, int(x))
        assert k.value == ol.ref().ref_fold_exception_bytes(f, s)
        assert ol.oracle().ans_oracle_unfold(f, s, None) == ol.ref().ref_fold_undo_mapping(f, s)


@needs_ref
def test_adjust_freqs_vs_reference():
    rng = np.random.default_rng(5)
    for f in (1, 3, 5):
        nf = 1 << (f + 9)
        for trial in range(40):
            kind = trial % 4
            freqs = np.zeros(nf, dtype=np.uint64)
            nsym = int(rng.integers(1, min(nf, 1021 << (f - 1))))
            if kind == 0:
                freqs[:nsym] = rng.integers(0, 50, size=nsym)
            elif kind == 1:
                freqs[:nsym] = (1e6 / np.arange(1, nsym + 1) ** 1.3).astype(np.uint64)
            elif kind == 2:
                freqs[rng.integers(0, nsym, size=max(1, nsym // 10))] = rng.integers(1, 1 << 20)
            else:
                freqs[:nsym] = 1
                freqs[0] = int(rng.integers(1, 1 << 28))
            if freqs.sum() == 0:
                freqs[0] = 3
            largest = int(np.nonzero(freqs)[0].max())
            a = np.zeros(largest + 1, dtype=np.uint32)
            b = np.zeros(largest + 1, dtype=np.uint32)
            Ma = ol.oracle().ans_oracle_adjust_freqs(freqs, nf, largest, a)
            Mb = ol.ref().ref_adjust_freqs(freqs, nf, largest, b)
            assert Ma == Mb and np.array_equal(a, b), (f, trial)


@needs_ref
def test_prelude_vs_reference():
    rng = np.random.default_rng(11)
    for trial in range(60):
        nsyms = int(rng.integers(1, 3000))
        lg = int(rng.integers(max(1, int(np.ceil(np.log2(nsyms)))), 17))
        M = 1 << lg
        # random composition of M into nsyms parts (zeros allowed)
        cuts = np.sort(rng.integers(0, M + 1, size=nsyms - 1))
        nf = np.diff(np.concatenate([[0], cuts, [M]])).astype(np.uint32)
        a = np.zeros(8 * nsyms + 64, dtype=np.uint8)
        b = np.zeros(8 * nsyms + 64, dtype=np.uint8)
        bits = C.c_uint32(0)
        na = ol.oracle().ans_oracle_write_prelude(nf, nsyms, M, a, C.byref(bits))
        nb = ol.ref().ref_serialize_prelude(nf, nsyms, M, b)
        assert na == nb
        vb = bits.value % 32
        if vb:
            w = int.from_bytes(b[nb - 4:nb].tobytes(), "little") & ((1 << vb) - 1)
            b[nb - 4:nb] = np.frombuffer(w.to_bytes(4, "little"), dtype=np.uint8)
        assert np.array_equal(a[:na], b[:nb]), trial
        back = np.zeros(nsyms, dtype=np.uint32)
        assert ol.ref().ref_load_prelude(a, back) == nsyms
        assert np.array_equal(back, nf)


@needs_ref
@pytest.mark.parametrize("kind", [ol.FOLD, ol.RFOLD])
@pytest.mark.parametrize("f", [1, 3, 5])
def test_streams_vs_reference(kind, f):
    for fam in ["uniform256", "uniform20", "geom0.01", "zipf20s1.2", "sparse_large", "boundaries"]:
        for n in (1, 4, 6, 999, 20000):
            d = ol.gen_inputs(fam, n, seed=77 * f + n)
            if kind == ol.RFOLD:
                d = d % np.uint32(1 << 20)
            s, info, _, _ = ol.oracle_encode(kind, f, d)
            for libname in ("libans_ref.so", "libans_ref_pattern.so"):
                r = ol.ref_encode(kind, f, d, libname)
                assert len(r) == len(s)
                assert np.array_equal(ol.canonicalize(r, info), s), (fam, n, libname)
            back = ol.ref_decode(kind, f, s, n)
            if kind == ol.RFOLD and info.reorder_flag == 0:
                # the reference decoder subtracts T unconditionally
                assert np.array_equal(back, ol.oracle_decode(kind, f, s, n, ref_f3_compat=True))
            else:
                assert np.array_equal(back, d)


# ---------------------------------------------------------------- properties

@pytest.mark.parametrize("f", [1, 2, 3, 4, 5, 6, 7])
def test_fold_unfold_identity(f):
    """src/test.cpp:54-64: unfold(fold(x)) == x with the k stripped low bytes cleared."""
    rng = np.random.default_rng(100 + f)
    T = 1 << (f + 7)
    xs = [0, 1, T - 1, T, T + 1, 256 * T - 1, 256 * T, 65536 * T - 1, 65536 * T, (1 << 30) - 1,
          (1 << 32) - 1] + [int(v) for v in rng.integers(0, 1 << 32, size=5000, dtype=np.uint64)]
    k = C.c_uint32(0)
    k2 = C.c_uint32(0)
    for x in xs:
        s = ol.oracle().ans_oracle_fold(f, x, C.byref(k))
        assert s < T + 3 * (255 << (f - 1))
        hi = ol.oracle().ans_oracle_unfold(f, s, C.byref(k2))
        assert k.value == k2.value
        assert hi == (x >> (8 * k.value)) << (8 * k.value)


@pytest.mark.parametrize("kind", [ol.FOLD, ol.RFOLD])
def test_roundtrip_and_f3(kind):
    # F3 trigger: sigma < T but values >= T -> reference decode is off by T, ours is not
    d = np.array([5, 1000, 5, 70000, 5, 5, 1000], dtype=np.uint32)
    s, info, _, _ = ol.oracle_encode(kind, 1, d)
    assert np.array_equal(ol.oracle_decode(kind, 1, s, d.size), d)
    if kind == ol.RFOLD:
        assert info.reorder_flag == 0
        bug = ol.oracle_decode(kind, 1, s, d.size, ref_f3_compat=True)
        assert list(bug) == [5, 744, 5, 69744, 5, 5, 744]
    for f in (1, 3, 5):
        for n in (1, 2, 3, 4, 5, 1023, 50001):
            d = ol.gen_inputs("zipf24", n, seed=n) if n < 2000 else ol.gen_inputs("zipf20s1.2", n, seed=n)
            s, info, _, _ = ol.oracle_encode(kind, f, d)
            assert np.array_equal(ol.oracle_decode(kind, f, s, n), d)


def test_checkpoints_restart_points():
    """Checkpoint (state, offset) pairs recorded by the oracle are valid decoder restart points:
    the four states and offset at symbol index s*C let decoding resume there."""
    d = ol.gen_inputs("zipf20s1.2", 10007, seed=3)
    C_ = 1024
    s, info, st, off = ol.oracle_encode(ol.FOLD, 1, d, ckpt_interval=C_)
    nfull = d.size - d.size % 4
    assert st.shape[0] == (nfull + C_ - 1) // C_ - 1
    M = 1 << info.log2_frame
    # offsets strictly decreasing with s (decoder walks the stream backwards)
    assert np.all(np.diff(off.astype(np.int64)) < 0)
    assert off[0] < len(s) - 32 and off[-1] >= info.prelude_bytes
    assert np.all(st >= 16 * M) and np.all(st < (16 * M) << 32)
    # the last 32 bytes hold state - 16M of the whole-block encoder = restart point of segment 0
    tail = np.frombuffer(s[-32:].tobytes(), dtype=np.uint64)
    assert np.array_equal(tail + np.uint64(16 * M), np.array(list(info.final_states), dtype=np.uint64))


needs_ref2 = pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")


@needs_ref2
def test_msb_streams_vs_reference():
    """ANSmsb (include/ans_msb.hpp, methods.hpp:499-515): restatement vs the compiled reference."""
    edge = np.array([0, 255, 256, 257, 65535, 65536, 65537, (1 << 24) - 1, 1 << 24, (1 << 24) + 1,
                     (1 << 30) - 1], dtype=np.uint32)
    for fam in ["uniform256", "uniform20", "geom0.01", "zipf20s1.2", "sparse_large", "boundaries"]:
        for n in (1, 4, 6, 999, 20000):
            d = ol.gen_inputs(fam, n, seed=5 + n)
            if n >= 999:
                d[:edge.size] = edge
            s, info, _, _ = ol.oracle_encode(ol.MSB, 0, d)
            for libname in ("libans_ref.so", "libans_ref_pattern.so"):
                r = ol.ref_encode(ol.MSB, 0, d, libname)
                assert len(r) == len(s) and np.array_equal(ol.canonicalize(r, info), s), (fam, n, libname)
            assert np.array_equal(ol.ref_decode(ol.MSB, 0, s, n), d)
            assert np.array_equal(ol.oracle_decode(ol.MSB, 0, s, n), d)


# ---------------------------------------------------------------- ANSint and pseudo_adaptive blocks

@needs_ref
def test_ansint_matches_reference():
    """ANSint (ans_int.hpp, methods.hpp:484-497): identity symbols, 32-bit frequencies, no u16 exit."""
    rng = np.random.default_rng(3)
    cases = [ol.gen_inputs("uniform256", 5000, 1), ol.gen_inputs("geom0.01", 20000, 2), ol.gen_inputs("geom0.4", 7, 3),
             np.minimum(ol.gen_inputs("zipf20s1.2", 30000, 4), 30000).astype(np.uint32),
             np.concatenate([np.ones(299990, dtype=np.uint32), np.arange(2, 12, dtype=np.uint32)]),  # frame above 2^16
             np.array([3] * 4000 + [1], dtype=np.uint32), np.array([5, 1], dtype=np.uint32),
             rng.integers(1, 2000, 2001).astype(np.uint32)]
    big = 0
    for d in cases:
        s, info, _, _ = ol.oracle_encode(ol.INT, 0, d)
        r = ol.ref_encode(ol.INT, 0, d)
        assert np.array_equal(ol.canonicalize(r, info), s), d[:8]
        assert np.array_equal(ol.oracle_decode(ol.INT, 0, s, d.size), d)
        assert np.array_equal(ol.ref_decode(ol.INT, 0, s, d.size), d)
        big += info.log2_frame > 16
    assert big >= 1  # the wide-frequency regime is covered


@needs_ref
@pytest.mark.parametrize("kind,f", [(ol.MSB, 0), (ol.INT, 0), (ol.FOLD, 1), (ol.FOLD, 3)])
def test_pseudo_adaptive_block_matches_reference(kind, f):
    """One block of src/pseudo_adaptive.cpp:85-130: alphabet header + codec stream of the rank-remapped block."""
    for fam, n in (("zipf20s1.2", 16384), ("uniform256", 4096), ("geom0.01", 8192), ("sparse_large", 3000),
                   ("constant", 1000), ("boundaries", 513), ("uniform20", 2048), ("zipf20s1.2", 3)):
        d = ol.gen_inputs(fam, n, seed=7)
        if fam == "sparse_large":
            d = d % np.uint32(1 << 20)  # the harness keeps the running sums of distinct values in 32 bits
        s, pinfo, info, _, _ = ol.oracle_pa_encode(kind, f, d)
        r, hb = ol.ref_pa_encode(kind, f, d)
        assert hb == pinfo.header_bytes and r.size == s.size, (fam, n)
        assert np.array_equal(ol.canonicalize_pa(r, pinfo, info), s), (fam, n)
        assert np.array_equal(ol.oracle_pa_decode(kind, f, s, n), d), (fam, n)
        assert pinfo.sigma == np.unique(d).size


def test_golden_compaction_and_ansint():
    """tests/golden/pa.json (made from oracle/_ref by make_pa_golden.py): pseudo_adaptive blocks for ANSmsb /
    ANSint / ANSfold and plain ANSint streams."""
    gold = _load("pa.json")
    assert len(gold) > 100
    kinds = {"fold": ol.FOLD, "msb": ol.MSB, "int": ol.INT}
    for e in gold:
        d = ol.gen_inputs(e["family"], e["n"], e["seed"])
        assert hashlib.sha256(d.tobytes()).hexdigest() == e["input_sha256"], "generator drifted"
        tag = (e["mode"], e["kind"], e["f"], e["family"], e["n"])
        if e["mode"] == "pa":
            s, pinfo, info, _, _ = ol.oracle_pa_encode(kinds[e["kind"]], e["f"], d)
            assert len(s) == e["stream_len"] and pinfo.sigma == e["sigma"] and pinfo.header_bytes == e["header_bytes"], tag
            assert hashlib.sha256(s.tobytes()).hexdigest() == e["stream_sha256"], tag
            assert np.array_equal(ol.oracle_pa_decode(kinds[e["kind"]], e["f"], s, e["n"]), d), tag
        else:
            s, info, _, _ = ol.oracle_encode(ol.INT, 0, d)
            assert len(s) == e["stream_len"], tag
            assert hashlib.sha256(ol.canonicalize(s, info).tobytes()).hexdigest() == e["stream_sha256"], tag


def _ansint_golden_input(e):
    if e["input"] == "family":
        return np.minimum(ol.gen_inputs(e["family"], e["n"], e["seed"]), np.uint32(e["clip"]))
    d = np.zeros(e["n"], dtype=np.uint32)
    for pos, val in e["pairs"]:
        d[pos] = val
    return d


def test_golden_plain_ansint_streams(oracle_built):
    """tests/golden/ansint.json: ans_int_compress bytes made by the real reference (make_ansint_golden.py) on lists whose
    values fit the GPU path's 16384-symbol model, frames from 2^5 to 2^20 (32-bit frequencies above 2^16)."""
    with open(os.path.join(GOLD, "ansint.json")) as fh:
        gold = json.load(fh)
    assert len(gold) >= 28 and max(e["log2_frame"] for e in gold) > 16
    for e in gold:
        d = _ansint_golden_input(e)
        assert hashlib.sha256(d.tobytes()).hexdigest() == e["input_sha256"]
        s, info, _, _ = ol.oracle_encode(ol.INT, 0, d)
        c = ol.canonicalize(s, info)
        assert c.size == e["stream_len"] and info.log2_frame == e["log2_frame"]
        if "stream_hex" in e:
            assert c.tobytes().hex() == e["stream_hex"]
        else:
            assert hashlib.sha256(c.tobytes()).hexdigest() == e["stream_sha256"]
        assert np.array_equal(ol.oracle_decode(ol.INT, 0, s, d.size), d)


def test_golden_plain_ansint_large_values(oracle_built):
    """tests/golden/ansint_large.json (round 4): ans_int_compress bytes made by the real reference on lists whose values
    reach 2^17, 2^20 and 2^22 -- the reference sizes its model by the largest value (ans_int.hpp:41-51); the restatement
    stays dense, the GPU path models such lists in rank space (csrc/ansx_intsparse.h).  Where oracle/_ref is present the
    reference itself is run on a few of them as well."""
    with open(os.path.join(GOLD, "ansint_large.json")) as fh:
        gold = json.load(fh)
    assert len(gold) == 39 and {e["log2_vmax"] for e in gold} == {17, 20, 22}
    for e in gold:
        d = ol.ansint_large_list(e["n"], 1 << e["log2_vmax"], e["seed"], e["shape"])
        assert hashlib.sha256(d.tobytes()).hexdigest() == e["input_sha256"]
        s, info, _, _ = ol.oracle_encode(ol.INT, 0, d)
        c = ol.canonicalize(s, info)
        assert c.size == e["stream_len"] and info.log2_frame == e["log2_frame"] and info.prelude_bytes == e["prelude_bytes"]
        assert hashlib.sha256(c.tobytes()).hexdigest() == e["stream_sha256"]
        if "stream_hex" in e:
            assert c.tobytes().hex() == e["stream_hex"]
        assert np.array_equal(ol.oracle_decode(ol.INT, 0, s, d.size), d)
        if ol.have_ref() and e["n"] in (7, 5000) and e["log2_vmax"] != 22:
            assert np.array_equal(ol.canonicalize(ol.ref_encode(ol.INT, 0, d), info), c)

