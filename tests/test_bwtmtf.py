"""tools/generate_bwtmtf (the reference's src/generate_bwtmtf.cpp pipeline: word / byte parse -> suffix
array -> BWT -> move-to-front ranks) against a naive Python restatement of the same steps, and the
committed config-5-shaped fixture (BWT-MTF ranks of a local text, expected streams made by oracle/_ref)."""
import collections
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TOOLS = os.path.join(ROOT, "ans_large_alphabet_amd", "tools")
GOLD = os.path.join(HERE, "golden")


def _tool():
    exe = os.path.join(TOOLS, "generate_bwtmtf.x")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", TOOLS, "generate_bwtmtf.x"])
    return exe


def _naive(T, n):
    """generate_bwtmtf.cpp:142-173 the slow way: sort the suffixes, BWT, deque move-to-front."""
    N = len(T)
    sa = sorted(range(N), key=lambda i: T[i:])
    bwt = [T[i - 1] if i else T[-1] for i in sa]
    seq = min(N - 1, n)
    alpha = collections.deque(range(max(T[:-1]) + 1))
    out = []
    for s in bwt[:seq]:
        r = alpha.index(s)
        out.append(r)
        del alpha[r]
        alpha.appendleft(s)
    return np.array(out, dtype=np.uint32)


def _word_parse(data, n):
    """generate_bwtmtf.cpp:67-99 (boost::split with token_compress_on)."""
    words = re.split(rb"[;, \n.?'()\-\"]+", data.lower())
    ids, T = {}, []
    for w in words:
        if w not in ids:
            ids[w] = len(ids) + 1
        T.append(ids[w])
        if len(T) >= n:
            break
    return T + [0]


@pytest.mark.parametrize("words", [True, False])
def test_tool_matches_naive_pipeline(tmp_path, words):
    text = open(os.path.join(ROOT, "SURVEY.md"), "rb").read()[:9000]
    text = b". " + text + b" end."          # leading / trailing delimiter runs: empty first / last token
    src = tmp_path / "in.txt"
    src.write_bytes(text)
    n = 1200 if words else 5000
    args = [_tool(), "-i", str(src), "-n", str(n), "-o", str(tmp_path / "out")] + (["-w"] if words else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    tag = "-WORD" if words else "-CHAR"
    got_text = np.fromfile(tmp_path / ("out%s.u32" % tag), dtype=np.uint32)
    got = np.fromfile(tmp_path / ("out%s-BWTMTF.u32" % tag), dtype=np.uint32)
    T = _word_parse(text, n) if words else list(text[:n]) + [0]
    assert np.array_equal(got_text, np.array(T[:-1][:n], dtype=np.uint32))
    assert np.array_equal(got, _naive(T, n))
    # -t writes the same numbers as decimal text (util.hpp read_file_text format)
    subprocess.check_call(args + ["-t"], stdout=subprocess.DEVNULL)
    txt = np.loadtxt(tmp_path / ("out%s-BWTMTF.txt" % tag), dtype=np.uint32)
    assert np.array_equal(txt, got)


def _run_tool_on(T, mode, n, tmp_path, tag):
    """Rebuild a text whose parse is T (words "w<id>" separated by blanks / the bytes themselves), run the tool."""
    src = tmp_path / ("in_%s.txt" % tag)
    if mode == "word":
        src.write_bytes(b" ".join(b"w%d" % t for t in T))
    else:
        src.write_bytes(bytes(T))
    args = [_tool(), "-i", str(src), "-n", str(n), "-o", str(tmp_path / ("out_%s" % tag))] + (["-w"] if mode == "word" else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    suffix = "-WORD" if mode == "word" else "-CHAR"
    text = np.fromfile(tmp_path / ("out_%s%s.u32" % (tag, suffix)), dtype=np.uint32)
    mtf = np.fromfile(tmp_path / ("out_%s%s-BWTMTF.u32" % (tag, suffix)), dtype=np.uint32)
    return text, mtf


def test_tool_matches_reference_pipeline_fixture(tmp_path):
    """tests/golden/bwtmtf_pipeline.json (made by make_bwtmtf_pipeline_golden.py from oracle/_ref's ref_bwtmtf: the
    statements of src/generate_bwtmtf.cpp:142-173 around the UNMODIFIED include/qsufsort.hpp): the tool's own suffix
    sorter and Fenwick-tree move-to-front must give the reference's ranks on every case; where oracle/_ref is present
    the same comparison runs live on fresh random texts too."""
    doc = json.load(open(os.path.join(GOLD, "bwtmtf_pipeline.json")))
    assert len(doc["cases"]) >= 7
    for i, c in enumerate(doc["cases"]):
        text, mtf = _run_tool_on(c["T"], c["mode"], c["n"], tmp_path, "g%d" % i)
        assert np.array_equal(text, np.array(c["T"][:c["n"]], dtype=np.uint32)), c["name"]
        assert np.array_equal(mtf, np.array(c["mtf"], dtype=np.uint32)), c["name"]
    if ol.have_ref() and hasattr(ol.ref(), "ref_bwtmtf"):
        rng = np.random.default_rng(99)
        for j in range(6):
            mode = "word" if j % 2 == 0 else "byte"
            m = int(rng.integers(50, 3000))
            if mode == "word":
                raw = np.minimum(rng.zipf(1.2 + 0.2 * j, size=m), 500).tolist()
                ids, T = {}, []
                for x in raw:
                    ids.setdefault(x, len(ids) + 1)
                    T.append(ids[x])
            else:
                T = rng.integers(1, 1 + int(rng.integers(2, 200)), size=m).tolist()
            text, mtf = _run_tool_on(T, mode, m, tmp_path, "l%d" % j)
            assert np.array_equal(mtf, ol.ref_bwtmtf(T + [0], m)), (mode, m)


def test_bwtmtf_fixture_against_reference_streams(oracle_built):
    """tests/golden/bwtmtf.u32 (+ .json made from oracle/_ref by make_bwtmtf_golden.py): the oracle
    reproduces the reference's bytes on real-text BWT-MTF ranks, whole list and per 16 Ki block."""
    with open(os.path.join(GOLD, "bwtmtf.json")) as fh:
        meta = json.load(fh)
    data = np.fromfile(os.path.join(GOLD, "bwtmtf.u32"), dtype=np.uint32)
    assert hashlib.sha256(data.tobytes()).hexdigest() == meta["input_sha256"]
    for e in meta["streams"]:
        part = data[e["first"]:e["first"] + e["n"]]
        s, info, _, _ = ol.oracle_encode(ol.FOLD if e["kind"] == "fold" else ol.RFOLD, e["f"], part)
        assert len(s) == e["stream_len"], e
        assert hashlib.sha256(ol.canonicalize(s, info).tobytes()).hexdigest() == e["stream_sha256"], e
