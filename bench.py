#!/usr/bin/env python3
"""bench.py — ANSfold/ANSrfold encode+decode throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): ANSfold-1 on 256 Mi uint32 drawn from Zipf(s=1.2) over
{1..2^20}, HBM-resident, per GPU (weak scaling; config 4's 2e9 ints over 8 GPUs is the same per-GPU
size).  One step = one encode (histogram -> normalise -> prelude -> 4-state rANS -> container) plus one
decode of the whole batch; with N > 1 every rank also ships its container to the root over RCCL
(send/recv, xGMI) and the root concatenates them into ONE container with the library's merge kernel
(ansx_merge_containers_dev) inside the timed step.  value = total ints of all ranks / step time, in
Mints/s.  Output is bit-exact per block against the reference CPU encoder (tests/test_gpu_parity.py);
the round trip is verified here on every run, and at N > 1 the merged container is decoded on the root
after the last step and compared with the generator's values.

Inputs are drawn on the device by the package's own generator kernels (ansx_generate_dev, the reference's
distributions: src/generate_inputs.cpp, include/zipf_dist.hpp); seeds are in the JSON line.

Extra objects on the JSON line:
  roofline       dominant kernel: algorithmic bytes (encode 4+c, decode c+4 bytes/int) / its
                 average launch duration (hipEvents inside libansx on the launch stream), against the
                 8 TB/s HBM peak; `traffic` is replayed from the committed PMC summary (traffic_source)
  cpu_baseline   the reference itself (oracle/_ref, "reference") or the C restatement ("port"), single
                 thread, on a bounded sample of the same data, rank 0 at N = 1 only
  extra_configs  (N = 1) the other single-GPU BASELINE configs, a few steps each: config 3 (ANSfold-3 and
                 ANSrfold-3 on Zipf over 2^24), config 1's data shape (uniform(1..256)), and the config-5
                 fallback (BWT-MTF ranks of a local text) -- each with round-trip check,
                 bits/int, per-kernel times and its own roofline
"""
import argparse
import collections
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED = 1234


def canonical_spec(spec):
    return "uniform1-256" if spec == "uniform256" else spec  # BASELINE config 1: uniform(1..256)


def gen_input(torch, A, ctx, spec, n, seed, device, first_index=0):
    """Synthetic input drawn on the device by the package's generator kernels; element i is a pure
    function of (seed, first_index + i), so rank r of N draws ITS slice of one global list."""
    out = torch.empty(n, dtype=torch.int32, device=device)
    A.generate_dev(ctx, canonical_spec(spec), out.data_ptr(), n, seed=seed, first_index=first_index)
    torch.cuda.synchronize()
    return out


def cpu_baseline(sample, kind, f, block_ints=16384, budget_s=12.0):
    """Reference CPU path, one thread, whole-list encode()+decode() as table_efficiency.cpp times it
    (min over runs).  Uses oracle/ strictly as the measured CPU comparator."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol

    n = sample.size
    use_ref = ol.have_ref()
    runs = 3
    t_enc, t_dec = 1e30, 1e30
    stream = None
    t_start = time.perf_counter()
    for i in range(runs):
        t0 = time.perf_counter()
        stream = ol.ref_encode(kind, f, sample) if use_ref else ol.oracle_encode(kind, f, sample)[0]
        t_enc = min(t_enc, time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s / 2:
            runs = i + 1
            break
    for i in range(runs):
        t0 = time.perf_counter()
        back = ol.ref_decode(kind, f, stream, n) if use_ref else ol.oracle_decode(kind, f, stream, n)
        t_dec = min(t_dec, time.perf_counter() - t0)
    ok = bool(np.array_equal(back, sample))
    # like-for-like row: the same data encoded/decoded block by block (one reference encode() per
    # block_ints ints, exactly the units the GPU processes), on a slice of the sample
    blk_n = min(n, 8 * (1 << 20))
    tb_enc = tb_dec = 0.0
    streams = []
    done = 0
    for a in range(0, blk_n, block_ints):
        part = np.ascontiguousarray(sample[a:a + block_ints])
        t0 = time.perf_counter()
        s_ = ol.ref_encode(kind, f, part) if use_ref else ol.oracle_encode(kind, f, part)[0]
        tb_enc += time.perf_counter() - t0
        streams.append((s_, part.size))
        done += part.size
        if tb_enc > budget_s / 3:  # (ANSrfold's reference encoder sorts a (max value + 1)-entry vector per call)
            break
    blk_n = done
    for s_, m in streams:
        t0 = time.perf_counter()
        _ = ol.ref_decode(kind, f, s_, m) if use_ref else ol.oracle_decode(kind, f, s_, m)
        tb_dec += time.perf_counter() - t0
    blocked = {"block_ints": block_ints, "ints": blk_n, "value": blk_n / (tb_enc + tb_dec) / 1e6,
               "enc_mints": blk_n / tb_enc / 1e6, "dec_mints": blk_n / tb_dec / 1e6,
               "bits_per_int": 8.0 * sum(s_.size for s_, _ in streams) / blk_n,
               "note": "includes ~10 us of ctypes/numpy call overhead per block"}
    return {
        "blocked": blocked,
        "value": n / (t_enc + t_dec) / 1e6, "unit": "Mints/s", "cores": 1,
        "kind": "reference" if use_ref else "port",
        "build": ("oracle/_ref: unmodified reference headers, clang++ -O3 -ffp-contract=off, no -march=native "
                  "(built in the authoring container, executed here)") if use_ref else "oracle/ans_oracle.c, gcc -O3",
        "sample": "first %d ints of the workload, one whole-list encode()+decode(), min of %d runs" % (n, runs),
        "enc_mints": n / t_enc / 1e6, "dec_mints": n / t_dec / 1e6,
        "bits_per_int": 8.0 * stream.size / n, "roundtrip_ok": ok,
        "cpu": _cpu_model(),
    }


def effective_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup's CPU quota (a GPU box hands a
    one-GPU job a share of the host's cores)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    cores = min(cores, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                        cores = min(cores, max(1, q // int(fh2.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return cores


def all_cores_baseline(sample, kind, f, block_ints, budget_s=10.0):
    """The reference CPU path on every host core this process may use: blocks are independent encode() calls
    (SURVEY 8d "optional all-cores row"), native threads over oracle/_ref taking blocks from a shared counter
    (oracle/ref_shim.cpp::ref_blocks_mt).  Like-for-like with the GPU run: the same block size.  Bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol

    if not ol.have_ref() or not hasattr(ol.ref(), "ref_blocks_mt"):
        return None
    cores = effective_cores()
    # size the sample to the budget from a short single-thread probe
    probe = np.ascontiguousarray(sample[:min(sample.size, 8 * block_ints)])
    _, pe, pd, _ = ol.ref_blocks_mt(kind, f, probe, block_ints, 1)
    per_int = max((pe + pd) / probe.size, 1e-12)
    n = int(min(sample.size, max(block_ints, budget_s * cores / per_int)))
    n -= n % block_ints if n > block_ints else 0
    part = np.ascontiguousarray(sample[:n])
    ok, t_enc, t_dec, tot = ol.ref_blocks_mt(kind, f, part, block_ints, cores)
    return {"value": n / (t_enc + t_dec) / 1e6, "unit": "Mints/s", "cores": cores, "kind": "reference", "block_ints": block_ints,
            "ints": n, "enc_mints": n / t_enc / 1e6, "dec_mints": n / t_dec / 1e6, "bits_per_int": 8.0 * tot / n, "roundtrip_ok": bool(ok),
            "cpu": _cpu_model(),
            "sample": "first %d ints of the workload in blocks of %d, %d native threads over oracle/_ref" % (n, block_ints, cores)}


def gpu_enc_dec_rates(torch, codec, d_in, n, d_out, cap, d_back, stream, runs=5):
    """GPU encode and decode rates the way the reference's harness times a codec (table_efficiency.cpp:32,78-101:
    NUM_RUNS = 5, minimum kept), with events around the whole encode (histogram + model + prelude + encode + container)
    and the whole decode, data resident in HBM."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    best_e = best_d = 1e30
    for i in range(runs + 1):  # (the first pass is a warm-up)
        ev[0].record()
        nb = codec.encode_dev(d_in.data_ptr(), n, d_out.data_ptr(), cap, stream=stream)
        ev[1].record()
        codec.decode_dev(d_out.data_ptr(), nb, d_back.data_ptr(), n, stream=stream)
        ev[2].record()
        torch.cuda.synchronize()
        if i:
            best_e = min(best_e, ev[0].elapsed_time(ev[1]))
            best_d = min(best_d, ev[1].elapsed_time(ev[2]))
    return {"enc_mints": n / best_e / 1e3, "dec_mints": n / best_d / 1e3, "enc_ms": best_e, "dec_ms": best_d,
            "combined_mints": n / (best_e + best_d) / 1e3,
            "policy": "events around the full encode / full decode, minimum of %d runs after one warm-up "
                      "(table_efficiency.cpp:32,78-101), data resident in HBM" % runs}


def host_buffer_rates(codec, sample, runs=3):
    """The host-buffer entry points (ansx_encode / ansx_decode: H2D + device path + D2H through pageable memory):
    the PCIe-inclusive rate, never the headline value."""
    import numpy as np

    best_e = best_d = 1e30
    cont = None
    for _ in range(runs):
        t0 = time.perf_counter()
        cont = codec.encode(sample)
        best_e = min(best_e, time.perf_counter() - t0)
        t0 = time.perf_counter()
        back = codec.decode(cont, sample.size)
        best_d = min(best_d, time.perf_counter() - t0)
    return {"ints": int(sample.size), "enc_mints": sample.size / best_e / 1e6, "dec_mints": sample.size / best_d / 1e6,
            "combined_mints": sample.size / (best_e + best_d) / 1e6, "roundtrip_ok": bool(np.array_equal(back, sample)),
            "note": "ansx_encode / ansx_decode on pageable host buffers: H2D of the input, device path, D2H of the container "
                    "(and the reverse); minimum of %d runs" % runs}


def config1_rows(A):
    """BASELINE config 1: ANSfold-1 on 10^6 uniform(1..256) ints through the Table-10 harnesses -- the reference's
    loop over the reference's own codecs on the CPU (oracle/_ref/table10_cpu.x), and this package's harness
    (tools/table_efficiency.x, host buffers, so launch latency and PCIe dominate at this size)."""
    import tempfile

    out = {"input": "uniform(1..256), 1 000 000 ints, ansx_generate_host seed %d" % SEED}
    data = A.generate_host("uniform1-256", 1000000, seed=SEED)
    with tempfile.TemporaryDirectory() as tmp:
        data.tofile(os.path.join(tmp, "uniform1-256.u32"))
        cpu = os.path.join(ROOT, "oracle", "_ref", "table10_cpu.x")
        if os.path.exists(cpu):
            r = subprocess.run([cpu, "-i", tmp], capture_output=True, text=True, timeout=120)
            out["cpu_reference_rows"] = r.stdout if r.returncode == 0 else "failed: " + r.stderr[-300:]
            out["cpu_reference_harness"] = "oracle/_ref/table10_cpu.x: src/table_efficiency.cpp's run<>() over oracle/_ref (1 thread, %s)" % _cpu_model()
        else:
            out["cpu_reference_rows"] = None
        tools = os.path.join(ROOT, "ans_large_alphabet_amd", "tools")
        exe = os.path.join(tools, "table_efficiency.x")
        try:
            if not os.path.exists(exe):
                subprocess.check_call(["make", "-s", "-C", tools, "table_efficiency.x"])
            r = subprocess.run([exe, "-i", tmp, "--bits"], capture_output=True, text=True, timeout=120)
            out["gpu_rows"] = r.stdout if r.returncode == 0 else "failed: " + r.stderr[-300:]
        except Exception as exc:  # noqa: BLE001
            out["gpu_rows"] = "failed: %r" % (exc,)
    return out


def compaction_rows(torch, A, ctx, device, n=64 * (1 << 20)):
    """bits/int and rate with and without the per-block alphabet compaction on two inputs: the headline Zipf list
    (every block sees ~2500 distinct values: listing them costs more than folding their low bytes) and a list whose
    blocks each use a small alphabet of LARGE values (document-local vocabularies: 200 distinct 24-bit values per
    16 Ki-int block, Zipf-ranked) -- the case the reference's pseudo_adaptive harness is about."""
    import numpy as np

    rng = np.random.default_rng(SEED)
    nblk = n // 16384
    vocab = rng.integers(1 << 16, 1 << 24, size=(nblk, 200), dtype=np.uint32)
    ranks = np.minimum(rng.zipf(1.3, size=(nblk, 16384)) - 1, 199)
    local = np.take_along_axis(vocab, ranks, axis=1).reshape(-1).astype(np.uint32)
    inputs = {"zipf20s1.2": None, "block-local vocabularies (200 distinct 24-bit values per block, Zipf(1.3) ranks)": local}
    rows = []
    for name, host in inputs.items():
        if host is None:
            d_in = gen_input(torch, A, ctx, "zipf20s1.2", n, SEED, device)
        else:
            d_in = torch.from_numpy(host.view("int32")).to(device)
        for label, cn, compact in (("ANSfold-1", "fold", False), ("ANSfold-1 + compaction", "fold", True), ("ANSint + compaction", "int", True),
                                   ("ANSint plain (the reference's value-range prelude per block)", "intplain", False)):
            r = run_single(torch, A, ctx, device, cn, 1 if cn == "fold" else 0, name, n, 3, 2, d_in=d_in, compact=compact,
                           block=8192 if cn == "int" else 0, profile=False)
            rows.append({"input": name, "codec": label, "bits_per_int": r["bits_per_int"], "value": r["value"], "unit": "Mints/s",
                         "roundtrip_ok": r["roundtrip_ok"]})
        del d_in
        torch.cuda.empty_cache()
    return rows


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary of this workload
    (profiles/*_hbm_traffic_pmc.json, one per profiled configuration: FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, separate passes, gfx950
    correction calibrated on k_fold_hist).  Counters cannot be read from inside this process, so the
    figure is REPLAYED from that file and only when its workload string is this run's; returns
    (bytes, source) or (None, reason)."""
    try:
        paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic_pmc.json")))
        if not paths:
            return None, "no PMC summary under profiles/"
        best = rel = None
        seen = set()
        for path in reversed(paths):  # newest (by name: r<round>_<tag>) summary of THIS workload
            with open(path) as fh:
                doc = json.load(fh)
            seen.add(doc.get("workload"))
            if doc.get("workload") == workload:
                best, rel = doc, os.path.relpath(path, ROOT)
                break
        if best is None:
            return None, "no PMC summary under profiles/ for this workload (profiled: %s)" % "; ".join(sorted(str(x) for x in seen))
        for name, v in best["kernels"].items():
            base = name.split("<")[0]
            # profile labels are the launch sites' ("k_decode", "k_encode_gtab"), the PMC summary has the
            # kernels' own names ("k_decode_rank<...>", "k_encode<2, ...>")
            if base == kernel or (kernel == "k_decode" and base == "k_decode_rank") or (kernel in ("k_encode_gtab", "k_encode") and base in ("k_encode", "k_encode_pc")) \
                    or (kernel == "k_rfold_remap" and base.startswith("k_rfold_remap")):
                return v["hbm_bytes"], "replayed from %s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)" % rel
        return None, "%s has no entry for %s" % (rel, kernel)
    except Exception as exc:  # noqa: BLE001
        return None, "unreadable PMC summary: %r" % (exc,)


# SIMD-level issue cost of a vector instruction on gfx950 (tests/tools/ubench_valu2.hip, round 4: four waves per SIMD of
# independent f64 / VOP3 / DPP instructions retire one per 4.1-4.7 cycles and SIMD; only 4-byte-encoded 32-bit VOP1/VOP2
# forms reach 2.4) and the clock the chip holds in these loops
ISSUE_CYCLES_PER_VALU = 4.3
ISSUE_CLOCK_GHZ = 2.1
N_SIMD = 1024


def issue_floor(kernel, workload_hint=None):
    """Lower bound on `kernel`'s duration from instruction issue alone: the vector instructions ALL its waves execute
    (SQ_INSTS_VALU from the newest committed rocprofv3 SQ-counter summary, profiles/*_sq_counters.json) spread evenly over
    the chip's 1024 SIMDs at ISSUE_CYCLES_PER_VALU cycles each.  Says which bound a kernel is against next to the HBM
    fraction: a kernel at 0.25 of the HBM peak whose issue floor is 60 % of its duration is not going to be fixed by
    memory-side work.  Returns (ms, source) or (None, reason)."""
    try:
        paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")))
        docs = []
        for path in reversed(paths):
            with open(path) as fh:
                docs.append((path, json.load(fh)))
        # the newest summary of THIS workload; summaries older than round 4 carry no workload string and are config 2's
        ranked = [pd for pd in docs if workload_hint and pd[1].get("workload") == workload_hint] \
            + [pd for pd in docs if pd[1].get("workload") is None and workload_hint and "zipf20s1.2" in workload_hint and "ANSfold-1," in workload_hint]
        for path, doc in ranked:
            for name, v in doc.get("kernels", {}).items():
                base = name.split("<")[0]
                if base == kernel or (kernel == "k_decode" and base == "k_decode_rank") or (kernel in ("k_encode_gtab", "k_encode") and base in ("k_encode", "k_encode_pc")):
                    # (collect.sh profiles `bench.py --steps 1 --warmup 0`: the discovery step + the timed one = 2 launches of
                    # every per-step kernel; summaries written since round 4 carry the dispatch count themselves)
                    waves = float(v["waves"]) / max(1.0, float(v.get("launches", 2)))
                    insts = float(v["SQ_INSTS_VALU_per_wave"]) * waves
                    ms = insts * ISSUE_CYCLES_PER_VALU / N_SIMD / (ISSUE_CLOCK_GHZ * 1e9) * 1e3
                    return ms, ("%s: %.0f VALU instructions per wave x %.0f waves per launch, %.1f cycles each per SIMD at %.1f GHz"
                                % (os.path.relpath(path, ROOT), v["SQ_INSTS_VALU_per_wave"], waves, ISSUE_CYCLES_PER_VALU, ISSUE_CLOCK_GHZ))
        return None, "no SQ-counter summary for this kernel under profiles/"
    except Exception as exc:  # noqa: BLE001
        return None, "unreadable SQ summary: %r" % (exc,)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_profile(torch, ctx, codec, d_in, n, d_out, cap, d_back, stream, c_bytes, workload, reps=3):
    """Per-kernel hipEvent timing pass (outside the timed region) -> (kernels, roofline)."""
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(reps):
        nbp = codec.encode_dev(d_in.data_ptr(), n, d_out.data_ptr(), cap, stream=stream)
        codec.decode_dev(d_out.data_ptr(), nbp, d_back.data_ptr(), n, stream=stream)
    torch.cuda.synchronize()
    prof = ctx.profile_get()
    ctx.profile(False)
    kernels = {name: {"avg_ms": ms / max(cnt, 1), "launches_per_step": cnt / reps} for name, ms, cnt in prof}
    per_step = {name: ms / reps for name, ms, cnt in prof}
    dom = max(per_step, key=per_step.get)
    avg_ms = kernels[dom]["avg_ms"]
    # algorithmic bytes per launch: the encoder reads 4 B/int and writes c, the decoder
    # reads c and writes 4; other kernels are priced by what they must touch
    enc, dec = (4 + c_bytes) * n, (c_bytes + 4) * n
    alg = {"k_encode": enc, "k_encode_gtab": enc, "k_decode": dec, "k_decode_gtab": dec, "k_decode_table": dec,
           "k_fold_hist": 4.0 * n, "k_model_fused": 4.0 * n, "k_compact": 2 * c_bytes * n, "k_assemble": 2 * c_bytes * n,
           "k_rfold_remap": 8.0 * n}.get(dom, 4.0 * n)
    achieved = alg / (avg_ms * 1e-3) / 1e9
    traffic, source = pmc_traffic(dom, workload)
    floor_ms, floor_src = issue_floor(dom, workload)
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
                "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms,
                "issue_floor_ms": floor_ms, "issue_floor_source": floor_src}
    return kernels, roofline


def single_stream_row(torch, A, ctx, device, codec_name, fidelity, d_in, m=4 * (1 << 20)):
    """ANSX_SINGLE_STREAM: ONE plain reference stream for the whole list (the bytes of ANSfold<f>::encode, decodable by the
    reference) -- four lanes of one wave, there for byte compatibility, not speed.  Timed on the first m ints."""
    codec = make_codec(A, ctx, codec_name, fidelity, block=A.SINGLE_STREAM)
    m = min(m, d_in.numel())
    cap = codec.bound(m)
    d_out = torch.empty(cap, dtype=torch.uint8, device=device)
    d_back = torch.zeros(m, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    best = [1e30, 1e30]
    nb = 0
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = codec.encode_dev(d_in.data_ptr(), m, d_out.data_ptr(), cap, stream=stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        codec.decode_dev(d_out.data_ptr(), nb, d_back.data_ptr(), m, stream=stream)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1)]
    ok = bool(torch.equal(d_back, d_in[:m]))
    return {"ints": m, "enc_mints": m / best[0] / 1e6, "dec_mints": m / best[1] / 1e6, "value": m / (best[0] + best[1]) / 1e6,
            "unit": "Mints/s", "bits_per_int": 8.0 * nb / m, "roundtrip_ok": ok}


def first_call_row(torch, ctx, codec, d_in, n, d_out, cap, d_back, stream):
    """The first encode + decode of a geometry on a context that knows nothing about it: the encoder reads the largest
    alphabet back mid-call and runs the exact model kernels, the decoder reads the header back before it launches anything
    (stats.path 0).  The steady state the headline quotes starts with the second call."""
    ctx.debug_set("ANSX_FORGET_HINTS", "1")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nb = codec.encode_dev(d_in.data_ptr(), n, d_out.data_ptr(), cap, stream=stream)
    path = ctx.last_encode_stats()["path"]
    codec.decode_dev(d_out.data_ptr(), nb, d_back.data_ptr(), n, stream=stream)
    torch.cuda.synchronize()
    first_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    nb = codec.encode_dev(d_in.data_ptr(), n, d_out.data_ptr(), cap, stream=stream)
    path2 = ctx.last_encode_stats()["path"]
    codec.decode_dev(d_out.data_ptr(), nb, d_back.data_ptr(), n, stream=stream)
    torch.cuda.synchronize()
    second_ms = (time.perf_counter() - t0) * 1e3
    return {"first_call_ms": first_ms, "encode_path": path, "second_call_ms": second_ms, "second_encode_path": path2}


def make_codec(A, ctx, codec_name, fidelity, block=0, ckpt=0, compact=False):
    if codec_name == "int":
        return A.ANSint(ctx=ctx, block_ints=block, ckpt_interval=ckpt)
    if codec_name == "intplain":  # the values themselves as symbols (ans_int.hpp): beyond 16384, a block's ranks (csrc/ansx_intsparse.h)
        return A.ANSint(ctx=ctx, block_ints=block, ckpt_interval=ckpt, compact=False)
    if codec_name == "msb":
        return A.ANSmsb(ctx=ctx, block_ints=block, ckpt_interval=ckpt, compact=compact)
    cls = A.ANSfold if codec_name == "fold" else A.ANSrfold
    return cls(fidelity, ctx=ctx, block_ints=block, ckpt_interval=ckpt, compact=compact)


def workload_string(codec_name, f, n, spec, block, ckpt):
    return "ANS%s-%d, %d ints, %s, block %d, ckpt %d" % (codec_name, f, n, spec, block, ckpt)


def run_single(torch, A, ctx, device, codec_name, fidelity, spec, n, steps, warmup, block=0, ckpt=0, d_in=None,
               seed=SEED, cpu_sample=0, compact=False, profile=True):
    """One single-GPU configuration: timed encode+decode steps, round trip, per-kernel pass."""
    codec = make_codec(A, ctx, codec_name, fidelity, block, ckpt, compact)
    if d_in is None:
        d_in = gen_input(torch, A, ctx, spec, n, seed, device)
    cap = min(codec.bound(n), 8 * n + (64 << 20))
    d_out = torch.empty(cap, dtype=torch.uint8, device=device)
    d_back = torch.zeros(n, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    nb = 0
    for _ in range(warmup):
        nb = codec.encode_dev(d_in.data_ptr(), n, d_out.data_ptr(), cap, stream=stream)
        codec.decode_dev(d_out.data_ptr(), nb, d_back.data_ptr(), n, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        nb = codec.encode_dev(d_in.data_ptr(), n, d_out.data_ptr(), cap, stream=stream)
        codec.decode_dev(d_out.data_ptr(), nb, d_back.data_ptr(), n, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(torch.equal(d_back, d_in))
    stats = ctx.last_encode_stats()
    c_bytes = nb / n
    workload = workload_string(codec_name, fidelity, n, spec, block or A.DEFAULT_BLOCK_INTS, ckpt or A.DEFAULT_CKPT_INTERVAL)
    kernels = roofline = None
    if profile:
        kernels, roofline = kernel_profile(torch, ctx, codec, d_in, n, d_out, cap, d_back, stream, c_bytes, workload)
    rates = gpu_enc_dec_rates(torch, codec, d_in, n, d_out, cap, d_back, stream, runs=5 if profile else 3)
    res = {"workload": workload, "codec": codec.name(), "distribution": spec, "ints": n, "seed": seed, "gpu_rates": rates,
           "value": n * steps / dt / 1e6, "unit": "Mints/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
           "roundtrip_ok": ok, "bits_per_int": 8 * c_bytes, "near_threshold_decisions": stats["near_threshold_decisions"],
           "encode_path": stats["path"], "roofline": roofline, "kernels": kernels}
    if cpu_sample:
        kind = A.FOLD if codec_name == "fold" else A.RFOLD
        sample = d_in[:min(n, cpu_sample)].cpu().numpy().view("uint32")
        res["cpu_baseline"] = cpu_baseline(sample, kind, fidelity, block or A.DEFAULT_BLOCK_INTS, budget_s=6.0)
        res["speedup_vs_cpu_1thread"] = res["value"] / res["cpu_baseline"]["value"]
    del d_out, d_back
    return res


def bwtmtf_ranks(max_words=8 * (1 << 20)):
    """Config 5 fallback: BWT-MTF ranks of a local text, made by the package's own
    generate_bwtmtf tool (the reference's src/generate_bwtmtf.cpp pipeline) from the Python standard
    library's sources of this image (English prose + code, a deterministic file list)."""
    import numpy as np

    tools = os.path.join(ROOT, "ans_large_alphabet_amd", "tools")
    exe = os.path.join(tools, "generate_bwtmtf.x")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", tools, "generate_bwtmtf.x"])
    files = sorted(glob.glob("/usr/lib/python3.10/**/*.py", recursive=True))
    files += sorted(glob.glob("/usr/local/lib/python3.10/dist-packages/torch/**/*.py", recursive=True))
    if not files:
        return None, "no local text found"
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ansx_bwtmtf_%d" % os.getuid())
    os.makedirs(tmp, exist_ok=True)
    corpus = os.path.join(tmp, "corpus.txt")
    total = 0
    with open(corpus, "wb") as out:
        for f in files:
            try:
                with open(f, "rb") as fh:
                    b = fh.read()
            except OSError:
                continue
            out.write(b)
            total += len(b)
            if total > 6 * max_words:  # ~6 bytes per word
                break
    subprocess.check_call([exe, "-i", corpus, "-n", str(max_words), "-w", "-o", os.path.join(tmp, "pylib")],
                          stdout=subprocess.DEVNULL)
    ranks = np.fromfile(os.path.join(tmp, "pylib-WORD-BWTMTF.u32"), dtype=np.uint32)
    return ranks, ("word-parsed BWT-MTF ranks of the first %d bytes of this image's Python sources (/usr/lib/python3.10, "
                   "then torch), tools/generate_bwtmtf.x" % total)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--ints", dest="n", type=int, default=256 * (1 << 20), help="ints per GPU")
    ap.add_argument("--codec", default="fold", choices=["fold", "rfold", "msb", "int"])
    ap.add_argument("--compact", action="store_true",
                    help="per-block alphabet compaction (src/pseudo_adaptive.cpp:85-130); implied by --codec int")
    ap.add_argument("--fidelity", type=int, default=1)
    ap.add_argument("--dist", default="zipf20s1.2", help="zipf<log2 n>[s<q>] | uniform<lo>-<hi> | uniform256 | geom<p>")
    ap.add_argument("--block", type=int, default=0, help="ints per block (0 = library default)")
    ap.add_argument("--ckpt", type=int, default=0, help="restart interval (0 = library default)")
    ap.add_argument("--cpu-sample", type=int, default=64 * (1 << 20))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs rows (N = 1)")
    ap.add_argument("--no-host-api", action="store_true",
                    help="skip the host-buffer API row (its calls run on a 64 Mi-int sample: under rocprofv3 --stats they "
                         "would pull the per-kernel averages away from the timed steps')")
    ap.add_argument("--no-frontier", action="store_true", help="skip the block size / restart interval frontier rows (N = 1)")
    ap.add_argument("--gather-root", default="auto", choices=["auto", "fixed", "rotate"],
                    help="N > 1: root of the per-step container gather: rank k mod N of step k (rotate: spreads the "
                         "traffic over all xGMI links), always rank 0 (fixed), or auto (default): rotate, set up and "
                         "VERIFIED during warm-up -- one merged container per root decoded and checked against the "
                         "generator -- with a fall-back to fixed, agreed by all ranks, on any failure")
    ap.add_argument("--test-fail-rotate", action="store_true",
                    help="rehearsals: the verification of the rotating root reports a mismatch on root 1, so that "
                         "--gather-root auto exercises its fall-back")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N > 1 control-flow rehearsal on ONE GPU: gloo backend, every rank on cuda:0, "
                         "containers gathered through host copies (not a measurement)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 through torch.distributed.run (one rank per GPU)")
    if args.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        if args.rehearse_gloo:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)  # RCCL on ROCm
    cdev = "cpu" if args.rehearse_gloo else device  # where control-plane tensors of the collectives live

    import ans_large_alphabet_amd as A

    ctx = A.Context(local_rank)
    codec = make_codec(A, ctx, args.codec, args.fidelity, args.block, args.ckpt, args.compact)
    kind = {"fold": A.FOLD, "rfold": A.RFOLD, "msb": A.MSB, "int": A.INT}[args.codec]
    n = args.n
    block_ints = args.block or (8192 if args.codec == "int" else A.DEFAULT_BLOCK_INTS)
    if world > 1 and n % block_ints:
        raise SystemExit("N > 1: ints per GPU must be a multiple of the block size (whole blocks per rank)")

    # ---- synthetic input, resident in HBM: rank r holds ints [r n, (r+1) n) of ONE global list
    d_in = gen_input(torch, A, ctx, args.dist, n, SEED, device, first_index=rank * n)
    cap = min(codec.bound(n), 8 * n + (64 << 20))
    rotate = world > 1 and args.gather_root in ("auto", "rotate")
    root_policy = {"requested": args.gather_root, "used": None, "fallback_reason": None}
    DEPTH = 1 if world == 1 else (min(4, max(2, world)) if rotate else 2)
    outs = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(DEPTH)]
    d_out = outs[0]
    d_back = torch.zeros(n, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    import datetime
    WAIT = datetime.timedelta(seconds=180)  # every wait on a transfer is bounded: a hang exits non-zero

    # N > 1.  Every step ends with the rank containers concatenated into one container on a root GPU.
    # A rank's container is ~0.29 GB, i.e. several ms on one xGMI link -- longer than the step's compute
    # -- so the transfer of step k overlaps the decode of step k and the following encode (DEPTH buffers;
    # a buffer is reused only after the transfer that reads it has been waited for).  Transfers have a
    # FIXED size agreed once before the timed region (largest warm-up container + slack: the true size is
    # in the container header), so no per-step size exchange or host read-back sits on the path; the root
    # merges with ansx_merge_containers_dev when the step's transfers have landed.
    groups = [None] * DEPTH
    setup_ok = 1
    if dist is not None and rotate:
        try:
            groups = [dist.new_group(ranks=list(range(world)), timeout=WAIT) for _ in range(DEPTH)]
        except Exception as exc:  # noqa: BLE001
            print("bench.py rank %d: rotating-root setup failed (%r)" % (rank, exc), file=sys.stderr)
            setup_ok = 0

    state = {"k": 0, "xfer": 0, "merged_bytes": 0, "merged_root": 0}
    pending = collections.deque()  # (step, root, works, slot) of the steps still in flight
    recv = {}                      # per pipeline slot on a root: receive buffer [world][xfer]
    merged = [None]

    def all_agree(flag_value):
        """MIN over ranks of a 0/1 flag: a failure on ANY rank sends EVERY rank the same way (ranks must not diverge)."""
        flag = torch.tensor([flag_value], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item())

    def to_fixed(reason):
        """Fall back to rank 0 as the root of every step (buffers of the rotating form are dropped)."""
        nonlocal rotate, DEPTH, outs, groups
        root_policy["fallback_reason"] = reason
        rotate = False
        DEPTH = 2
        outs = outs[:2]
        groups = [None] * DEPTH
        recv.clear()
        merged[0] = None
        alloc_root_buffers()

    def alloc_root_buffers():
        x = state["xfer"]
        if not x:
            return
        roots = range(world) if rotate else [0]
        if rank in roots:
            for slot in range(DEPTH):
                recv[slot] = torch.zeros(world * x, dtype=torch.uint8, device=cdev)
            merged[0] = torch.zeros(world * x + (1 << 20), dtype=torch.uint8, device=device)

    if dist is not None and rotate and all_agree(setup_ok) == 0:
        to_fixed("communicator setup failed on some rank")

    def finish_one():
        k, root, works, slot = pending.popleft()
        for w in works:
            if not w.wait(WAIT):
                raise SystemExit("bench.py rank %d: transfer of step %d timed out" % (rank, k))
        if rank == root and world > 1 and state["xfer"]:
            buf = recv[slot]
            x = state["xfer"]
            if args.rehearse_gloo:
                buf = buf.to(device)
            ptrs = [buf.data_ptr() + r * x for r in range(world)]
            state["merged_bytes"] = ctx.merge_containers_dev(ptrs, [x] * world, merged[0].data_ptr(), merged[0].numel())
            state["merged_root"] = root

    def check_merged(root):
        """On `root`: decode its merged container (N x n ints) and compare three rank slices with the generator's values."""
        mine = True
        if rank == root:
            total = world * n
            whole = torch.empty(total, dtype=torch.int32, device=device)
            codec.decode_dev(merged[0].data_ptr(), state["merged_bytes"], whole.data_ptr(), total, stream=stream)
            torch.cuda.synchronize()
            for r in sorted(set([0, world // 2, world - 1])):
                want = gen_input(torch, A, ctx, args.dist, n, SEED, device, first_index=r * n)
                mine = mine and bool(torch.equal(whole[r * n:(r + 1) * n], want))
            del whole
        return mine

    def step():
        k = state["k"]
        state["k"] += 1
        slot = k % DEPTH
        out = outs[slot]
        nb = codec.encode_dev(d_in.data_ptr(), n, out.data_ptr(), cap, stream=stream)
        works, root = [], 0
        if dist is not None and state["xfer"]:
            root = (k % world) if rotate else 0
            x = state["xfer"]
            if nb > x:
                raise SystemExit("container outgrew the agreed transfer size")
            src = out[:x].cpu() if args.rehearse_gloo else out[:x]
            if rank == root:
                buf = recv[slot]
                buf[rank * x:(rank + 1) * x].copy_(src)
                ops = [dist.P2POp(dist.irecv, buf[r * x:(r + 1) * x], r, groups[slot]) for r in range(world) if r != root]
            else:
                ops = [dist.P2POp(dist.isend, src, root, groups[slot])]
            works = dist.batch_isend_irecv(ops) if ops else []
        codec.decode_dev(out.data_ptr(), nb, d_back.data_ptr(), n, stream=stream)
        pending.append((k, root, works, slot))
        while len(pending) > DEPTH - 1:  # the next user of a buffer must find its transfer finished
            finish_one()
        return nb

    def sync_all():
        while pending:
            finish_one()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    nb = step()  # first warm-up step: also learns the alphabet hint and, at N > 1, the transfer size
    sync_all()
    if dist is not None:
        t = torch.tensor([nb], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        x = int(t.item())
        x = min(cap, (x + x // 32 + 4096 + 15) // 16 * 16)
        state["xfer"] = x
        alloc_root_buffers()
        if rotate:
            # establish every (communicator, root) connection, then one full cycle of roots with each merged
            # container decoded and checked BEFORE the timed region; any failure on any rank -> fixed root
            ok_flag, why = 1, None
            try:
                if not args.rehearse_gloo:
                    tiny = torch.zeros(64, dtype=torch.uint8, device=device)
                    for g_ in groups:
                        for d_ in range(world):
                            if rank == d_:
                                ops = [dist.P2POp(dist.irecv, tiny.clone(), r, g_) for r in range(world) if r != d_]
                            else:
                                ops = [dist.P2POp(dist.isend, tiny, d_, g_)]
                            for w in dist.batch_isend_irecv(ops):
                                if not w.wait(WAIT):
                                    raise RuntimeError("connection set-up timed out")
                    torch.cuda.synchronize()
                for _ in range(world):
                    step()
                    sync_all()
                    root = (state["k"] - 1) % world
                    if not check_merged(root) or (args.test_fail_rotate and root == 1 and rank == 1):
                        ok_flag, why = 0, "merged container of root %d did not match the generator" % root
            except SystemExit:
                raise
            except Exception as exc:  # noqa: BLE001
                print("bench.py rank %d: rotating-root verification failed (%r)" % (rank, exc), file=sys.stderr)
                ok_flag, why = 0, repr(exc)
            if all_agree(ok_flag) == 0:
                if args.gather_root == "rotate":
                    raise SystemExit("--gather-root rotate failed its verification: %s" % why)
                pending.clear()
                to_fixed(why or "verification failed on another rank")
    for _ in range(max(args.warmup - 1, 1 if dist is not None else 0)):
        nb = step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nb = step()
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ok = bool(torch.equal(d_back, d_in))
    merged_ok = None
    if dist is not None:
        # the root of the LAST step decodes the merged container and checks it against the generator
        last_root = ((state["k"] - 1) % world) if rotate else 0
        mo = torch.tensor([1 if check_merged(last_root) else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(mo, op=dist.ReduceOp.MIN)
        merged_ok = bool(int(mo.item()))
        ok = ok and merged_ok
        root_policy["used"] = "rotate" if rotate else "fixed"
    ms_per_step = dt / args.steps * 1e3
    value = world * n * args.steps / dt / 1e6
    c_bytes = nb / n  # compressed bytes per int, everything included
    stats = ctx.last_encode_stats()
    workload = workload_string(args.codec, args.fidelity, n, args.dist, block_ints, args.ckpt or A.DEFAULT_CKPT_INTERVAL)

    # ---- per-kernel timing pass (separate from the timed region) -> roofline of the dominant kernel
    roofline = kernels = None
    if not args.no_profile:
        kernels, roofline = kernel_profile(torch, ctx, codec, d_in, n, d_out, cap, d_back, stream, c_bytes, workload)

    gpu_rates = None
    if world == 1 and not args.no_profile:
        gpu_rates = gpu_enc_dec_rates(torch, codec, d_in, n, d_out, cap, d_back, stream)

    cpu = cpu_all = host_api = None
    if rank == 0 and world == 1 and not args.no_cpu and not args.compact and args.codec != "int":
        m = min(n, args.cpu_sample)
        sample = d_in[:m].cpu().numpy().view("uint32")
        cpu = cpu_baseline(sample, kind, args.fidelity, block_ints)
        cpu_all = all_cores_baseline(sample, kind, args.fidelity, block_ints)
        host_api = None if args.no_host_api else host_buffer_rates(codec, sample[:min(m, 64 * (1 << 20))])

    # ---- two rows the steady-state headline hides (N = 1): the byte-for-byte drop-in mode and a geometry's first call
    single_stream = first_call = None
    if rank == 0 and world == 1 and not args.no_extra:
        try:
            single_stream = single_stream_row(torch, A, ctx, device, args.codec, args.fidelity, d_in)
        except Exception as exc:  # noqa: BLE001
            single_stream = {"error": repr(exc)}
        try:
            first_call = first_call_row(torch, ctx, codec, d_in, n, outs[0], cap, d_back, stream)
        except Exception as exc:  # noqa: BLE001
            first_call = {"error": repr(exc)}

    # ---- the other single-GPU BASELINE configs, a few steps each (N = 1 only)
    extra = None
    if rank == 0 and world == 1 and not args.no_extra:
        del outs, d_out, d_back, d_in
        torch.cuda.empty_cache()
        extra = []
        rows = [("config 3a", "fold", 3, "zipf24s1.2", n), ("config 3b", "rfold", 3, "zipf24s1.2", n),
                ("config 1 data shape", "fold", 1, "uniform1-256", n),
                ("ANSfold-5 on config 2 data (table_efficiency.cpp:176-179 runs fidelities 1 and 5)", "fold", 5, args.dist, n)]
        for label, cn, f, spec, m in rows:
            try:
                r = run_single(torch, A, ctx, device, cn, f, spec, m, 3, 1,
                               cpu_sample=0 if args.no_cpu else 16 * (1 << 20))
                r["baseline_config"] = label
                extra.append(r)
            except Exception as exc:  # noqa: BLE001
                extra.append({"baseline_config": label, "error": repr(exc)})
            torch.cuda.empty_cache()
        try:
            ranks, src = bwtmtf_ranks()
            if ranks is None:
                extra.append({"baseline_config": "config 5 fallback", "error": src})
            else:
                d5 = torch.from_numpy(ranks.view("int32")).to(device)
                r = run_single(torch, A, ctx, device, "fold", 1, "bwtmtf-pylib", ranks.size, 5, 2, d_in=d5,
                               cpu_sample=0 if args.no_cpu else ranks.size)
                r["baseline_config"] = "config 5 fallback (the real datasets are not in the image)"
                r["source"] = src
                extra.append(r)
        except Exception as exc:  # noqa: BLE001
            extra.append({"baseline_config": "config 5 fallback", "error": repr(exc)})

    # ---- speed against size: block length / restart interval (the two options that decide the operating point)
    frontier = config1 = compaction = None
    if rank == 0 and world == 1 and not args.no_extra and not args.no_frontier:
        frontier = []
        for blk_, ck_ in ((16384, 512), (16384, 1024), (16384, 2048), (32768, 1024)):
            try:
                r = run_single(torch, A, ctx, device, args.codec, args.fidelity, args.dist, n, 3, 2, block=blk_, ckpt=ck_, profile=False)
                frontier.append({"block_ints": blk_, "ckpt_interval": ck_, "value": r["value"], "unit": "Mints/s",
                                 "bits_per_int": r["bits_per_int"], "enc_mints": r["gpu_rates"]["enc_mints"],
                                 "dec_mints": r["gpu_rates"]["dec_mints"], "roundtrip_ok": r["roundtrip_ok"]})
            except Exception as exc:  # noqa: BLE001
                frontier.append({"block_ints": blk_, "ckpt_interval": ck_, "error": repr(exc)})
            torch.cuda.empty_cache()
        try:
            config1 = config1_rows(A)
        except Exception as exc:  # noqa: BLE001
            config1 = {"error": repr(exc)}
        # where the per-block alphabet compaction (src/pseudo_adaptive.cpp:85-130) pays and where it does not
        try:
            compaction = compaction_rows(torch, A, ctx, device)
        except Exception as exc:  # noqa: BLE001
            compaction = {"error": repr(exc)}

    multi_gpu = "single GPU"
    if world > 1:
        # what the gather costs by arithmetic, so that a scaling record explains itself: every step moves
        # (N - 1) containers into one root, each over its sender's own xGMI link (one hop)
        LINK_GBS = 50.0  # assumed sustained one-direction rate of one xGMI link (peak ~64 GB/s; MI355X_MICROARCH.md: 7 links/GPU)
        x = state["xfer"]
        link_ms = x / (LINK_GBS * 1e9) * 1e3
        in_flight = DEPTH - 1
        multi_gpu = {
            "split": "contiguous block ranges per rank (rank r draws and encodes ints [r n, (r+1) n) of one list)",
            "gather": "per step every rank sends its container (fixed %d-byte transfers, RCCL send/recv) to the step's root, which "
                      "merges them into one container with ansx_merge_containers_dev inside the timed step; the merged container "
                      "of the last step is decoded on its root and checked against the generator" % x,
            "root_policy": root_policy, "transfers_in_flight": in_flight,
            "gather_bytes_per_step": (world - 1) * x, "gather_bytes_per_link_per_step": x,
            "assumed_link_GBps": LINK_GBS, "predicted_link_ms": link_ms,
            "predicted_note": ("fixed root: the root's N - 1 inbound links each carry one container per step, so a step cannot be "
                               "shorter than max(compute, link_ms); rotating root: consecutive steps use different roots and %d "
                               "transfers overlap, so the bound is max(compute, link_ms / %d)" % (in_flight, max(in_flight, 1))),
            "predicted_step_ms_bound": max(link_ms / max(in_flight, 1) if rotate else link_ms, 0.0),
        }
    if rank == 0:
        def brief(r):
            """[Gints/s, HBM-roofline fraction of the dominant kernel] of one configuration row"""
            if not r or "error" in r:
                return None
            rf = r.get("roofline") or {}
            return [round(r["value"] / 1e3, 1), round(rf["frac"], 3) if rf.get("frac") is not None else None]

        by_label = {}
        for r in extra or []:
            by_label[r.get("baseline_config", "")[:9]] = r
        summary = {"cfg2": [round(value / 1e3, 1), round(roofline["frac"], 3) if roofline else None],
                   "cfg3a": brief(by_label.get("config 3a")), "cfg3b": brief(by_label.get("config 3b")),
                   "cfg1shape": brief(by_label.get("config 1 ")), "cfg5fb": brief(by_label.get("config 5 ")),
                   "fold5": brief(by_label.get("ANSfold-5")),
                   "single_stream_mints": round(single_stream["value"], 1) if single_stream and "value" in single_stream else None,
                   "first_call_ms": round(first_call["first_call_ms"], 2) if first_call and "first_call_ms" in first_call else None,
                   "enc_dec_gints": [round(gpu_rates["enc_mints"] / 1e3, 1), round(gpu_rates["dec_mints"] / 1e3, 1)] if gpu_rates else None,
                   "unit": "[Gints/s encode+decode, roofline.frac of that config's dominant kernel]", "all_roundtrips_ok": None}
        rts = [ok] + [r.get("roundtrip_ok") for r in (extra or []) if "error" not in r]
        if single_stream and "roundtrip_ok" in single_stream:
            rts.append(single_stream["roundtrip_ok"])
        summary["all_roundtrips_ok"] = all(bool(x) for x in rts) and not any("error" in r for r in (extra or []))
        # Everything bulky (per-kernel tables, the extra configurations in full, the frontier, the Table-10 rows, the
        # compaction rows) goes to a side file and to stderr; the ONE stdout line stays short enough that a reader of its
        # tail sees the whole of it, and ends with `summary`.
        details = {"kernels": kernels, "extra_configs": extra, "frontier": frontier, "config1_table10": config1,
                   "alphabet_compaction": compaction, "cpu_all_cores": cpu_all, "host_buffer_api": host_api,
                   "single_stream": single_stream, "first_call": first_call, "gpu_rates": gpu_rates, "cpu_baseline_full": cpu,
                   "roofline_full": roofline, "workspace_mb": ctx.workspace_bytes() / 1e6}
        details_path = None
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            details_path = os.path.join("gpurun_out", "bench_details.json")
            with open(os.path.join(ROOT, details_path), "w") as fh:
                json.dump(details, fh, indent=1)
        except OSError:
            details_path = None
        print(json.dumps({"bench_details": details}), file=sys.stderr)
        if roofline:  # (the sources are sentences: the side file has them)
            roofline = {k: v for k, v in roofline.items() if k not in ("traffic_source", "issue_floor_source")}
        line = {
            "metric": "encode+decode Mints/s (uint32), bit-exact vs CPU reference per block",
            "value": value, "unit": "Mints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 symbols; ANS state carried as an exact f64 (integers below 2^53) in the encoder, u64 in the decoder",
            "data": "synthetic",
            "config": {"workload": "ANS%s-%d on %d uint32 per GPU, %s, blocks of %d ints, restart every %d"
                                   % (args.codec, args.fidelity, n, args.dist, block_ints,
                                      args.ckpt or A.DEFAULT_CKPT_INTERVAL),
                       "codec": codec.name(), "ints_per_gpu": n, "distribution": args.dist, "block_ints": block_ints,
                       "ckpt_interval": args.ckpt or A.DEFAULT_CKPT_INTERVAL, "seed": SEED, "multi_gpu": multi_gpu},
            "roundtrip_ok": ok, "merged_container_ok": merged_ok, "bits_per_int": 8 * c_bytes,
            "near_threshold_decisions": stats["near_threshold_decisions"], "encode_path": stats["path"],
            "roofline": roofline,
            "cpu_baseline": ({k: v for k, v in cpu.items() if k in ("value", "unit", "cores", "kind", "sample", "cpu", "enc_mints", "dec_mints", "bits_per_int")} if cpu else None),
            "details": details_path,
        }
        if cpu:
            line["speedup_vs_cpu_1thread"] = value / cpu["value"]
        line["summary"] = summary
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("round trip mismatch")


if __name__ == "__main__":
    main()
