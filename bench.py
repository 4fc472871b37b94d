#!/usr/bin/env python3
"""bench.py — ANSfold/ANSrfold encode+decode throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): ANSfold-1 on 256 Mi uint32 drawn from Zipf(s=1.2) over
{1..2^20}, HBM-resident, per GPU (weak scaling; config 4's 2e9 ints over 8 GPUs is the same
per-GPU size).  One step = one encode (histogram -> normalise -> prelude -> 4-state rANS -> container)
plus one decode of the whole batch; with N > 1 every rank also ships its compressed container to
rank 0 over RCCL (send/recv, xGMI), overlapped with its decode.  value = total ints of all ranks
/ step time, in Mints/s.  Output is bit-exact per block against the reference CPU encoder
(tests/test_gpu_parity.py); the round trip is verified here on every run.

Extra objects on the JSON line:
  roofline     dominant kernel: algorithmic bytes (encode 4+c, decode c+4 bytes/int)
               / its average launch duration (hipEvents inside libansx on the launch stream),
               against the 8 TB/s HBM peak
  cpu_baseline the reference itself (oracle/_ref, "reference") or the C restatement ("port"),
               single thread, on a bounded sample of the same data, rank 0 at N = 1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def gen_zipf(torch, n, log2_sigma, s, seed, device):
    """Zipf(s) over {1..2^log2_sigma} by inverse-CDF sampling on the GPU (synthetic input;
    torch is plumbing here, not the product)."""
    N = 1 << log2_sigma
    w = 1.0 / torch.arange(1, N + 1, dtype=torch.float64, device=device) ** s
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty(n, dtype=torch.int32, device=device)
    chunk = 1 << 25
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        u = torch.rand(m, generator=g, device=device, dtype=torch.float64)
        out[a:a + m] = (torch.searchsorted(cdf, u) + 1).clamp_(max=N).to(torch.int32)
    return out


def gen_uniform(torch, n, lo, hi, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randint(lo, hi + 1, (n,), generator=g, device=device, dtype=torch.int32)


def cpu_baseline(sample, kind, f, block_ints=16384):
    """Reference CPU path, one thread, whole-list encode()+decode() as table_efficiency.cpp
    times it (min over runs).  Uses oracle/ strictly as the measured CPU comparator."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol

    n = sample.size
    use_ref = ol.have_ref()
    runs = 3
    t_enc, t_dec = 1e30, 1e30
    stream = None
    for _ in range(runs):
        t0 = time.perf_counter()
        if use_ref:
            stream = ol.ref_encode(kind, f, sample)
        else:
            stream = ol.oracle_encode(kind, f, sample)[0]
        t_enc = min(t_enc, time.perf_counter() - t0)
    for _ in range(runs):
        t0 = time.perf_counter()
        back = ol.ref_decode(kind, f, stream, n) if use_ref else ol.oracle_decode(kind, f, stream, n)
        t_dec = min(t_dec, time.perf_counter() - t0)
    ok = bool(np.array_equal(back, sample))
    # like-for-like row: the same data encoded/decoded block by block (one reference encode()
    # per block_ints ints, exactly the units the GPU processes), on a slice of the sample
    blk_n = min(n, 8 * (1 << 20))
    tb_enc = tb_dec = 0.0
    streams = []
    for a in range(0, blk_n, block_ints):
        part = np.ascontiguousarray(sample[a:a + block_ints])
        t0 = time.perf_counter()
        s_ = ol.ref_encode(kind, f, part) if use_ref else ol.oracle_encode(kind, f, part)[0]
        tb_enc += time.perf_counter() - t0
        streams.append((s_, part.size))
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
        "sample": "first %d ints of the workload, one whole-list encode()+decode(), min of %d runs" % (n, runs),
        "enc_mints": n / t_enc / 1e6, "dec_mints": n / t_dec / 1e6,
        "bits_per_int": 8.0 * stream.size / n, "roundtrip_ok": ok,
        "cpu": _cpu_model(),
    }


def pmc_traffic(kernel, args, n):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/*_hbm_traffic_pmc.json: FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, separate
    passes, gfx950 correction calibrated on k_fold_hist).  Counters cannot be read from inside
    this process, so the figure is only reported when the workload is the profiled one."""
    try:
        import glob

        best = None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic_pmc.json"))):
            with open(path) as fh:
                best = json.load(fh)
        if best is None:
            return None
        want = "ANS%s-%d, %d ints, %s, block %d, ckpt %d" % (args.codec, args.fidelity, n, args.dist,
                                                              args.block or 16384, args.ckpt or 1024)
        if best.get("workload") != want:
            return None
        for name, v in best["kernels"].items():
            if name.split("<")[0] == kernel:
                return v["hbm_bytes"]
    except Exception:
        return None
    return None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--ints", dest="n", type=int, default=256 * (1 << 20), help="ints per GPU")
    ap.add_argument("--codec", default="fold", choices=["fold", "rfold"])
    ap.add_argument("--fidelity", type=int, default=1)
    ap.add_argument("--dist", default="zipf20s1.2", help="zipf<log2 sigma>s<exponent> | uniform256")
    ap.add_argument("--block", type=int, default=0, help="ints per block (0 = library default)")
    ap.add_argument("--ckpt", type=int, default=0, help="restart interval (0 = library default)")
    ap.add_argument("--cpu-sample", type=int, default=64 * (1 << 20))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--gather-root", default="rotate", choices=["rotate", "fixed"],
                    help="N > 1: root of the per-step container gather: rank k mod N of step k (default; "
                         "spreads the traffic over all xGMI links) or always rank 0")
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

    import ans_large_alphabet_amd as A

    ctx = A.Context(local_rank)
    cls = A.ANSfold if args.codec == "fold" else A.ANSrfold
    codec = cls(args.fidelity, ctx=ctx, block_ints=args.block, ckpt_interval=args.ckpt)
    kind = A.FOLD if args.codec == "fold" else A.RFOLD
    n = args.n

    # ---- synthetic input, resident in HBM
    if args.dist.startswith("zipf"):
        lg, s = args.dist[4:].split("s")
        d_in = gen_zipf(torch, n, int(lg), float(s), 1234 + rank, device)
    elif args.dist == "uniform256":
        d_in = gen_uniform(torch, n, 1, 256, 1234 + rank, device)
    else:
        raise SystemExit("unknown --dist")
    cap = min(codec.bound(n), 8 * n + (64 << 20))
    # N > 1.  Every step ends with the rank containers concatenated on ONE GPU (RCCL send/recv: each
    # sender uses its direct xGMI link to the root).  A rank's container is ~0.29 GB, i.e. several ms
    # on one link -- longer than the step's compute -- so (1) the transfer of step k overlaps the
    # decode of step k and the following encodes (DEPTH container buffers; a buffer is reused only
    # after the transfer that reads it has been waited for), (2) the root rotates (rank k mod N), so
    # consecutive steps use different links / directions instead of funnelling everything into rank
    # 0's seven links, and (3) each pipeline slot has its own communicator, because groups issued on
    # one communicator execute back to back on its stream.
    rotate = world > 1 and args.gather_root == "rotate"
    DEPTH = 1 if world == 1 else (min(4, max(2, world)) if rotate else 2)
    outs = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(DEPTH)]
    d_out = outs[0]
    d_back = torch.zeros(n, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream

    from ans_large_alphabet_amd import dist as adist
    import collections

    groups = [None] * DEPTH
    if dist is not None and rotate:
        try:
            groups = [dist.new_group(ranks=list(range(world))) for _ in range(DEPTH)]
            # establish every (communicator, root) connection before anything is timed
            tiny = torch.zeros(64, dtype=torch.uint8, device="cpu" if args.rehearse_gloo else device)
            for g_ in groups:
                for d_ in range(world):
                    adist.gather_containers(tiny, 64, dst=d_, group=g_, async_op=False)
            if not args.rehearse_gloo:
                torch.cuda.synchronize()
        except Exception as exc:  # same code on every rank: a setup failure is symmetric
            if rank == 0:
                print("bench.py: rotating-root setup failed (%r); using the fixed root" % (exc,), file=sys.stderr)
            rotate = False
            DEPTH = 2
            outs = outs[:2]
            groups = [None] * DEPTH

    state = {"k": 0}
    pending = collections.deque()  # (works, receive buffer kept alive) of the steps still in flight

    def drain_one():
        works, _keep = pending.popleft()
        for w in works:
            w.wait()

    def step():
        k = state["k"]
        state["k"] += 1
        slot = k % DEPTH
        out = outs[slot]
        nb = codec.encode_dev(d_in.data_ptr(), n, out.data_ptr(), cap, stream=stream)
        works, buf = [], None
        if dist is not None:
            src = out[:nb].cpu() if args.rehearse_gloo else out
            buf, _sizes, works = adist.gather_containers(src, nb, dst=(k % world) if rotate else 0,
                                                         group=groups[slot], async_op=True)
        codec.decode_dev(out.data_ptr(), nb, d_back.data_ptr(), n, stream=stream)
        pending.append((works, buf))
        while len(pending) > DEPTH - 1:  # the next user of a buffer must find its transfer finished
            drain_one()
        return nb

    def drain():
        while pending:
            drain_one()

    def sync_all():
        drain()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    nb = 0
    for _ in range(args.warmup):
        nb = step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nb = step()
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_gloo else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ok = bool(torch.equal(d_back, d_in))
    ms_per_step = dt / args.steps * 1e3
    value = world * n * args.steps / dt / 1e6
    c_bytes = nb / n  # compressed bytes per int, everything included

    # ---- per-kernel timing pass (separate from the timed region) -> roofline of the dominant kernel
    roofline = None
    kernels = None
    if not args.no_profile:
        ctx.profile(True)
        ctx.profile_reset()
        reps = 3
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
        # algorithmic bytes per launch: the encoder reads 4 B/int and writes c,
        # the decoder reads c and writes 4; other kernels are priced by what they must touch
        alg = {"k_encode": (4 + c_bytes) * n, "k_encode_gtab": (4 + c_bytes) * n, "k_decode": (c_bytes + 4) * n, "k_decode_gtab": (c_bytes + 4) * n, "k_decode_table": (c_bytes + 4) * n,
               "k_fold_hist": 4.0 * n, "k_compact": 2 * c_bytes * n, "k_rfold_remap": 8.0 * n}.get(dom, 4.0 * n)
        achieved = alg / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, args, n),
                    "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        m = min(n, args.cpu_sample)
        sample = d_in[:m].cpu().numpy().view("uint32")
        cpu = cpu_baseline(sample, kind, args.fidelity, args.block or A.DEFAULT_BLOCK_INTS)

    if rank == 0:
        line = {
            "metric": "encode+decode Mints/s (uint32), bit-exact vs CPU reference per block",
            "value": value, "unit": "Mints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 symbols / u64 ANS states (f64 only in model normalisation)",
            "data": "synthetic",
            "config": {"workload": "ANS%s-%d on %d uint32 per GPU, %s, blocks of %d ints, restart every %d"
                                   % (args.codec, args.fidelity, n, args.dist,
                                      args.block or A.DEFAULT_BLOCK_INTS, args.ckpt or A.DEFAULT_CKPT_INTERVAL),
                       "ints_per_gpu": n, "distribution": args.dist, "codec": codec.name(),
                       "block_ints": args.block or A.DEFAULT_BLOCK_INTS,
                       "ckpt_interval": args.ckpt or A.DEFAULT_CKPT_INTERVAL,
                       "multi_gpu": ("contiguous block ranges per rank; per-step RCCL send/recv gather of the rank containers to "
                                     + ("a root that rotates per step (k mod N), %d transfers in flight on %d communicators" % (DEPTH - 1, DEPTH)
                                        if rotate else "rank 0, overlapped with the next step")) if world > 1 else "single GPU"},
            "roundtrip_ok": ok, "compressed_bytes_per_int": c_bytes, "bits_per_int": 8 * c_bytes,
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
            "workspace_mb": ctx.workspace_bytes() / 1e6,
        }
        if cpu:
            line["speedup_vs_cpu_1thread"] = value / cpu["value"]
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("round trip mismatch")


if __name__ == "__main__":
    main()
