#!/usr/bin/env python3
"""Benchmark of the quantized-linear forward hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.

  step      one QLinear forward over one batch: int4g32, 1 x 4096 -> 4096, fp16 (BASELINE.json
            configs[1], the configuration the metric is quoted on).  Every step uses a DIFFERENT weight
            set out of a rotation larger than the 256 MB Infinity Cache, so weights really come from HBM.
  value     whole-job algorithmic GB/s = n_gpus * steps * bytes_per_step / wall time of the timed region
            (inputs resident in HBM; barrier + synchronize on both sides; max over ranks).
  roofline  algorithmic bytes per launch / average per-launch time from HIP events recorded on the
            launch stream around the same timed region, against the 8 TB/s HBM3E peak.
  cpu_baseline  the C oracle (oracle/liboracle.so) timed on this box's host cores on a bounded sample.

The timed steps are replayed from ONE captured HIP graph (K kernel nodes, strictly sequential on one
stream) so that the figure is not the Python/ctypes launch overhead; `--launch eager` times plain
launches instead.  Multi-GPU: the path does not shard (DESIGN.md "replicas only"): `--gpus N` runs N
independent replicas, one process per GPU, no data-path collective; scaling is "weak".
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)
K_DIM = 4096
N_DIM = 4096
GROUP = 32


def alg_bytes_w4(M, N, K, esize=2, bias=False):
    """K*N/2 + (K/32)*N*s + M*K*s + M*N*s (+ N*s)."""
    return K * N // 2 + (K // GROUP) * N * esize + M * K * esize + M * N * esize + (N * esize if bias else 0)


def cpu_baseline(budget_s: float = 12.0):
    """Time the C oracle (dense dequant + matmul, the reference's CPU formula
    chatglm_q/int4/qlinear.py:20-33,50) on the host cores, bounded sample."""
    import numpy as np
    lib_path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib_path):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(lib_path)
    lib.oracle_w4_fwd.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 4 + [ctypes.c_int]
    rng = np.random.default_rng(0)
    qw = rng.integers(0, 256, (K_DIM // 2, N_DIM), dtype=np.uint8)
    sc = (rng.random((K_DIM // GROUP, N_DIM)) * 0.02 + 0.002).astype(np.float16)
    a = rng.standard_normal((1, K_DIM)).astype(np.float16)
    c = np.zeros((1, N_DIM), dtype=np.float16)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    times = []
    t_end = time.perf_counter() + budget_s
    lib.oracle_w4_fwd(p(a), p(qw), p(sc), None, p(c), 1, N_DIM, K_DIM, GROUP, 1)      # warm-up
    while len(times) < 5 or (time.perf_counter() < t_end and len(times) < 200):
        t0 = time.perf_counter()
        lib.oracle_w4_fwd(p(a), p(qw), p(sc), None, p(c), 1, N_DIM, K_DIM, GROUP, 1)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {
        "value": round(alg_bytes_w4(1, N_DIM, K_DIM) / med / 1e9, 4),
        "unit": "GB/s",
        "cores": int(lib.oracle_num_threads()),
        "kind": "port",
        "sample": f"{len(times)} forwards of int4g32 1x{K_DIM}->{N_DIM} fp16 through oracle/liboracle.so "
                  f"(OpenMP dense dequant + matmul), median {med * 1e3:.2f} ms",
    }


def pmc_traffic():
    """HBM bytes per launch of the headline kernel from the committed rocprofv3 PMC passes
    (profiles/rNN_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 fetch correction)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None, None
    try:
        pmc = json.load(open(files[-1]))["pmc"]
        return int(pmc["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except (KeyError, ValueError, OSError):
        return None, None


def max_over_ranks(values, dist, device):
    """Replica aggregation used by the N > 1 run: every rank contributes its own timings, the job's time is
    the slowest rank's (all_reduce MAX).  No data-path collective exists - this is the only exchange."""
    import torch
    if dist is None:
        return [float(v) for v in values]
    t = torch.tensor(list(values), device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def whole_job_gbps(world, steps, bytes_per_step, wall_s):
    """Aggregate throughput of N independent replicas that each ran `steps` steps in `wall_s` (max over ranks)."""
    return world * steps * bytes_per_step / wall_s / 1e9


def make_layers(torch, n_sets, device, bias=False):
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    layers = []
    g = torch.Generator(device=device).manual_seed(1234)
    for _ in range(n_sets):
        layer = DynamicQuantizeLinear(K_DIM, N_DIM, bias=bias, dtype=torch.float16, device=device)
        # synthetic weights of the int4g32 format: uniform nibbles, fp16 scales of realistic size
        layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=device, generator=g))
        layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=device, generator=g) * 0.02 + 0.002).half())
        layer.prepare()
        layers.append(layer)
    return layers


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1440)
    ap.add_argument("--warmup", type=int, default=144)
    ap.add_argument("--sets", type=int, default=72, help="distinct weight sets in the rotation (72 x 9.4 MB = 680 MB)")
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph")
    ap.add_argument("--layout", choices=["auto", "canonical"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the whole-token sweep / other shapes")
    args = ap.parse_args()

    if args.layout == "canonical":
        os.environ["QLINEAR_W4_LAYOUT"] = "canonical"

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback for device work")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from chatglm_q_amd import _lib
    _lib.get_lib()  # fail loudly if the HIP extension is missing

    layers = make_layers(torch, args.sets, device)
    x = torch.randn(1, K_DIM, device=device, dtype=torch.float16)
    bytes_per_step = alg_bytes_w4(1, N_DIM, K_DIM)

    def run_steps(n, offset=0):
        out = None
        with torch.no_grad():
            for i in range(n):
                out = layers[(offset + i) % len(layers)](x)
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (untimed) -------------------------------------------------------------------
    run_steps(args.warmup)
    torch.cuda.synchronize()

    stream = torch.cuda.Stream(device=device)
    graph = None
    launch_mode = args.launch
    with torch.cuda.stream(stream):
        if launch_mode == "graph":
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    run_steps(args.steps)
                graph.replay()                      # one untimed replay
                stream.synchronize()
            except Exception as e:                  # pragma: no cover - capture unsupported
                print(f"[bench] graph capture failed ({e}); falling back to eager", file=sys.stderr)
                graph = None
                launch_mode = "eager"
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        launches_before = _lib.launch_count()
        barrier()
        t0 = time.perf_counter()
        ev0.record(stream)
        if graph is not None:
            graph.replay()
        else:
            run_steps(args.steps)
        ev1.record(stream)
        stream.synchronize()
        barrier()
        t1 = time.perf_counter()
    wall_s = t1 - t0
    ev_ms = ev0.elapsed_time(ev1)
    wall_s, ev_ms = max_over_ranks([wall_s, ev_ms], dist, device)
    if graph is None:
        assert _lib.launch_count() - launches_before >= args.steps, "steps did not go through the HIP library"

    ms_per_step = wall_s * 1e3 / args.steps
    us_per_launch_ev = ev_ms * 1e3 / args.steps
    value = whole_job_gbps(world, args.steps, bytes_per_step, wall_s)
    achieved = bytes_per_step / (us_per_launch_ev * 1e-6) / 1e9

    traffic, traffic_src = pmc_traffic()
    result = {
        "metric": "QLinear fwd GB/s + tok/s ChatGLM2-6B int4g32 decode, 1xMI355X",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 6),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {
            "workload": "int4g32 QLinear forward 1x4096->4096 (decode shape), fp16 activations",
            "M": 1, "K": K_DIM, "N": N_DIM, "group": GROUP,
            "weight_sets_rotated": args.sets,
            "rotation_bytes": args.sets * bytes_per_step,
            "launch": launch_mode,
            "layout": "derived (column-major) cache of the canonical buffers" if args.layout == "auto" else "canonical",
            "parallelism": "replicas" if world > 1 else "single",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": bytes_per_step,
            "us_per_launch_hip_events": round(us_per_launch_ev, 4),
            "note": "per-launch time = HIP-event time of the timed region / steps (strictly sequential "
                    "launches, so it contains the ~1.5 us dependent-launch boundary that an EMPTY kernel also "
                    "pays on this chip; a pure streaming read of the same 9.4 MB measures 3.5 us per launch - "
                    "see DESIGN.md 'Measured ceilings')",
        },
    }

    if rank == 0 and world == 1 and not args.no_extras:
        try:
            import bench_extras
            result["extras"] = bench_extras.run(torch, device)
        except Exception as e:      # extras never invalidate the headline
            result["extras"] = {"error": repr(e)}
    if rank == 0 and isinstance(result.get("extras"), dict):
        e2e = result["extras"].get("e2e_generate", {})
        if isinstance(e2e.get("graph_sync_every_token"), dict):
            # second half of BASELINE.json's metric: ChatGLM2-6B int4g32 decode tok/s (reference timing definition)
            result["decode_tok_per_s"] = e2e["graph_sync_every_token"]["gen_tok_per_s"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
