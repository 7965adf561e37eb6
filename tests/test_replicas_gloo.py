"""N > 1 path of bench.py on CPU: two gloo processes stand in for two GPU replicas.  The path does not
shard, so the only exchange is the max-over-ranks of the timings."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # each replica does its own (CPU-branch) forward; replicas never exchange data
    from chatglm_q_amd.int8.qlinear import DynamicQuantizeLinear
    torch.manual_seed(rank)
    layer = DynamicQuantizeLinear(64, 32, bias=False, dtype=torch.float32)
    layer.apply_weights_(torch.randint(-127, 128, (32, 64), dtype=torch.int8), torch.rand(32) * 0.01)
    y = layer(torch.randn(1, 64))
    wall = 1.0 + rank            # rank 1 is the slow replica
    ev_ms = 900.0 + 100 * rank
    dist.barrier()
    got = bench.max_over_ranks([wall, ev_ms], dist, torch.device("cpu"))
    q.put((rank, got, float(y.abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_max_over_ranks_and_aggregate_value():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, got, _ in res:
        assert got == [2.0, 1000.0]                   # both ranks agree on the slowest replica's time
    assert res[0][2] != res[1][2]                     # replicas really ran independent work
    steps, bytes_per_step = 100, bench.alg_bytes_w4(1, 4096, 4096)
    assert bytes_per_step == 9453568
    v1 = bench.whole_job_gbps(1, steps, bytes_per_step, 1.0)
    v2 = bench.whole_job_gbps(2, steps, bytes_per_step, 1.0)
    assert abs(v2 - 2 * v1) < 1e-9                    # weak scaling: N replicas, N x the units


def test_single_process_passthrough():
    assert bench.max_over_ranks([0.5, 3.0], None, None) == [0.5, 3.0]


def _last_json_line(text):
    import json
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text                      # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_n_spawns_n_replicas_itself():
    """`python bench.py --gpus 2` with no launcher around it (--gpus was parsed and never read): two worker processes,
    one rendezvous, rank 0 prints the one line with n_gpus 2 and one card per rank.  Dry mode (gloo, CPU, no kernel) - what can
    run here; the GPU box runs the same path with nccl and two devices; test_bench_refuses_more_gpus_than_visible checks the refusal."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.abspath(bench.__file__))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    line = _last_json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 7 and line["dry_run"] is True and line["value"] is None
    assert line["scaling"] == "weak" and line["config"]["parallelism"] == "replicas"
    assert [c["rank"] for c in line["ranks"]] == [0, 1] and line["ranks"][0]["pid"] != line["ranks"][1]["pid"]
    assert line["max_over_ranks_is_slowest_rank"] is True


def test_bench_under_the_drivers_launcher():
    """The driver's form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 (RANK / WORLD_SIZE from the env)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.abspath(bench.__file__))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                          "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _last_json_line(out.stdout)
    assert line["n_gpus"] == 2 and len(line["ranks"]) == 2


def test_bench_refuses_more_gpus_than_visible():
    import subprocess
    import sys
    root = os.path.dirname(os.path.abspath(bench.__file__))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 1
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(2, n)), "--steps", "3"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2 and "visible" in out.stderr and not out.stdout.strip()
