"""bench.py's control flow for N > 1 (sharded stepping, one all-to-all per tick, convergence section,
the JSON contract) driven on CPU: two gloo ranks, CPU tensors and — injected by THIS test, bench.py itself
never touches it outside its cpu_baseline leg — the oracle library behind the same C ABI.  It guards the
driver's multi-GPU run, which cannot be rehearsed here (one GPU per gpurun call)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, extra=()):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    import bench
    from tests._oracle import load_oracle

    try:
        args = bench.parse_args(["--gpus", str(world), "--steps", "20", "--warmup", "10", "--nodes-per-gpu", "2048",
                                 "--view-slots", "64", "--ring", "32", "--no-cpu-baseline", "--allow-drops", *extra])
        out = bench.run(args, lib=load_oracle(), dev=torch.device("cpu"), backend="gloo")
        q.put((rank, json.dumps(out) if out is not None else "null"))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, "ERR " + repr(e)))
        raise


@pytest.mark.parametrize("world", [1, 2])
def test_bench_control_flow(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from tests._scenario import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    res = dict(q.get(timeout=5) for _ in procs)
    assert not any(str(v).startswith("ERR") for v in res.values()), res
    out = json.loads(res[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "rounds_to_99"):
        assert key in out, key
    assert out["n_gpus"] == world and out["steps"] == 20 and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["value"] > 0 and out["config"]["workload"].startswith(f"{2048 * world} nodes")
    r99 = out["rounds_to_99"]
    assert 32 <= r99["n"] <= 64 and 1 <= r99["median"] <= 60 and r99["p90"] >= r99["median"] and sum(r99["histogram"].values()) == r99["n"]
    assert r99["window_ticks"][0] == 320 + 400, "the convergence window starts at a fixed tick, whatever --steps / --warmup are"
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    if world > 1:
        assert all(res[r] == "null" for r in range(1, world))
    # the ONE line the driver parses: short enough that an 8 KB tail holds all of it (round 5's 21 KB line lost its head)
    sys.path.insert(0, ROOT)
    import bench
    line = bench.compact(out)
    txt = json.dumps(line)
    assert len(txt) < bench.LINE_LIMIT < 6000, len(txt)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "rounds_to_99", "detail_file"):
        assert key in line, key
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and len(line["config"]["workload"]) <= 300
    assert line["value"] == pytest.approx(out["value"], rel=1e-4)


def test_bench_control_flow_random_fanout_on_two_ranks():
    # (r4, r5) the N > 1 line on memberlist's kRandomNodes: ShardedSim's all-to-all of the packed slabs, the JSON's exchange section
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from tests._scenario import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, ("--fanout-model", "krandomnodes"))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    res = dict(q.get(timeout=5) for _ in procs)
    assert not any(str(v).startswith("ERR") for v in res.values()), res
    out = json.loads(res[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["fanout_model"] == "krandomnodes"
    assert out["exchange"]["collective"].startswith("equal-split all-to-all of packed slabs") and out["exchange"]["chunks"] == 2
    x = out["exchange"]
    # what leaves a rank: (V - 1) / V of its slabs; the slabs are the packets (64-byte cells) plus a few per cent (12 sigma of room,
    # a count byte per target, headers) — not the O(N) per shard of round 4's all-gather
    assert x["bytes_arriving_per_gpu_per_tick"] == x["bytes_leaving_gpu_per_tick"] == x["bytes_per_gpu_per_tick"] // 2
    assert x["bytes_per_gpu_per_tick"] < 1.4 * x["packet_bytes_per_gpu_per_tick"]   # (2 048 nodes per rank: 12 sigma is 9 % here; the oracle's slab carries 8 index bytes per packet)
    assert 1 <= out["rounds_to_99"]["median"] <= 60


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts and supervises its ranks itself.  The
    rehearsal script runs bench.main() unchanged — argument parsing, self_launch, the watchdog — with run() handed the
    oracle library, CPU tensors and gloo (this test's doing: bench.py has no such switch)."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "tests", "_bench_cpu_rehearsal.py"), "--gpus", "2", "--steps", "10", "--warmup", "5",
           "--preroll", "40", "--nodes-per-gpu", "2048", "--view-slots", "64", "--ring", "32", "--no-cpu-baseline", "--no-convergence",
           "--allow-drops", "--backend", "gloo"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "error" not in out
    assert out["distributed"]["world_size"] == 2 and out["distributed"]["backend"] == "gloo"
    # (r5) at every N the headline is the reference's kRandomNodes (one exchange of packed slabs per round), the bijection — its
    # all-to-all issued chunk-wise — next to it: the 1 -> 8 GPU series is ONE model
    assert out["config"]["fanout_model"] == "krandomnodes" and out["exchange"]["chunks"] == 2
    assert out["fanout_models"]["bijection"]["exchange"]["chunks"] == 2 and out["fanout_models"]["bijection"]["value"] > 0


def test_bench_reports_a_dead_rank_instead_of_hanging():
    """A rank that dies (here: an impossible --chunks) takes the run down with an "error" line, after one retry."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "tests", "_bench_cpu_rehearsal.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--preroll", "8", "--nodes-per-gpu", "2048", "--view-slots", "64", "--ring", "32", "--no-cpu-baseline", "--no-convergence",
           "--allow-drops", "--backend", "gloo", "--fanout", "9"]   # fan-out 9 does not exist: sim_create fails on every rank
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "error" in json.loads(lines[0]) and json.loads(lines[0])["value"] is None
