"""bench.py's control flow for N > 1 (sharded stepping, one all-to-all per tick, convergence section,
the JSON contract) driven on CPU: two gloo ranks, CPU tensors and — injected by THIS test, bench.py itself
never touches it outside its cpu_baseline leg — the oracle library behind the same C ABI.  It guards the
driver's multi-GPU run, which cannot be rehearsed here (one GPU per gpurun call)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    import bench
    from tests._oracle import load_oracle

    try:
        args = bench.parse_args(["--gpus", str(world), "--steps", "20", "--warmup", "10", "--nodes-per-gpu", "2048",
                                 "--view-slots", "64", "--ring", "32", "--no-cpu-baseline", "--allow-drops"])
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
    port = 29900 + (os.getpid() % 200) + world
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
    assert out["rounds_to_99"]["n"] == 8 and 1 <= out["rounds_to_99"]["median"] <= 60
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    if world > 1:
        assert all(res[r] == "null" for r in range(1, world))
