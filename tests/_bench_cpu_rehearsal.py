"""CPU rehearsal of `python bench.py --gpus N` (tests/test_bench_plumbing.py): bench.main() as it is — argument parsing, the
self-launch of the ranks and their supervision, the watchdog, the JSON — with run() given the oracle library behind the
same C ABI, CPU tensors and gloo.  The substitution lives HERE, in the test tree; bench.py has no switch for it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from tests._oracle import load_oracle  # noqa: E402

_run = bench.run
bench.run = lambda args, backend="nccl": _run(args, lib=load_oracle(), dev=torch.device("cpu"), backend="gloo")
bench.main()
