"""Sharding rules + the N>1 launch path on CPU (gloo, world_size 2, real processes)."""
import os
import subprocess
import sys

import pytest

from consolver_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(5000, 8), (128, 8), (7, 8), (1001, 3), (16, 1), (0, 4)])
def test_shard_rules_match_reference(n, world):
    spans = [launch.shard_bounds(n, world, r) for r in range(world)]           # gen_ppo.py:349-357
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert all(e - s == n // world for s, e in spans[:-1])                      # last rank takes the remainder
    spans = [launch.shard_bounds_ceil(n, world, r) for r in range(world)]      # generate_ours.py:176-177
    assert sum(e - s for s, e in spans) == n
    from oracle import solver_oracle as so
    assert all(launch.shard_bounds(n, world, r) == so.shard_bounds(n, world, r) for r in range(world))
    assert all(launch.shard_bounds_ceil(n, world, r) == so.shard_bounds_ceil(n, world, r) for r in range(world))


def test_batch_plan_seeds_and_names():
    plans = [list(launch.batch_plan(100, 8, r, 5, seed=43)) for r in range(8)]
    seen = []
    for r, plan in enumerate(plans):
        for b, idx, seed, stems in plan:
            assert seed == 43 + b                                   # same seed on every rank (gen_ppo.py:258-260)
            assert stems[0] == f"{r}_{b * 5:08d}"                     # gen_ppo.py:319-330
            seen += idx
    assert sorted(seen) == list(range(100))
    assert len(plans[7]) == 4 and len(plans[0]) == 3                 # 12 prompts per rank, last rank 16


WORKER = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from consolver_amd import launch
rank, world, local, dist = launch.init_distributed("gloo")
assert dist is not None and world == 2
lo, hi = launch.shard_bounds(37, world, rank)
dist.barrier()
t = launch.reduce_max_seconds(dist, 1.0 + rank)
rep = launch.gather_report(dist, hi - lo, float(sum(range(lo, hi))))
assert t == 2.0, t
g = launch.average_gradients(dist, torch.arange(6, dtype=torch.float32) * (rank + 1))      # DDP-style mean of the policy gradients
assert torch.equal(g, torch.arange(6, dtype=torch.float32) * 1.5), g
assert sum(c for c, _ in rep) == 37 and sum(s for _, s in rep) == sum(range(37)), rep
dist.barrier()
dist.destroy_process_group()
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"rank{rank}.ok"), "w").write("ok")   # a file per rank: stdout of two processes interleaves
'''


def test_two_process_gloo_launch(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").read_text() == "ok" and (tmp_path / "rank1.ok").read_text() == "ok"
