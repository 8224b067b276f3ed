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


def test_bench_launcher_branch_dry_run():
    """`python bench.py --gpus 2 --dry-run`: bench.py's own child-launch branch (torch.distributed.run, one process per rank, 127.0.0.1
    rendezvous) + sharding + start barrier + max-over-ranks reduction + the single JSON line from rank 0, with gloo in place of RCCL and no
    CUDA call (the scaling curve itself has never been measured: the driver owns the 8-GPU node)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # exactly ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec["dry_run"] and rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["shards"] == [[0, 16], [16, 32]] and rec["images"] == 16 * 2 * 3
    assert rec["max_elapsed_s"] == 0.002                  # MAX over ranks (rank 1 reported the larger time)
    assert rec["cuda_initialised"] is False
    # the self-proving N > 1 record: one row per rank, gathered by the collective (launch.gather_rank_records), distinct ranks / devices / shards
    pr = rec["per_rank"]
    assert len(pr) == 2 and [p["rank"] for p in pr] == [0, 1] and len({p["device"] for p in pr}) == 2
    assert [p["prompt_shard"] for p in pr] == [[0, 16], [16, 32]] and all(p["images"] == 16 * 3 for p in pr)
    assert max(p["elapsed_s"] for p in pr) == rec["max_elapsed_s"]
    assert rec["per_rank_distinct_devices"] == 2 and len({p["host_id"] for p in pr}) == 1      # one node: same host id, distinct (host, device) pairs
    # N > 1: nothing but the sampling loop runs around the barriers -- the CPU baseline, the sub-record extras and the vendor ceilings are N = 1 only
    sw = rec["side_work_after_timed_region"]
    assert sw["cpu_baseline"] is False and sw["extras"] is False and sw["ceilings"] is False
    # the per-rank shards tile [0, 16 N) exactly as gen_ppo.py:349-357 cuts the prompt list (contiguous, in rank order, last rank takes the remainder) -- at the
    # world sizes the driver's scaling run uses, through the same launch path
    for N in (4, 8):
        rN = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(N), "--steps", "1", "--warmup", "0", "--dry-run"],
                            capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert rN.returncode == 0, rN.stdout + rN.stderr
        recN = json.loads([l for l in rN.stdout.splitlines() if l.startswith("{")][0])
        shards = [p["prompt_shard"] for p in sorted(recN["per_rank"], key=lambda p: p["rank"])]
        per = (16 * N) // N
        assert shards == [[r * per, (r + 1) * per if r < N - 1 else 16 * N] for r in range(N)], shards
        assert shards[0][0] == 0 and shards[-1][1] == 16 * N and all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
        assert recN["n_gpus"] == N and recN["images"] == 16 * N and recN["per_rank_distinct_devices"] == N
    # N = 1 takes the in-process path (no launcher)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    rec1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert rec1["n_gpus"] == 1 and rec1["shards"] == [[0, 16]] and len(rec1["per_rank"]) == 1
    assert rec1["side_work_after_timed_region"]["cpu_baseline"] is True and rec1["side_work_after_timed_region"]["extras"] is True


GEN_WORKER = r'''
import os, sys, types, torch
sys.path.insert(0, %r)
from consolver_amd import generate as gen, launch
rank, world, local, dist = launch.init_distributed("gloo")
out = sys.argv[1]
N, bs, seed = 11, 3, 43
prompts = [f"prompt {i}" for i in range(N)]
pe = torch.arange(N, dtype=torch.float32).view(N, 1, 1).repeat(1, 2, 4)          # prompt i is recognisable from its embedding
ne = torch.zeros_like(pe)

class Engine:                                   # stands in for SDSamplingEngine: image = f(prompt embedding, noise), no GPU
    unet = types.SimpleNamespace(config=dict(in_channels=4, sample_size=8))
    def generate(self, pe, ne, latents=None, num_inference_steps=8, use_graph=False, output_type="pt"):
        B = pe.shape[0]
        img = torch.zeros(B, 3, 16, 16)
        img[:, 0] = pe[:, 0, 0].view(B, 1, 1) / 16.0                                # red channel = global prompt index / 16
        img[:, 1] = torch.sigmoid(latents.float().mean(dim=(1, 2, 3))).view(B, 1, 1)  # green = a function of the batch's noise
        return img

dist.barrier()
n = gen.generate_imgs(out, prompts, pe, ne, Engine(), 8, rank, world, seed, batch_size=bs, device=torch.device("cpu"))
rep = launch.gather_report(dist, n, 0.0)
assert [c for c, _ in rep] == [5, 6], rep
dist.barrier()
dist.destroy_process_group()
'''


def test_two_process_generate_writes_rank_indexed_files(tmp_path):
    """consolver_amd.generate's rank logic (gen_ppo.py:333-379) in two real processes (gloo): shard rule, ragged last batch, per-batch seed
    `seed + batch_idx` identical on both ranks, `{rank}_{idx:08d}.png/.txt` naming, and that each file holds ITS prompt's image."""
    import numpy as np
    import torch
    from PIL import Image
    from consolver_amd import generate as gen
    script = tmp_path / "gen_worker.py"
    script.write_text(GEN_WORKER % ROOT)
    out = tmp_path / "generation"
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(script), str(out)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    names = sorted(f for f in os.listdir(out) if f.endswith(".png"))
    assert names == [f"0_{i:08d}.png" for i in range(5)] + [f"1_{i:08d}.png" for i in range(6)]
    for rank, lo, cnt in ((0, 0, 5), (1, 5, 6)):
        for i in range(cnt):
            assert open(out / f"{rank}_{i:08d}.txt").read() == f"prompt {lo + i}"
            px = np.asarray(Image.open(out / f"{rank}_{i:08d}.png").convert("RGB"))
            assert px[0, 0, 0] == round((lo + i) / 16.0 * 255)                 # the image generated FOR that prompt
            # green encodes the batch noise: seed + batch_idx with batch_idx LOCAL to the rank, so both ranks share seeds 43, 44
            noise = gen.prepare_latents(min(3, cnt - (i // 3) * 3), (4, 8, 8), 43 + i // 3, torch.device("cpu"))
            want = torch.sigmoid(noise.float().mean(dim=(1, 2, 3)))[i % 3]
            assert abs(int(px[0, 0, 1]) - round(float(want) * 255)) <= 1
