"""The N>1 path of bench.py on CPU: world_size 2, gloo.  The hot path shards by sequence with no data-path
collective; what crosses ranks is the max-over-ranks clock and ONE all_gather of the trajectories."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import bench
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    streams, steps = 3, 5
    ids = bench.shard_ids(rank, streams)
    assert ids == [rank * streams + s for s in range(streams)]
    all_ids = [None] * world
    dist.all_gather_object(all_ids, ids)
    flat = sum(all_ids, [])
    assert sorted(flat) == list(range(world * streams)), flat          # disjoint cover
    traj = np.zeros((streams, steps, 12))
    for s, sid in enumerate(ids):
        traj[s] = sid * 100 + np.arange(steps)[:, None] + np.arange(12)[None] / 100.0
    g = bench.gather_trajectories(dist, traj, "cpu")
    assert g.shape == (world, streams, steps, 12)
    for r in range(world):
        for s in range(streams):
            assert np.array_equal(g[r, s], (r * streams + s) * 100 + np.arange(steps)[:, None] + np.arange(12)[None] / 100.0)
    t = bench.max_over_ranks(dist, 1.0 + rank, "cpu")
    assert t == float(world)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                       capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2
