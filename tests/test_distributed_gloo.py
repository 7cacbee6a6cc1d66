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

    # ---- the whole control flow of bench.run_benchmark (warm-up, barriers, timed region, MAX over ranks, the one
    # collective with REAL per-frame rows) with stand-in shards: what a rank does at N > 1, executed, not just its helpers
    import time, types
    class StubCtx:
        def ba_launch_stats(self, reset=False): return dict(launches=1, windows=1, ms=1.0)
        def ba_service_times(self): return {}
        def synchronize(self): pass
    class StubShard:
        def __init__(self, sid):
            self.id, self.traj, self.ctx, self.frame = sid, [], StubCtx(), 0
        def state(self):
            return types.SimpleNamespace(ba_trials=10 * self.frame, ba_solves=self.frame, ba_edges=100 * self.frame, frame_no=self.frame)
        def run(self, n):
            for _ in range(n):
                self.traj.append(np.full(12, 1000.0 * self.id + self.frame))
                self.frame += 1
            time.sleep(0.01 * (1 + rank))             # the slower rank defines the clock
    class StubEnv:
        device = "cpu"
        def init_process_group(self, d):
            # (ONE process group per process: re-initialising after a destroy re-uses the launcher's store and can hang)
            if not d.is_initialized(): d.init_process_group("gloo")
        def sync(self): pass
        def make_shard(self, sid, args, ba_mode, pipeline, **kw): return StubShard(sid)
    # a step = --frames-per-step consecutive frames of every shard: 4 steps x 3 frames timed after 2 x 3 warm-up frames
    args = bench.parse(["--gpus", str(world), "--steps", "4", "--warmup", "2", "--streams", "3", "--frames-per-step", "3"])
    R = bench.run_benchmark(args, StubEnv())
    assert R["world"] == world and R["nframes"] == 12 and R["traj_all"].shape == (world, 3, 12, 12)
    for r in range(world):
        for s_ in range(3):
            sid = r * 3 + s_
            assert np.array_equal(R["traj_all"][r, s_, :, 0], 1000.0 * sid + np.arange(6, 18)), R["traj_all"][r, s_, :, 0]
    assert R["elapsed"] >= 0.01 * world                                    # MAX over ranks
    assert abs(R["value"] - world * 3 * 12 / R["elapsed"]) < 1e-9          # whole-job aggregate
    import torch.distributed as dist2
    dist2.barrier()
    dist2.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)  # (a cold page cache makes the first torch import take minutes)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2
