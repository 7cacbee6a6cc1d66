"""Dev probe: host wall time of every C-ABI call of one frame (single stream), run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, __graft_entry__ as g
mvo = g.load_package(); ctx = mvo.Context(0, max_keypoints=2000)
seq = mvo.synth.Sequence(640, 480, 8, seed=1234, tex_size=1024)
frames = [torch.from_numpy(seq.frame(i)).cuda() for i in range(8)]
pb = mvo.synth.ba_problem(5, 2000, 7)
h = ctx.ba_prepare(pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"], fix_points=False)
T = {}
def tick(name, t0):
    T.setdefault(name, []).append(time.perf_counter() - t0)
prev = None
for it in range(60):
    t = frames[it % 8]
    t0 = time.perf_counter(); k = ctx.calc_keypoints_dev(t.data_ptr(), 640, 480, 1920, 3, cap=2016); tick("calc_keypoints_dev", t0)
    t0 = time.perf_counter(); k, _, dptr = ctx.calc_descriptors_dev(k, want_host=False); tick("calc_descriptors_dev", t0)
    if prev:
        t0 = time.perf_counter(); m = ctx.match_features_dev(prev[0], prev[1], dptr, len(k), 2, 2.0, 0.8); tick("match_features_dev", t0)
    prev = (dptr, len(k))
    t0 = time.perf_counter(); ctx.ba_solve_resident(h); tick("ba_solve_resident(async)", t0)
    t0 = time.perf_counter(); ctx.ba_fetch(h, want_points=False); tick("ba_fetch", t0)
for k_, v in T.items():
    v = np.array(v[10:]) * 1e3
    print("%-28s median %.3f ms  p90 %.3f" % (k_, np.median(v), np.percentile(v, 90)))
