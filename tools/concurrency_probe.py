"""Dev probe: do BA solves / extractions from several ctx (streams, host threads) overlap on the GPU?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, __graft_entry__ as g
mvo = g.load_package()
seq = mvo.synth.Sequence(640, 480, 4, seed=1234, tex_size=1024)
frames = [torch.from_numpy(seq.frame(i)).cuda() for i in range(4)]
pb = mvo.synth.ba_problem(5, 2000, 7)
args = (pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"])
NMAX = 16
ctxs = [mvo.Context(0, max_keypoints=2000) for _ in range(NMAX)]
hs = [c.ba_prepare(*args, fix_points=False) for c in ctxs]
for c in ctxs:
    c.calc_keypoints_dev(frames[0].data_ptr(), 640, 480, 1920, 3, cap=2016)

def run(kind, n, iters):
    def work(i):
        c = ctxs[i]
        for it in range(iters):
            if kind == "ba":
                c.ba_solve_resident(hs[i]); c.ba_fetch(hs[i], want_points=False)
            else:
                k = c.calc_keypoints_dev(frames[it % 4].data_ptr(), 640, 480, 1920, 3, cap=2016)
                c.calc_descriptors_dev(k, want_host=False)
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    return n * iters / dt

for kind, iters in (("ba", 30), ("extract", 200)):
    for n in (1, 2, 4, 8, 16):
        run(kind, n, 3)
        print(kind, "threads", n, "-> %.0f /s" % run(kind, n, iters))
