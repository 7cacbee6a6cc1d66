"""Dev probe: N host threads, one ctx each, bundle_adjustment in throughput mode (resident solver service), optionally with an
extraction + matching loop on a second ctx per thread."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as g
mvo = g.load_package()
N = int(sys.argv[1]); reps = int(sys.argv[2]); with_extract = int(sys.argv[3]) if len(sys.argv) > 3 else 0
svc = int(sys.argv[4]) if len(sys.argv) > 4 else 2   # mvo_debug_set("ba_service"): 2 = always the resident grid, 0 = launch path
pbs = [mvo.synth.ba_problem(5, 2000, 7 + k) for k in range(4)]
img = mvo.synth.small_test_image(1, 640, 480)
errs = []
def work(k):
    try:
        c = mvo.Context(0); c.ba_set_mode("throughput"); mvo.debug_set("ba_service", svc)
        ce = mvo.Context(0, max_keypoints=2000) if with_extract else None
        for r in range(reps):
            pb = pbs[(k + r) % len(pbs)]
            if ce is not None:
                kp = ce.calc_keypoints(img); ce.calc_descriptors(img, kp, reuse_pyramid=True)
            c.bundle_adjustment(pb["poses0"], pb["points0"], pb["edge_pose"], pb["edge_point"], pb["edge_uv"], pb["focal"], pb["cx"], pb["cy"], fix_points=False)
        c.close()
    except Exception as e:
        errs.append((k, repr(e)))
th = [threading.Thread(target=work, args=(k,)) for k in range(N)]
t0 = time.time()
for t in th: t.start()
for t in th: t.join()
dt = time.time() - t0
c = mvo.Context(0)
st = c.ba_launch_stats()
print("N %d reps %d extract %d: %.2fs -> %.0f solves/s errors %s | per window: %.3f ms, %.0f kcycles | %s" % (N, reps, with_extract, dt, N * reps / dt, errs[:2], st["ms"] / max(st["windows"], 1), st.get("resident_cycles", 0) / max(st["resident_windows"], 1) / 1e3, st))
