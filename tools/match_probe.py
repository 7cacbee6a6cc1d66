"""Dev probe: k_knn2 / k_knn2_mfma kernel time (HIP events) with and without the in-kernel delivery into pinned host memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as g
mvo = g.load_package(); ctx = mvo.Context(0)
for nq, nt in ((2000, 2000), (4000, 4000), (1000, 2000)):
    q, t = mvo.synth.match_inputs("perturbed", nq, nt)
    for mfma in (1, 0):
        for host in (1, 0):
            mvo.debug_set("match_mfma", mfma); mvo.debug_set("match_host_out", host)
            for _ in range(5): ctx.match_knn2(q, t)
            ctx.profile_enable(True); ctx.profile_reset()
            for _ in range(50): ctx.match_knn2(q, t)
            pr = ctx.profile_get(); ctx.profile_enable(False)
            n, ms = pr.get("k_knn2_mfma", pr.get("k_knn2"))
            print("%dx%d mfma=%d host_out=%d: %.2f us per launch" % (nq, nt, mfma, host, ms / n * 1e3))
mvo.debug_set("match_mfma", 1); mvo.debug_set("match_host_out", 1)
