"""The bundle-adjustment kernel SOURCE (csrc/ba_kernels.hip) and its host side (csrc/ba_host.cpp, csrc/mvo_api_ba.cpp),
compiled for x86 against tests/sim/hip_emu and executed thread for thread on the CPU: every GPU thread of k_ba_lm is a
fiber, every workgroup an OS thread, wave operations (shuffles, v_readlane, the f64 MFMA) are rendez-vous points, the
cross-workgroup hand-offs go through real shared memory.  The result must equal the oracle's blocked restatement BIT FOR
BIT -- the same check tests/test_gpu_ba.py::test_bitwise_* makes on the MI355X, available where no GPU exists
(pytest -m "not gpu").  The emulation is a test aid: nothing of it is linked into libmvo_hip.so (test_abi.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
import test_gpu_ba as gpu_ba_tests
from test_gpu_ba import _args, _bitwise, _fix

SIM_DIR = os.path.join(ROOT, "tests", "sim")
SIM_LIB = os.path.join(SIM_DIR, "_build", "libmvo_ba_sim.so")


@pytest.fixture(scope="module")
def simlib():
    subprocess.check_call(["make", "-C", SIM_DIR, "-s", "-j4"])
    lib = C.CDLL(SIM_LIB)
    lib.mvo_last_error.restype = C.c_char_p
    lib.mvo_destroy.restype = None
    return lib


@pytest.fixture()
def simctx(mvo, simlib):
    class SimContext(mvo.Context):  # the ctypes mirror of the C-ABI, bound to the emulated build of the BA sources
        def __init__(self):
            self.lib = simlib
            h = C.c_void_p()
            assert simlib.mvo_create(C.byref(h), 0) == 0
            self.h, self.device, self.params = h, 0, {}

    c = SimContext()
    yield c
    simlib.mvo_debug_set(b"ba_wgs", 0)
    simlib.mvo_debug_set(b"ba_mfma", 1)
    c.close()


def test_kernel_source_reproduces_the_blocked_oracle_on_the_benchmarked_window(mvo, O, simctx):
    """BA5 as bench.py solves it: 5 poses / 2000 landmarks / ~9.4k edges, no fixed vertex, 32 workgroups x 512 threads,
    all 50 iterations, every trial's lambda / chi2 / rho / decision, final poses and landmarks: zero difference."""
    st, plan = _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=False)
    assert st["iterations"] == 50 and st["trials"] > 60 and plan["wgs"] == 28  # (one XCD minus the 4 CUs left to other kernels)


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("case", ["pose_only", "anchored", "fixed0_info", "tiny_one_range", "ragged_chain_tails", "dups"])
def test_kernel_source_variants(mvo, O, simctx, simlib, case, mfma):
    simlib.mvo_debug_set(b"ba_mfma", mfma)
    if case == "pose_only":
        _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=True)
    elif case == "anchored":
        pb = mvo.synth.ba_problem(5, 1000, 7)
        pb["poses0"][:2] = pb["poses_gt"][:2]
        _bitwise(mvo, O, simctx, pb, fix_points=False, pose_fixed=_fix(5, 2))
    elif case == "fixed0_info":
        _bitwise(mvo, O, simctx, mvo.synth.ba_problem(4, 700, 21), fix_points=False, pose_fixed=_fix(4, 1),
                 info=(2.0, 0.3, 0.3, 1.5), huber_delta=1.5)
    elif case == "tiny_one_range":
        st, plan = _bitwise(mvo, O, simctx, mvo.synth.ba_problem(3, 40, 5), fix_points=False)
        assert plan["wgs"] == 1
    elif case == "ragged_chain_tails":
        for L in (37, 41, 42, 43, 150):
            _bitwise(mvo, O, simctx, mvo.synth.ba_problem(3, L, 100 + L), fix_points=False, max_iterations=6)
            _bitwise(mvo, O, simctx, mvo.synth.ba_problem(3, L, 200 + L), fix_points=False, max_iterations=6, pose_fixed=_fix(3, 1))
    else:
        pb = mvo.synth.ba_problem(3, 300, 6)
        pb = dict(pb, edge_pose=np.concatenate([pb["edge_pose"], pb["edge_pose"][:90]]),
                  edge_point=np.concatenate([pb["edge_point"], pb["edge_point"][:90]]),
                  edge_uv=np.concatenate([pb["edge_uv"], pb["edge_uv"][:90] + 0.3]))
        _bitwise(mvo, O, simctx, pb, fix_points=False, pose_fixed=_fix(3, 1))


@pytest.mark.parametrize("order", ["reverse", "shuffle"])
def test_result_does_not_depend_on_the_thread_order(mvo, O, simctx, order, monkeypatch):
    """A missing barrier / hand-off shows as a result that depends on the order in which the emulated threads of a
    workgroup run between two rendez-vous points."""
    monkeypatch.setenv("EMU_ORDER", order)
    _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 600, 3), fix_points=False, max_iterations=10)
    _bitwise(mvo, O, simctx, mvo.synth.ba_problem(4, 300, 4), fix_points=True, max_iterations=10)


def test_batched_windows_equal_their_single_solves(mvo, simctx):
    pbs = [mvo.synth.ba_problem(5, 600, 7), mvo.synth.ba_problem(3, 300, 11), mvo.synth.ba_problem(4, 400, 21),
           mvo.synth.ba_problem(3, 40, 77)]
    singles = [simctx.bundle_adjustment(*_args(pb), fix_points=False, max_iterations=8) for pb in pbs]
    batch = simctx.ba_solve_batch([_args(pb) for pb in pbs], fix_points=False, max_iterations=8)
    for (P, X, st), (Pb, Xb, stb) in zip(singles, batch):
        assert np.array_equal(P, Pb) and np.array_equal(X, Xb) and st["trials"] == stb["trials"]


def test_throughput_mode_cuts_the_window_into_fewer_workgroups(mvo, O, simctx, simlib):
    """mvo_ba_set_mode(THROUGHPUT) under load (forced here: ba_service = 2): the benchmarked window on 14 workgroups (two
    windows per XCD, 4 CUs of it left to other kernels) -- more than 512 edges per range (the second edge of a thread keeps
    its rows in LDS), the Schur operands in two chunks -- still bit for bit the oracle.  Without load (the default policy,
    ba_service = 1, a lone caller) the same mode keeps the latency cut on the launch path."""
    simctx.ba_set_mode("throughput")
    simlib.mvo_debug_set(b"ba_service", 2)
    try:
        st, plan = _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=False)
        assert plan["wgs"] == 14 and plan["nsplit"] >= 2 and st["iterations"] == 50
        _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=True)
    finally:
        simlib.mvo_debug_set(b"ba_service", 1)
    st, plan = _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=False, max_iterations=3)
    assert plan["wgs"] == 28


@pytest.mark.parametrize("wgs,groups", [(40, 2), (72, 4)])
def test_windows_of_more_than_one_xcd_sum_their_schur_partials_per_group(mvo, O, simctx, simlib, wgs, groups, monkeypatch):
    """More than 32 workgroups (forced here on the benchmarked window): the partials of the workgroups g = k mod K are added
    first (the launch places a group on one XCD; the emulator's XCC id is blockIdx & 7 like the dispatcher's round-robin),
    then the K group sums -- the pose blocks that ride along from the second iteration on included; the trials that end at the
    failed factorisation (no all-to-all behind them) alternate between the two buffers of the group sums.  Still the oracle's
    bits, whatever order the emulated threads run in."""
    simlib.mvo_debug_set(b"ba_wgs", wgs)
    try:
        st, plan = _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=False, max_iterations=30)
        assert plan["wgs"] == wgs and plan["groups"] == groups and st["trials"] > st["iterations"]
        monkeypatch.setenv("EMU_ORDER", "shuffle")
        _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 9), fix_points=False, max_iterations=5, pose_fixed=_fix(5, 1))
        _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=True, max_iterations=4)
    finally:
        simlib.mvo_debug_set(b"ba_wgs", 0)


@pytest.mark.parametrize("block", [0, 1])
def test_six_to_ten_pose_windows_use_the_block_solver(mvo, O, simctx, simlib, block):
    """n + 1 in 33..64 (6..10 free poses): workgroup-wide block LDL^T (panel on one wave, MFMA tile updates on all); with
    the knob also for the 5-pose class.  BASELINE configs[3] (BA10: 10 poses / 4000 landmarks / ~36k edges) fits 64
    workgroups."""
    simlib.mvo_debug_set(b"ba_block_solver", block)
    try:
        _bitwise(mvo, O, simctx, mvo.synth.ba_problem(7, 900, 31), fix_points=False, max_iterations=8)
        _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 800, 5), fix_points=False, max_iterations=8)
        if block == 0:
            pb = mvo.synth.ba_problem(10, 4000, 13, width=1242, height=375, K=mvo.synth.KITTI_K)
            st, plan = _bitwise(mvo, O, simctx, pb, fix_points=False, max_iterations=3)
            assert plan["wgs"] <= 64 and plan["groups"] == 2  # (more than one XCD's worth of workgroups: grouped Schur exchange)
    finally:
        simlib.mvo_debug_set(b"ba_block_solver", 0)


def test_concurrent_clients_share_the_launch_service(mvo, simlib):
    """Several host threads with a ctx each submit windows at the same time: the launch thread packs them into grids within
    its CU share, queues the next grid behind the running one, the completion thread publishes -- every window must come
    out exactly as when it is solved alone."""
    import threading

    class Ctx(mvo.Context):
        def __init__(self):
            self.lib = simlib
            h = C.c_void_p()
            assert simlib.mvo_create(C.byref(h), 0) == 0
            self.h, self.device, self.params = h, 0, {}

    pbs = [mvo.synth.ba_problem(4, 300 + 40 * k, 50 + k) for k in range(6)]
    ref = Ctx()
    singles = [ref.bundle_adjustment(*_args(pb), fix_points=False, max_iterations=6) for pb in pbs]
    ref.close()
    simlib.mvo_debug_set(b"ba_cu_share", 16)  # (small share: forces several launches in flight)
    out = [None] * len(pbs)

    def work(k):
        c = Ctx()
        for _ in range(2):
            out[k] = c.bundle_adjustment(*_args(pbs[k]), fix_points=False, max_iterations=6)
        c.close()

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(pbs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    simlib.mvo_debug_set(b"ba_cu_share", 0)
    for (P, X, st), (Pb, Xb, stb) in zip(singles, out):
        assert np.array_equal(P, Pb) and np.array_equal(X, Xb) and st["trials"] == stb["trials"]


def test_register_resident_rows_flavour(mvo, O, simctx, simlib):
    """The planner keeps the Jacobian rows of a range in LDS when they fit (latency cut of the benchmarked window) and in the
    registers of the edge-owning threads otherwise; forced to registers the result must be the same bits."""
    simlib.mvo_debug_set(b"ba_edge_rows", 1)
    try:
        st, plan = _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=False, max_iterations=12)
        assert plan["wgs"] == 28
        _bitwise(mvo, O, simctx, mvo.synth.ba_problem(4, 700, 21), fix_points=False, pose_fixed=_fix(4, 1), max_iterations=12)
    finally:
        simlib.mvo_debug_set(b"ba_edge_rows", -1)


def test_resident_solver_service(mvo, O, simlib):
    """Throughput-mode windows go to the resident grid (k_ba_service): slots that stay on the device and pull windows from
    pinned mailboxes -- no launch per window, inputs read from the pinned image.  Several client threads, more jobs than
    slots, a latency-mode window in between (the resident grid leaves, a launch runs, the grid comes back): every result
    must equal the oracle's bits / the launch path's."""
    import threading

    class Ctx(mvo.Context):
        def __init__(self):
            self.lib = simlib
            h = C.c_void_p()
            assert simlib.mvo_create(C.byref(h), 0) == 0
            self.h, self.device, self.params = h, 0, {}

    a, b = C.c_longlong(), C.c_longlong()
    ms = C.c_double()
    simlib.mvo_ba_launch_stats(0, C.byref(a), C.byref(b), C.byref(ms), 1)
    simlib.mvo_debug_set(b"ba_service", 2)      # (the default policy only brings the grid up under load)
    pbs = [mvo.synth.ba_problem(4, 500 + 60 * k, 70 + k) for k in range(5)]
    c0 = Ctx()
    c0.ba_set_mode("throughput")
    st, plan = _bitwise(mvo, O, c0, pbs[0], fix_points=False, max_iterations=5)      # bit for bit through the service
    ref = [c0.bundle_adjustment(*_args(pb), fix_points=False, max_iterations=5) for pb in pbs]
    out = [[None, None] for _ in pbs]

    def work(k):
        c = Ctx()
        c.ba_set_mode("throughput")
        for rep in range(2):
            out[k][rep] = c.bundle_adjustment(*_args(pbs[k]), fix_points=False, max_iterations=5)
        c.close()

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(pbs))]
    for t in th:
        t.start()
    lat = Ctx()                                                                       # default mode: launch path
    Pl, Xl, stl = lat.bundle_adjustment(*_args(pbs[1]), fix_points=False, max_iterations=5)
    lat.close()
    for t in th:
        t.join()
    for k in range(len(pbs)):
        for rep in range(2):
            assert np.array_equal(ref[k][0], out[k][rep][0]) and np.array_equal(ref[k][1], out[k][rep][1])
    assert np.abs(Pl - ref[1][0]).max() < 1e-6                                        # (another cut: same optimum, other bits)
    simlib.mvo_ba_launch_stats(0, C.byref(a), C.byref(b), C.byref(ms), 0)
    assert b.value >= 1 + len(pbs) * 3 + 1 and ms.value > 0
    c0.close()
    simlib.mvo_debug_set(b"ba_service", 1)


def test_kernel_source_against_the_independent_sequential_oracle(mvo, O, simctx):
    """The two comparisons of tests/test_gpu_ba.py that do not go through the blocked twin, run on the emulated kernel: the
    bench window up to gauge (every landmark with >= 2 views within 1e-4) and the converged BA10-shaped window."""
    gpu_ba_tests.test_bench_window_matches_the_sequential_oracle_up_to_gauge(mvo, O, simctx)
    gpu_ba_tests.test_ba10_converged_against_the_sequential_oracle(mvo, O, simctx)


def test_windows_of_changing_shape_on_one_context(mvo, O, simctx):
    """One ctx, windows whose workgroup count / pose count change from call to call: the exchange areas sit at the head of the
    pooled device block and are cleared whenever their layout changes (a granule is matched by its tag alone -- bytes that
    held another window's data must never pass for one).  Every solve still equals the oracle bit for bit."""
    shapes = [(5, 2000, 7), (4, 500, 70), (5, 2000, 8), (3, 40, 5), (7, 900, 31), (4, 500, 71), (5, 2000, 7)]
    for F, L, seed in shapes:
        _bitwise(mvo, O, simctx, mvo.synth.ba_problem(F, L, seed), fix_points=False, max_iterations=3)
    _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=True, max_iterations=3)



def test_windows_of_more_than_ten_poses(mvo, O, simctx):
    gpu_ba_tests._many_pose_windows(mvo, O, simctx)


def test_stale_step_after_a_failed_solve(mvo, O, simctx, simlib):
    """g2o's behaviour after a failed linear solve (the solver's x stays, is applied and scored; a negative predicted decrease
    accepts it), on a benchmark window that meets both outcomes: the kernel source equals the blocked oracle trial by trial."""
    simctx.ba_set_mode("throughput")
    simlib.mvo_debug_set(b"ba_service", 2)
    try:
        gpu_ba_tests._stale_step_window(mvo, O, simctx)
    finally:
        simlib.mvo_debug_set(b"ba_service", 1)


def test_measurements_in_lds_or_device_memory(mvo, O, simctx, simlib):
    """A > 512-edge range keeps its measurements in LDS when that costs no extra chunk of U (BA5 on 14 workgroups), in device
    memory otherwise (forced here with the A/B knob on 13 workgroups, where the LDS form needs three chunks): the oracle's bits
    either way."""
    simctx.ba_set_mode("throughput")
    simlib.mvo_debug_set(b"ba_service", 2)
    try:
        st, plan = _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=False, max_iterations=12)
        assert plan["wgs"] == 14 and plan["nsplit"] == 4
        simlib.mvo_debug_set(b"ba_wgs", 13)
        st, plan = _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=False, max_iterations=12)
        assert plan["wgs"] == 13 and plan["nsplit"] == 4
        simlib.mvo_debug_set(b"ba_uv_global", 0)
        st, plan = _bitwise(mvo, O, simctx, mvo.synth.ba_problem(5, 2000, 7), fix_points=False, max_iterations=12)
        assert plan["wgs"] == 13 and plan["nsplit"] == 6      # (three chunks of two pieces)
    finally:
        simlib.mvo_debug_set(b"ba_wgs", 0)
        simlib.mvo_debug_set(b"ba_uv_global", 1)
        simlib.mvo_debug_set(b"ba_service", 1)
