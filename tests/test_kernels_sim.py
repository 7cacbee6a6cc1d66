"""The extraction, matching, tracking and key-frame kernel SOURCES (csrc/orb_kernels.hip, match_kernels.hip, track_kernels.hip) with
their host side (csrc/mvo_api.cpp, orb_host.cpp, track_host.cpp), compiled for x86 against tests/sim/hip_emu and executed thread for
thread on the CPU -- every GPU thread a fiber, wave operations (ballots, shuffles, the i8 MFMA of the matcher) rendez-vous points of
the 64 fibers of a wave, arrival counters and write-through partials real shared memory.  The tests below are the MI355X tests of
tests/test_gpu_*.py themselves, run through that build: the same calls into the same C-ABI, the same comparison with the oracle,
available where no GPU exists (pytest -m "not gpu").  The emulation is a test aid: nothing of it is linked into libmvo_hip.so
(test_abi.py); the product has no CPU path."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
import test_gpu_keyframe as T_kf
import test_gpu_match as T_match
import test_gpu_orb as T_orb
import test_gpu_track as T_track

SIM_DIR = os.path.join(ROOT, "tests", "sim")
SIM_LIB = os.path.join(SIM_DIR, "_build", "libmvo_sim.so")


@pytest.fixture(scope="module")
def simlib():
    subprocess.check_call(["make", "-C", SIM_DIR, "-s", "-j8", "_build/libmvo_sim.so"])
    lib = C.CDLL(SIM_LIB)
    lib.mvo_last_error.restype = C.c_char_p
    lib.mvo_destroy.restype = None
    return lib


@pytest.fixture()
def simmvo(mvo, simlib, monkeypatch):
    """The product's Python mirror of the C-ABI with its library handle pointing at the emulated build."""
    real_init = mvo.Context.__init__

    class SimContext(mvo.Context):
        def __init__(self, device=0, **orb_params):
            real_init(self, device, **orb_params)

    class HostTensor:  # "device memory" of the emulated runtime is host memory: stands in for torch's .cuda() tensors
        def __init__(self, a):
            self.a = np.ascontiguousarray(a)

        def data_ptr(self):
            return self.a.ctypes.data

    monkeypatch.setattr(mvo, "load_library", lambda: simlib)
    monkeypatch.setattr(mvo, "Context", SimContext)
    monkeypatch.setattr(T_track, "_to_device", HostTensor)
    return mvo


@pytest.fixture()
def simctx(simmvo):
    c = simmvo.Context(0)
    yield c
    c.close()


# ------------------------------------------------------------------------------------------------ extraction
@pytest.mark.parametrize("w,h,ch", [(160, 120, 3), (333, 251, 1), (640, 480, 3)])
def test_pyramid_blur_candidates(simmvo, O, simctx, w, h, ch):
    T_orb.test_pyramid_blur_candidates_bit_exact(simmvo, O, simctx, w, h, ch)


@pytest.mark.parametrize("nlevels,sf", [(4, 1.2), (8, 1.2), (3, 2.0)])
def test_both_pyramid_kernels(simmvo, O, simctx, nlevels, sf):
    T_orb.test_both_pyramid_kernels_bit_exact(simmvo, O, simctx, nlevels, sf)


@pytest.mark.parametrize("w,h,ch", [(160, 120, 3), (333, 251, 1), (640, 480, 3)])
def test_keypoints_and_descriptors(simmvo, O, simctx, w, h, ch):
    T_orb.test_keypoints_and_descriptors_bit_exact(simmvo, O, simctx, w, h, ch)


def test_selection_shapes_and_degenerate_images(simmvo, O, simctx):
    T_orb.test_legacy_inter_linear_pyramid_bit_exact(simmvo, O, simctx)
    T_orb.test_quota_limited_selection_and_small_thresholds(simmvo, O, simctx)
    T_orb.test_other_pyramid_shapes(simmvo, O, simctx)
    T_orb.test_grid_latching_and_degenerate_images(simmvo, O, simctx)
    T_orb.test_error_paths(simmvo, simctx)


@pytest.mark.parametrize("w,h", [(333, 251), (640, 480)])
def test_both_candidate_orderings_and_both_descriptor_forms(simmvo, O, w, h):
    T_orb.test_both_candidate_orderings_bit_exact(simmvo, O, w, h)


# ------------------------------------------------------------------------------------------------ matching
@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("kind,nq,nt", [("uniform", 2000, 2000), ("ties", 1, 1), ("perturbed", 63, 65), ("ties", 129, 15),
                                        ("perturbed", 1000, 2500), ("uniform", 17, 255)])
def test_knn2(simmvo, O, simctx, kind, nq, nt, mfma):
    T_match.test_knn2_bit_exact(simmvo, O, simctx, kind, nq, nt, mfma)


def test_matcher_structured_empty_radius(simmvo, O, simctx):
    T_match.test_knn2_mfma_on_structured_descriptors(simmvo, O, simctx)
    T_match.test_knn2_empty_sets(simmvo, O, simctx)
    for nq, nt, r in [(500, 700, 50.0), (64, 64, 0.0), (100, 3, 1e6)]:
        T_match.test_radius_l1_bit_exact(simmvo, O, simctx, nq, nt, r)


@pytest.mark.parametrize("method", [1, 2, 3])
def test_match_features(simmvo, O, simctx, method):
    T_match.test_match_features_bit_exact(simmvo, O, simctx, method, "perturbed")
    T_match.test_match_features_bit_exact(simmvo, O, simctx, method, "ties")


# ------------------------------------------------------------------------------------------------ tracking rows
@pytest.mark.parametrize("seed,n_map", [(3, 4000), (4, 1), (6, 1025)])
def test_map_points_in_view(simmvo, O, simctx, seed, n_map):
    T_track.test_map_points_in_view_bit_exact(simmvo, O, simctx, seed, n_map)


def test_solve_pnp_ransac(simmvo, O, simctx):
    T_track.test_solve_pnp_ransac_matches_the_oracle(simmvo, O, simctx, 11, {})
    T_track.test_solve_pnp_ransac_matches_the_oracle(simmvo, O, simctx, 12, dict(outlier_frac=0.5))
    T_track.test_all_hypotheses_bit_exact(simmvo, O, simctx)
    T_track.test_solve_pnp_ransac_edge_cases(simmvo, O, simctx)
    T_track.test_host_corrects_a_wrong_device_choice(simmvo, O, simctx)
    # contexts that share the GPU: the first 32 hypotheses, the rest only when the sequential loop goes on (one chunk / two chunks)
    T_track.test_solve_pnp_ransac_in_chunks_matches_the_oracle(simmvo, O, 11, {})
    T_track.test_solve_pnp_ransac_in_chunks_matches_the_oracle(simmvo, O, 19, dict(outlier_frac=0.8))


def test_tracking_step(simmvo, O, simctx):
    T_track.test_tracking_step_end_to_end(simmvo, O, simctx)


# ------------------------------------------------------------------------------------------------ key-frame row
def test_triangulation_and_essential_matrix(simmvo, O, simctx):
    T_kf.test_triangulate_points_bit_exact(simmvo, O, simctx, 600, 21, {})
    T_kf.test_triangulate_points_bit_exact(simmvo, O, simctx, 257, 23, dict(outlier_frac=0.5))
    T_kf.test_triangulate_edge_cases(simmvo, O, simctx)
    T_kf.test_retain_good_triangulation_matches_oracle(simmvo, O, simctx)
    T_kf.test_find_essential_inliers_matches_the_oracle(simmvo, O, simctx, 500, 8, {})
    T_kf.test_find_essential_inliers_matches_the_oracle(simmvo, O, simctx, 500, 9, dict(outlier_frac=0.5))
    T_kf.test_find_essential_inliers_edge_cases(simmvo, O, simctx)


def test_keyframe_insertion(simmvo, O, simctx):
    T_kf.test_keyframe_insertion_end_to_end(simmvo, O, simctx)


# ------------------------------------------------------------------------------------------------ the whole hot path, several callers
def test_concurrent_contexts_and_siblings(simmvo):
    """Six host threads, a ctx each, extraction + matching + BA at the same time through one emulated device; a sibling ctx on
    its parent's stream: every caller gets what a lone ctx gets (tests/test_gpu_concurrency.py)."""
    import test_gpu_concurrency as T_conc
    T_conc.test_concurrent_contexts_reproduce_the_serial_results(simmvo)
    T_conc.test_sibling_context_shares_the_stream_and_nothing_else(simmvo)


# ------------------------------------------------------------------------------------------------ the C++ side of the boundary
@pytest.fixture()
def sim_as_the_library(simlib, tmp_path, monkeypatch):
    """The C++ test programs of host/tests and the headless run_vo link libmvo_hip.so by name (DT_RUNPATH): a directory in front of the
    search path that holds the emulated build under that name makes the SAME binaries run on the CPU."""
    d = tmp_path / "simlib"
    d.mkdir()
    os.symlink(SIM_LIB, d / "libmvo_hip.so")
    monkeypatch.setenv("LD_LIBRARY_PATH", str(d) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    return d


def test_cpp_dropin_translation_units(mvo, O, tmp_path, sim_as_the_library):
    """The two drop-in translation units (feature_match_mvo.cpp, g2o_ba_mvo.cpp) behind the reference-shaped headers, driven like the
    reference's call sites (tests/test_gpu_host_adapter.py): extraction, the three matchers, both bundle adjustments, every function
    of the replaced headers."""
    import test_gpu_host_adapter as T_host
    T_host.test_cpp_dropin_matches_oracle(mvo, O, tmp_path)
    T_host.test_every_function_of_the_replaced_headers_runs_like_the_reference(mvo, O, tmp_path)


def test_cpp_tracking_and_keyframe_mirrors(mvo, O, tmp_path, sim_as_the_library):
    import test_gpu_host_adapter as T_host
    T_host.test_cpp_tracking_mirror_matches_oracle(mvo, O, tmp_path)
    T_host.test_cpp_keyframe_mirror_matches_oracle(mvo, O, tmp_path)


def test_tracking_loop_and_headless_run_vo(mvo, O, tmp_path, sim_as_the_library):
    """The frame loop of the C++ mirror on a synthetic sequence and the headless run_vo on PNG frames, stage by stage against the
    oracle chain (tests/test_gpu_host_adapter.py, tests/test_gpu_run_vo.py) -- extraction to bundle adjustment in one process."""
    import test_gpu_host_adapter as T_host
    import test_gpu_run_vo as T_vo
    T_host.test_cpp_tracking_loop_follows_the_ground_truth(mvo, tmp_path)
    T_vo.test_run_vo_equals_the_oracle_chain(mvo, O, tmp_path)


@pytest.mark.parametrize("order", ["reverse", "shuffle"])
def test_results_do_not_depend_on_the_thread_order(simmvo, O, simctx, order, monkeypatch):
    """A missing barrier or wave hand-off shows as a result that depends on the order in which the emulated threads of a workgroup
    run between two rendez-vous points: extraction (both descriptor forms), both matchers, the PnP hypotheses."""
    monkeypatch.setenv("EMU_ORDER", order)
    T_orb.test_keypoints_and_descriptors_bit_exact(simmvo, O, simctx, 333, 251, 1)
    T_orb.test_both_candidate_orderings_bit_exact(simmvo, O, 333, 251)
    for mfma in (1, 0):
        T_match.test_knn2_bit_exact(simmvo, O, simctx, "ties", 300, 700, mfma)
    T_track.test_map_points_in_view_bit_exact(simmvo, O, simctx, 6, 1025)
    T_track.test_all_hypotheses_bit_exact(simmvo, O, simctx)
    T_kf.test_find_essential_inliers_matches_the_oracle(simmvo, O, simctx, 500, 8, {})


BENCH_PARITY_SCRIPT = r"""
# bench.py's own parity check (the `parity` object of its JSON line) executed on the CPU: two sequence shards of the native frame
# loop run on the emulated build (loaded under the library's name, so that libmvo_frame_loop.so resolves to it as well), THROUGHPUT
# mode with the resident solver grid forced, then bench.parity_check holds what the loops produced to the oracle.
import ctypes as C, os, sys, types
import numpy as np
root, simdir = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
import bench
import __graft_entry__ as graft
mvo = graft.load_package()
sim = C.CDLL(os.path.join(simdir, "libmvo_hip.so"))
sim.mvo_last_error.restype = C.c_char_p
sim.mvo_destroy.restype = None
mvo.load_library = lambda: sim
class HostTensor:
    def __init__(self, a): self.a = np.ascontiguousarray(a)
    def data_ptr(self): return self.a.ctypes.data
# (the benchmark's own window size: smaller windows keep their rows in LDS and never go to the resident grid)
args = bench.parse(["--streams", "1", "--frames", "3", "--windows", "2", "--max-kp", "500", "--width", "320", "--height", "240", "--ba-cut", "throughput"])
shards = []
for sid in (0,):
    seq = mvo.synth.Sequence(args.width, args.height, args.frames, seed=1234 + sid, tex_size=512)
    host = [seq.frame(i) for i in range(args.frames)]
    shards.append(bench.Shard(mvo, types.SimpleNamespace(), 0, sid, args, "rebuild", True, frames=(host, [HostTensor(f) for f in host])))
ok = False
try:
    bench.run_steps(shards, 2)
    par = bench.parity_check(args, shards, 2)
    print(par, flush=True)
    assert par["ba"] == "bit-exact" and par["ba_windows_checked"] == 1 and par["orb"] == "bit-exact" and par["match"] == "bit-exact", par
    assert par["ba_workgroups_per_window"] == [14], par            # the throughput cut ...
    assert shards[0].ctx.ba_launch_stats()["resident_windows"] >= 4, "... through the resident grid"
    # a corrupted trajectory row must be reported, not waved through
    shards[0].traj[-1] = np.asarray(shards[0].traj[-1]) + 1e-9
    bad = bench.parity_check(args, shards, 2)
    assert bad["ba"] == "MISMATCH" and bad["ba_mismatches"][0]["shard"] == 0, bad
    ok = True
    print("PARITY-OK", flush=True)
finally:
    sys.stdout.flush()
    sys.stderr.flush()
    if not ok:
        import traceback
        traceback.print_exc()
    os._exit(0 if ok else 1)      # (the emulated resident grid's threads must not keep a failed run alive)
"""


def test_bench_parity_check_on_the_emulated_build(tmp_path, sim_as_the_library):
    """The `parity` object of bench.py's JSON line (round-4 verdict: the bench checked its timed outputs for finiteness only): the
    native frame loops, the window export (restore -> marshal -> flatten), the capture of the loop's last extraction and the
    comparison with the oracle, executed without a GPU."""
    script = tmp_path / "bench_parity.py"
    script.write_text(BENCH_PARITY_SCRIPT)
    env = dict(os.environ, MVO_BA_SERVICE="2")
    r = subprocess.run([sys.executable, str(script), ROOT, str(sim_as_the_library)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "PARITY-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
