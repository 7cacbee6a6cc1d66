"""CPU checks of the measurement plumbing: the PMC summary tool and bench.py's reader of its output, the algorithmic
work table, the JSON contract fields that do not need a GPU."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _counter_csv(path, counter, rows):
    with open(path, "w") as f:
        f.write('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id",'
                '"Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count",'
                '"Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n')
        for i, (name, v) in enumerate(rows):
            f.write('%d,%d,1,1,1,1,256,1,"%s",256,0,0,64,0,32,"%s",%s,0,1\n' % (i, i, name, counter, v))


def test_pmc_summary_and_reader(tmp_path, monkeypatch):
    f, w, out = tmp_path / "f.csv", tmp_path / "w.csv", tmp_path / "r99_pmc_fetch_write_size_per_kernel.csv"
    _counter_csv(f, "FETCH_SIZE", [("void k_ba_lm<false, 32>(BaBatch)", 1000.0), ("void k_ba_lm<false, 32>(BaBatch)", 3000.0),
                                   ("k_brief(unsigned char const*, DevDescKp const*)", 10.0)])
    _counter_csv(w, "WRITE_SIZE", [("void k_ba_lm<false, 32>(BaBatch)", 500.0), ("k_brief(unsigned char const*, DevDescKp const*)", 1.0)])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "fetch_write", str(f), str(w), str(out), "cmd"])
    text = out.read_text()
    assert "kernel,calls,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg" in text and "k_ba_lm,2,2000.00,500.00" in text and "k_brief,1,10.00,1.00" in text
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.replace(out, tmp_path / "profiles" / out.name)
    tr = bench.pmc_traffic("k_ba_lm")
    assert tr is not None and abs(tr[0] - 2500.0 * 1024) < 1e-6 and tr[1].endswith(out.name)
    assert bench.pmc_traffic("k_nothing") is None


def test_committed_pmc_summary_is_what_the_bench_reads():
    import bench
    tr = bench.pmc_traffic("k_ba_lm")
    assert tr is not None and tr[1].startswith("profiles/r") and 1e5 < tr[0] < 1e8


def test_algorithmic_work_table_names_the_kernels_of_a_frame():
    import bench
    args = bench.parse([])
    work = bench.algorithmic_work(args)
    # (k_brief: latency-mode contexts; k_blur + k_brief_sample: the throughput-mode contexts of the headline run)
    assert set(work) == {"k_pyramid", "k_fast_harris", "k_brief", "k_blur", "k_brief_sample", "k_knn2", "k_knn2_mfma"}
    assert work["k_knn2"] == ("valu", 16.0 * 2000 * 2000)
    assert work["k_knn2_mfma"] == ("mfma_i8", 512.0 * 2000 * 2000)      # the kernel that runs: an i8 Gram on the matrix cores
    # BA5 is 11.2 MFLOP per LM trial
    assert abs(bench.ba_trial_flops(9386, 2000, 5, False) / 1e6 - 11.2) < 0.3


def test_roofline_side_figures_come_from_the_newest_rounds_profile_files():
    """What bench.py's roofline block reads for the resident solver grid: the matrix-core counters measured ON the grid
    (profiles/r06_pmc_grid_mfma_busy.txt), the HBM-side bytes per window of the newest round (round 6: FETCH_SIZE / WRITE_SIZE passes ON the
    grid, profiles/r06_pmc_fetch_write_size_per_kernel.csv; round 5 had to fall back to the launch-path form of the same cut), the compiler's
    register report of the shipped build."""
    import bench
    gb = bench.pmc_grid_busy()
    assert gb and gb["source"].startswith("profiles/r") and gb["workgroups_per_window"] == 14 and gb["windows"] >= 1000
    busy = gb["SQ_VALU_MFMA_BUSY_CYCLES"] / (gb["resident_cycles"] * gb["workgroups_per_window"] * 4)
    assert 0.03 < busy < 0.3, busy
    lp = bench.pmc_traffic("k_ba_lm_per_window", "r[0-9]*_pmc_launch_path_fetch_write_size.csv")
    assert lp and 1e5 < lp[0] < 1e8
    tg = bench.pmc_traffic("k_ba_service_per_window")
    assert tg and tg[1].startswith("profiles/r06") and 1e6 < tg[0] < 5e7      # bytes per window, measured on the resident grid itself
    res = bench.kernel_resources("k_ba_service<32,2>")
    assert res and res["vgpr"] == 256 and res["source"].startswith("profiles/r06")
    fh = bench.kernel_resources("k_fast_harris")
    assert fh["vgpr"] <= 64 and fh["scratch_bytes_per_lane"] == 0 and fh["occupancy"] == 8


def test_ba_flop_count_credits_each_stage_where_it_runs():
    """SURVEY 8d's F_trial split by stage: the linearisation once per iteration, Schur + factorisation per trial, back-substitution
    + chi2 only for the trials whose step is applied (a trial that ends at the failed factorisation runs neither)."""
    import bench
    E, L, F = 9386, 2000, 5
    full = bench.ba_trial_flops(E, L, F, False)
    # one iteration, one applied trial = one full trial
    assert abs(bench.ba_solve_flops(E, L, F, False, 1, 1, 1) - full) < 1e-6 * full
    # the benchmarked window: 50 iterations, ~85 trials of which ~35 end at the failed factorisation -> ~817 MFLOP, not 85 x 11.3
    w = bench.ba_solve_flops(E, L, F, False, 50, 85.1, 50.0)
    assert 780e6 < w < 850e6 and w < 0.9 * 85.1 * full
    # pose-only: no landmark terms
    assert bench.ba_solve_flops(E, L, F, True, 1, 1, 1) == 330 * E + (6 * F) ** 3 / 3 + 60 * E


def test_sequences_are_rendered_by_worker_processes_like_the_generator_itself():
    """bench.render_sequences (every shard its own sequence, seed 1234 + shard): worker processes `python synth.py render`
    write .npy blocks; what comes back is the generator's output, frame for frame."""
    import bench
    import __graft_entry__ as graft
    args = bench.parse(["--frames", "3", "--width", "160", "--height", "120", "--tex-size", "256"])
    blocks = bench.render_sequences(args, [0, 5])
    synth = graft.load_package().synth
    for sid, block in blocks.items():
        assert block.shape == (3, 120, 160, 3) and block.dtype == np.uint8
        seq = synth.Sequence(160, 120, 3, seed=1234 + sid, tex_size=256)
        for i in range(3):
            assert np.array_equal(block[i], seq.frame(i))
    assert not np.array_equal(blocks[0][0], blocks[5][0])          # different shards, different scenes
    # the defaults are the contract's: 150 frames from a 2048 x 2048 texture
    d = bench.parse([])
    assert d.frames == 150 and d.tex_size == 2048 and d.width == 640 and d.height == 480
