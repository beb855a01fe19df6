"""CPU rehearsal of the GPU parity tests: the product's host driver + kernel SOURCE compiled
against the SIMT emulator in tests/emu (test infrastructure), checked against the oracle.

This exercises kernel logic, the C ABI, the host driver and the Python entry points without a
GPU.  It proves nothing about the real hardware build -- tests/test_gpu_parity.py does that
on an MI355X -- and the emulator library is never loaded by the product package: this module
swaps it into peaksegdisk_amd._native for the duration of its own tests only."""
import ctypes
import os
import subprocess

import pytest

import test_gpu_parity as gp
import test_gpu_round2 as gp2
import test_gpu_round3 as gp3
import test_gpu_round4 as gp4
from conftest import ROOT

EMU_DIR = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def psd():
    import __graft_entry__ as entry
    entry.build_hip()  # the package refuses to import without its HIP library
    subprocess.run(["make", "-s", "-C", EMU_DIR], check=True)
    import peaksegdisk_amd
    from peaksegdisk_amd import _native
    # (PSD_EMU_LIB_OVERRIDE: an emulator build of an experimental source tree, for A/B work)
    emu = _native.declare(ctypes.CDLL(os.environ.get(
        "PSD_EMU_LIB_OVERRIDE", os.path.join(EMU_DIR, "_build", "libpeaksegdisk_emu.so"))))
    real = _native.lib
    _native.lib = emu
    try:
        yield peaksegdisk_amd
    finally:
        _native.lib = real


def test_emu_known_answers(psd, oracle_det, known_answers, tmp_path):
    gp.test_known_answers_byte_identical(psd, oracle_det, known_answers, tmp_path)


def test_emu_error_cases(psd, known_answers, tmp_path):
    gp.test_error_cases(psd, known_answers, tmp_path)


def test_emu_mono27ac(psd, oracle_det, oracle_libm, known_answers, tmp_path, monkeypatch):
    monkeypatch.setattr(gp, "MONO_PENALTIES", ["1952.6", "Inf", "157.994737329317"])
    gp.test_mono27ac_files_and_store_identical(psd, oracle_det, oracle_libm, known_answers,
                                               tmp_path)


def test_emu_arena_equals_oracle_db(psd, oracle_det, tmp_path):
    gp.test_arena_equals_oracle_db(psd, oracle_det, tmp_path)


def test_emu_synthetic(psd, oracle_det, oracle_libm, tmp_path):
    gp.test_synthetic_grid_vs_oracle(psd, oracle_det, oracle_libm, tmp_path, 1500, 12)


def test_emu_python_entry_points(psd, tmp_path):
    gp.test_python_entry_points(psd, tmp_path, with_search=False)


def test_emu_adversarial_spill(psd, oracle_det, tmp_path):
    gp.test_adversarial_increasing_counts_spill(psd, oracle_det, tmp_path, 1500)


def test_emu_newton_step_cap_fallback(oracle_det, tmp_path):
    """The root finders' 100-step fallback (bracket midpoint vs last iterate, fpl:109-120,
    170-183) never triggers on real data, and the kernels keep the bracket bookkeeping out of
    their hot loops, re-solving with it only when the cap is hit.  Build oracle and kernels
    with the cap lowered to 5 steps so that most solves take that path, and compare."""
    import numpy as np
    from conftest import Oracle
    import peaksegdisk_amd  # noqa: F401
    from peaksegdisk_amd import _native, synthetic
    from peaksegdisk_amd.grid import ProblemSet
    subprocess.run(["make", "-s", "-C", EMU_DIR, "all"], check=True)
    emu5 = _native.declare(ctypes.CDLL(os.path.join(EMU_DIR, "_build",
                                                    "libpeaksegdisk_emu_steps5.so")))
    oracle5 = Oracle("det_steps5")
    cs, ce, cnt = synthetic.poisson_coverage(1200, seed=21)
    pens = ["0.5", "40", "3000"]
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens],
                      lib=emu5)
    pset.solve()
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    differs_from_100 = False
    for i, pen in enumerate(pens):
        db5 = str(tmp_path / ("o5_%d.db" % i))
        assert oracle5.solve(bg, pen, db5) == 0
        dbg = str(tmp_path / ("g5_%d.db" % i))
        pset.export_db(i, ce, dbg)
        assert open(dbg, "rb").read() == open(db5, "rb").read(), pen
        db100 = str(tmp_path / ("o100_%d.db" % i))
        assert oracle_det.solve(bg, pen, db100) == 0
        differs_from_100 = differs_from_100 or open(db100, "rb").read() != open(db5, "rb").read()
    assert differs_from_100, "cap of 5 steps did not change anything: fallback not exercised"
    pset.close()


def test_emu_throughput_build(psd, oracle_det, tmp_path, monkeypatch):
    gp.test_throughput_build_identical(psd, oracle_det, tmp_path, monkeypatch, 400, 4)


def test_emu_sequential_envelope_replay(oracle_det, tmp_path):
    """min_env's compaction is only used when every "same function" decision is also a bitwise
    equality; otherwise lane 0 replays the intervals sequentially (min_env_serial), and the
    latency build's specialised step hands such a data point to the general step.  No data set
    has needed that so far, so build the kernels with the replay forced for every envelope
    and compare the whole store with the oracle's."""
    import numpy as np
    import peaksegdisk_amd  # noqa: F401
    from peaksegdisk_amd import _native, synthetic
    from peaksegdisk_amd.grid import ProblemSet
    subprocess.run(["make", "-s", "-C", EMU_DIR, "all"], check=True)
    cs, ce, cnt = synthetic.poisson_coverage(1000, seed=31)
    pens = ["0.7", "60", "5000"]
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    # ... and the other hand-over nothing realistic triggers: a step whose exp/log met a rare
    # argument (the specialised path evaluates them without the branch for those) is redone by
    # the general path; forced for every step with -DPSD_FORCE_RARE
    for variant, build in (("serial", "lat"), ("serial", "thr"), ("rare", "lat"), ("rare", "thr")):
        lib = _native.declare(ctypes.CDLL(os.path.join(EMU_DIR, "_build",
                                                       "libpeaksegdisk_emu_%s.so" % variant)))
        os.environ["PEAKSEG_HIP_VARIANT"] = build
        try:
            pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))],
                              [(0, float(p)) for p in pens], lib=lib)
            pset.solve()
        finally:
            del os.environ["PEAKSEG_HIP_VARIANT"]
        assert pset.kernel_build == build
        for i, pen in enumerate(pens):
            r = pset.result(i)
            assert r.status == 0 and (r.n_serial_env > 0 or variant == "rare")
            want = str(tmp_path / ("o_%d.db" % i))
            if not os.path.exists(want):
                assert oracle_det.solve(bg, pen, want) == 0
            got = str(tmp_path / ("g_%s_%s_%d.db" % (variant, build, i)))
            pset.export_db(i, ce, got)
            assert open(got, "rb").read() == open(want, "rb").read(), (variant, build, pen)
        pset.close()


def test_emu_varied_data_shapes(psd, oracle_det, tmp_path, monkeypatch):
    gp.test_varied_data_shapes(psd, oracle_det, tmp_path, monkeypatch, 3, 7)


def test_emu_grid_properties_small(psd):
    """The property checks of the full-size GPU test (test_gpu_parity.check_grid_properties) on
    a small grid, and determinism of a second solve."""
    import numpy as np
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.poisson_coverage(3000, seed=1)
    pens = synthetic.penalty_grid(6)
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens])
    pset.solve()
    first = gp.check_grid_properties(pset, pens, cs, ce, cnt, 3000)
    pset.solve()
    for i in range(len(pens)):
        start, mean = pset.segments(i)
        assert np.array_equal(start, first[i][0])
        assert np.array_equal(mean.view(np.uint64), first[i][1].view(np.uint64))
    pset.close()


def test_emu_fuzz_tiny_problems(psd, oracle_det, tmp_path):
    gp.test_fuzz_tiny_problems(psd, oracle_det, tmp_path, 150, 6)


def test_emu_resident_search_exact_sequence(psd, oracle_det, known_answers, tmp_path, monkeypatch):
    # host-side search logic: on the build without helper waves (half the fibers to emulate)
    monkeypatch.setenv("PEAKSEG_HIP_VARIANT", "thr")
    gp2.test_resident_search_exact_sequence(psd, oracle_det, known_answers, tmp_path)


def test_emu_search_batch_equals_single_searches(psd, tmp_path, monkeypatch):
    monkeypatch.setenv("PEAKSEG_HIP_VARIANT", "thr")  # host-side logic, as above
    gp2.test_search_batch_equals_single_searches(psd, tmp_path, n_bins=400, with_mono=False)


def test_emu_dir_batch_cache_and_timing(psd, oracle_det, tmp_path):
    gp2.test_dir_batch_cache_and_timing(psd, oracle_det, tmp_path, 500)


def test_emu_spill_pool_and_arena_regrowth(psd, oracle_det, tmp_path, monkeypatch):
    gp2.test_spill_pool_and_arena_regrowth(psd, oracle_det, tmp_path, monkeypatch, 700)


def test_emu_checkpointed_store(psd, oracle_det, tmp_path, monkeypatch):
    gp2.test_checkpointed_store_equals_full_store(psd, oracle_det, tmp_path, monkeypatch, 800,
                                                  (16, 3000))


def test_emu_checkpointed_store_limits(psd, tmp_path, monkeypatch):
    gp2.test_checkpointed_store_region_regrowth_and_limits(psd, tmp_path, monkeypatch, 1500, 0, 1000,
                                                           (("thr", ""), ("lat", "1024")))


def test_emu_mixed_launch(psd, tmp_path, monkeypatch):
    gp2.test_mixed_launch_of_unequal_contigs(psd, tmp_path, monkeypatch, 300, 25, 63)


def test_emu_solve_grid(psd, tmp_path):
    gp3.test_solve_grid_on_device(psd, tmp_path, n_contigs=4, scale=0.0006, n_pen=5,
                                  oracle_every=1)


def test_emu_adversarial_counts(psd, oracle_det, tmp_path):
    gp3.test_adversarial_counts_1e5(psd, oracle_det, tmp_path, n_bins=1200, want_max=0,
                                    spill_from=700)


def test_emu_worst_case_vignette_grid(psd, oracle_det, tmp_path):
    gp3.test_worst_case_vignette_grid(psd, oracle_det, tmp_path, sizes=(10, 100, 300))


def test_emu_sequential_search_synthetic(psd, tmp_path, monkeypatch):
    monkeypatch.setenv("PEAKSEG_HIP_VARIANT", "thr")  # host-side logic, as above
    gp3.test_sequential_search_on_a_long_contig(psd, tmp_path, n_bins=1500, peaks_int=2)


def test_emu_write_failures_dp_branch(psd, tmp_path, monkeypatch):
    monkeypatch.setenv("PSD_TEST_NATIVE_LIB",
                       os.path.join(EMU_DIR, "_build", "libpeaksegdisk_emu.so"))
    gp3.test_write_failures_injected_on_the_dp_branch(psd, tmp_path, n_bins=600)


def test_emu_batch_with_duplicate_problems(psd, oracle_det, tmp_path):
    gp3.test_batch_with_duplicate_problems(psd, oracle_det, tmp_path, n_bins=400)


def test_emu_arena_regrowth_resumes(psd, oracle_det, tmp_path, monkeypatch):
    gp3.test_arena_regrowth_resumes_instead_of_repeating(psd, oracle_det, tmp_path, monkeypatch,
                                                         n_bins=1000)


def test_emu_checkpointed_store_large_penalties(psd, monkeypatch):
    gp3.test_checkpointed_store_large_penalties_single_launch(psd, monkeypatch, n_bins=1500)


def test_emu_arena_grows_while_the_kernel_runs(psd, oracle_det, tmp_path, monkeypatch):
    # (the emulator runs the kernel on the calling thread; the thread that adds blocks is real)
    gp3.test_arena_grows_while_the_kernel_runs(psd, oracle_det, tmp_path, monkeypatch, n_bins=1500,
                                               block_log2="12")


def test_emu_search_for_most_peaks_minus_one(psd, tmp_path, monkeypatch):
    from peaksegdisk_amd import synthetic
    monkeypatch.setenv("PEAKSEG_HIP_VARIANT", "thr")  # host-side logic, as above
    cs, ce, cnt = synthetic.poisson_coverage(700, seed=41)
    text = "".join("chrSynth\t%d\t%d\t%d\n" % t for t in zip(cs.tolist(), ce.tolist(), cnt.tolist()))
    gp4.test_search_for_most_peaks_minus_one(psd, tmp_path, text=text, most_peaks=None)


def test_emu_search_with_too_many_peaks_on_six_points(psd, tmp_path):
    gp4.test_search_with_too_many_peaks_on_six_points(psd, tmp_path)


def test_emu_parked_long_functions_survive_several_exhaustions(psd, oracle_det, tmp_path,
                                                               monkeypatch):
    gp4.test_parked_long_functions_survive_several_exhaustions(psd, oracle_det, tmp_path,
                                                               monkeypatch, n_bins=1200)


def test_emu_packed_build_parity_planner_and_hand_over(psd, oracle_det, tmp_path, monkeypatch):
    # (a device of 32 CUs: 128 problems fill the throughput build, 192 the packed one)
    monkeypatch.setenv("PSD_EMU_CUS", "32")
    gp4.test_packed_build_parity_planner_and_hand_over(psd, oracle_det, tmp_path, monkeypatch,
                                                       n_contigs=40, n_bins=100, adv_bins=800,
                                                       many=(270, 200))


def test_emu_spill_pool_exhaustion_parks(psd, oracle_det, tmp_path, monkeypatch):
    gp4.test_spill_pool_exhaustion_parks_instead_of_starting_over(psd, oracle_det, tmp_path,
                                                                  monkeypatch, n_bins=800)


def test_emu_knob_combinations_on_a_mixed_set(psd, oracle_det, tmp_path, monkeypatch):
    gp4.test_knob_combinations_on_a_mixed_set(psd, oracle_det, tmp_path, monkeypatch,
                                              n_poisson=600, n_increasing=500, n_shapes=2)


def test_emu_division_near_a_midpoint(psd, oracle_det, tmp_path, monkeypatch):
    # (the host divides exactly: this pins the data set and the oracle's stores, not psd_div)
    gp4.test_division_near_a_midpoint(psd, oracle_det, tmp_path, monkeypatch)

