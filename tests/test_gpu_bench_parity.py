"""Parity of exactly what bench.py times: reads of its own configs[2] batch (50 kb, C9orf72 GGGGCC x
{200, 500, 1000, 1500, 2000}, both strands) through the production launch geometry -- four waves per alignment
sharing one float32 score table, align_forward_seg_kernel<15, 6, false, 4, 4> -- at the initial 8192-column
overlap, at the overlap adapted to the previous sub-batch's scores, and on a batch the overlap was NOT tuned on.
All six fields of every row against the CPU oracle (reference src/align_raw.h:106-158, scripts/STRique.py:581-618)."""
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT
import oracle_pool

pytestmark = pytest.mark.gpu


def _fresh_counter(pm, cfg, targets):
    from strique_amd.counter import repeatCounter
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    rc.add_target("c9orf72", *targets["c9orf72"])
    return rc


def _bench_batch(pm, cfg, n, first):
    sys.path.insert(0, ROOT)
    import bench
    return bench.make_batch(pm, cfg, n, 50000, first)


@pytest.mark.parametrize("mode", ["coarse", "fine"])
def test_benchmarked_reads_through_the_screen_all_fields(pm, cfg, targets, mode):
    """bench.py's own reads on the path its default line runs: an upper-bound screen over the whole reads -- the coarse one (two
    flank rows per DP row, both flanks of a read per wave: the default) or the fine one -- and the float32 DP over the windows it
    leaves (one wave per window).  Every field of every row equals the oracle's."""
    sigs, strands, nreps = _bench_batch(pm, cfg, 40, 0)
    rc = _fresh_counter(pm, cfg, targets)
    if mode == "fine":
        rc.ctx.set_option("STRQ_SCREEN_MODE", "fine")
    got = rc.detect_batch([("c9orf72", s, st) for s, st in zip(sigs, strands)])
    scr = rc.ctx.last_screen(); geo = rc.ctx.last_geometry(); redo = rc.ctx.last_second_round()
    rc.ctx.close()
    assert scr["mode"] == mode and scr["screened"] == 80 and scr["windowed"] == 80 and scr["scale"] == (512 if mode == "coarse" else 1024), scr
    assert scr["window_columns"] < (0.05 if mode == "coarse" else 0.02) * sum(2 * len(s) for s in sigs), scr          # (coarse: both looks)
    # (coarse: alignments whose first look missed its certificate take the second look -- a handful; fine: none)
    assert redo[0] <= (8 if mode == "coarse" else 0), (geo, redo)
    target = targets["c9orf72"]
    want = oracle_pool.detect_many([(s, st, target) for s, st in zip(sigs, strands)])
    for i, (w, a) in enumerate(zip(want, got)):
        assert tuple(a[:6]) == tuple(w[:6]), (i, a, w)
        assert abs(a[0] - nreps[i]) <= 2


def test_benchmarked_kernel_instance_all_fields(pm, cfg, targets, monkeypatch):
    monkeypatch.setenv("STRQ_CLASS_MIN", "1")          # 64 alignments do not fill the chip: keep them on four waves per alignment anyway
    monkeypatch.setenv("STRQ_NO_SCREEN", "1")          # the float32 pass over whole reads (what runs when the screen is paused or cannot prune)
    sigs1, strands1, nreps1 = _bench_batch(pm, cfg, 32, 0)          # bench.py's rank-0 batch, reads 0..31
    sigs2, strands2, nreps2 = _bench_batch(pm, cfg, 16, 4096)       # reads of another step's batch
    assert set(strands1) == {"+", "-"} and set(nreps1) == {200, 500, 1000, 1500, 2000}
    b1 = [("c9orf72", s, st) for s, st in zip(sigs1, strands1)]
    b2 = [("c9orf72", s, st) for s, st in zip(sigs2, strands2)]
    rc = _fresh_counter(pm, cfg, targets)
    production = dict(waves_per_alignment=4, tables_per_cu=4, wpe=4, packed=0)

    first = rc.detect_batch(b1)                        # initial overlap
    g1 = rc.ctx.last_geometry()
    second = rc.detect_batch(b1)                       # overlap adapted to this batch's own scores
    g2 = rc.ctx.last_geometry()
    third = rc.detect_batch(b2)                        # overlap adapted to a different batch
    g3 = rc.ctx.last_geometry()
    rc.ctx.close()
    for g in (g1, g2, g3):
        assert {k: g[k] for k in production} == production and g["rows_per_lane"] in (14, 15), g
    assert g1["overlap_first"] == 8192 and g1["overlap_worst"] > 14000
    assert g2["overlap_first"] != 8192 and g3["overlap_first"] == g2["overlap_first"], (g1, g2, g3)      # the adapted overlap is live

    target = targets["c9orf72"]
    want = oracle_pool.detect_many([(s, st, target) for s, st in zip(sigs1 + sigs2, strands1 + strands2)])
    for i, (w, a, b) in enumerate(zip(want[:32], first, second)):
        assert tuple(a[:6]) == tuple(w[:6]), (i, a, w)
        assert tuple(b[:6]) == tuple(w[:6]), (i, b, w)
        assert abs(a[0] - nreps1[i]) <= 2
    for i, (w, a) in enumerate(zip(want[32:], third)):
        assert tuple(a[:6]) == tuple(w[:6]), (i, a, w)
        assert abs(a[0] - nreps2[i]) <= 2


def _run_bench(extra, world=1, port="29541", timeout=900):
    """Runs bench.py; returns the JSON lines per rank and the FULL records (bench.py --detail) -- after holding the printed line
    to the compact contract: one line, under 4 KB, carrying the keys the driver parses."""
    import json
    import subprocess
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="strq_bench_"), "detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra + ["--detail", detail]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world > 1:
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(world))
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)) if world > 1 else env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(o[0].decode()[-2000:], o[1].decode()[-2000:]) for o in outs]
    lines = [[ln for ln in o[0].decode().splitlines() if ln.startswith("{")] for o in outs]
    for l in lines:
        if l:
            assert len(l[0]) < 4096, len(l[0])
            c = json.loads(l[0])
            assert {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "roofline", "config", "check_ok", "detail"} <= set(c), sorted(c)
            full = json.load(open(c["detail"]))
            assert c["value"] == pytest.approx(full["value"], rel=1e-5) and c["n_gpus"] == full["n_gpus"]
    return lines, [json.load(open(json.loads(l[0])["detail"])) for l in lines if l]


def test_bench_two_ranks_on_one_gpu():
    """bench.py's own N > 1 path (reference scripts/STRique.py:733-746 spreads reads over worker
    processes): two rank processes sharing HIP device 0, gloo for the gather -- rank 0 prints the one JSON line,
    the collective saw two ranks, both ranks' reads are in `value`, and the spot check compares whole rows."""
    lines, recs = _run_bench(["--gpus", "2", "--backend", "gloo", "--share-device", "--reads", "64", "--steps", "2", "--warmup", "1",
                              "--no-cpu-baseline", "--check", "1", "--synth-workers", "2"], world=2)
    assert len(lines[0]) == 1 and len(lines[1]) == 0          # exactly one line, from rank 0
    r = recs[0]
    assert r["n_gpus"] == 2 and r["world_size_seen_by_the_collective"] == 2 and r["scaling"] == "weak"
    assert r["value"] > 0 and abs(r["value"] - 2 * 64 * 2 / (r["ms_per_step"] * 2 / 1e3)) < 1e-6 * r["value"]
    assert r["check_ok"] and r["check"] and all(c["all_fields_equal"] for c in r["check"])
    assert r["vs_baseline"] is None and "cpu_baseline" not in r


@pytest.mark.parametrize("world", [4, 8])
def test_bench_four_and_eight_ranks_on_one_gpu(world, tmp_path, pm, cfg):
    """The N = 4 and N = 8 launches the driver's scaling run makes, on the one GPU of this box: rank processes sharing HIP
    device 0, gloo for the collectives.  One JSON line; the single final gather saw N ranks and delivered steps x N x reads rows;
    every rank's rows arrived intact (digests) and EQUAL what one process computes for the same reads; every rank is pinned to
    its own share of the CPUs and stays within a bounded host footprint."""
    import bench
    dump = str(tmp_path / "rows.npy")
    reads, steps, warmup, read_nt = 64, 2, 1, 50000
    # N = 4: weak scaling, the job's reads made once (every rank 1 / N of the pool, all ranks run all of it in rotated orders);
    # N = 8: strong scaling (--total-reads 506 -> 64 per GPU and step), every rank its own reads
    shared = world == 4
    mode = ["--reads", str(reads)] if shared else ["--scaling", "strong", "--total-reads", str(8 * reads - 6), "--private-reads"]
    lines, recs = _run_bench(["--gpus", str(world), "--backend", "gloo", "--share-device", "--steps", str(steps), "--warmup", str(warmup),
                              "--read-nt", str(read_nt), "--no-cpu-baseline", "--check", "1", "--synth-workers", "1", "--dump-rows", dump] + mode,
                             world=world, port=str(29560 + world), timeout=1500)
    assert [len(l) for l in lines] == [1] + [0] * (world - 1)
    r = recs[0]
    assert r["n_gpus"] == world and r["world_size_seen_by_the_collective"] == world and r["scaling"] == ("weak" if shared else "strong") and r["check_ok"]
    assert r["config"]["reads_per_gpu_per_step"] == reads and r["config"]["total_reads"] == (None if shared else 8 * reads - 6)
    assert len(coll_ms := r["collective"]["ranks"]["ms_per_step_per_rank"]) == world and max(coll_ms) <= r["ms_per_step"] * 1.001
    coll = r["collective"]
    assert coll["rows_on_rank_0"] == steps * world * reads and "once" in coll["when"]
    ranks = coll["ranks"]
    assert ranks["rows_equal_every_ranks_digest"] is True and ranks["hip_device_per_rank"] == [0] * world
    assert max(ranks["peak_host_rss_gb_per_rank"]) < 6.0, ranks
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= world:
        assert sum(ranks["cpus_pinned_per_rank"]) == ncpu and min(ranks["cpus_pinned_per_rank"]) >= ncpu // world - 1, ranks
    assert abs(r["value"] - world * reads * steps / (r["ms_per_step"] * steps / 1e3)) < 1e-6 * r["value"]
    # the same reads through one process.  Private reads: rank k's batches are reads [k * nb * reads, (k + 1) * nb * reads) of the recipe;
    # one pool: its batches are the pool's, starting with batch k mod nb, the reads of a batch rotated by k * reads / N
    table = np.load(dump)
    counter = _fresh_counter(pm, cfg, {"c9orf72": tuple(cfg["repeat"]["c9orf72"][3:6])})
    nb = r["config"]["distinct_batches_per_gpu"]
    assert nb == (3 if shared else 2)
    pool = bench.make_batch(pm, cfg, nb * reads, read_nt, 0) if shared else None
    for rank in range(world):
        if shared:
            order = np.concatenate([((b + rank) % nb) * reads + (np.arange(reads) + rank * reads // world) % reads for b in range(nb)])
            sigs = [pool[0][g] for g in order]; strands = [pool[1][g] for g in order]
        else:
            sigs, strands, _ = bench.make_batch(pm, cfg, nb * reads, read_nt, rank * nb * reads)
        for j, k in enumerate(range(warmup, warmup + steps)):
            bi = k % nb
            got = counter.detect_batch([("c9orf72", s, st) for s, st in zip(sigs[bi * reads:(bi + 1) * reads], strands[bi * reads:(bi + 1) * reads])])
            part = table[(j * world + rank) * reads:(j * world + rank + 1) * reads]
            for g, t in zip(got, part):
                assert (g[0], g[1], g[2], g[3], g[4], g[5]) == (int(t["count"]), float(t["score_prefix"]), float(t["score_suffix"]), float(t["log_p"]), int(t["offset"]), int(t["ticks"]))


@pytest.mark.parametrize("screen", [True, False])
def test_bench_line_single_gpu_small(screen, monkeypatch):
    """The default single-GPU line at a reduced batch (512 reads so that four waves per alignment is what runs):
    roofline + host leg present, three distinct batches rotated, rows of the host-buffer leg equal the resident
    run's, the spot check covers all six fields on reads of different batches.  With the upper-bound screen (the default) the
    line's roofline is the screen kernel's and the float32 kernel's figures sit under `exact_pass`; without, as in round 3."""
    if not screen:
        monkeypatch.setenv("STRQ_NO_SCREEN", "1")
    lines, recs = _run_bench(["--reads", "512", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--check", "2", "--host-leg-batches", "2", "--leg-steps", "2"]
                             + ([] if screen else ["--no-legs"]))
    r = recs[0]
    roof = r["roofline"]
    assert roof["bound"] == "valu"
    if screen:
        # the A/B legs of the same run, on the same resident batches: no screen, the fine screen, degraded reads -- each with its roofline
        legs = r["legs"]
        assert set(legs) == {"serial", "no_screen", "fine_screen", "degraded"}
        assert legs["no_screen"]["screen"]["alignments_screened_per_step"] == 0 and re.match(r"align_forward_seg_kernel<14, 6, false, 4", legs["no_screen"]["roofline"]["kernel"])
        assert legs["fine_screen"]["roofline"]["kernel"].startswith("align_screen1_kernel") and legs["fine_screen"]["screen"]["mode"] == "fine"
        assert r["value_no_screen"] == legs["no_screen"]["value"] > 0 and r["value_degraded"] == legs["degraded"]["value"] > 0
        assert legs["degraded"]["check"]["all_fields_equal"] and legs["degraded"]["planted_count_recovered"]["of"] == 1024
        assert r["roofline_viterbi"]["kernel"].startswith("viterbi_g2_kernel") and 0 < r["roofline_viterbi"]["frac"] < 1
        assert r["host"]["peak_host_rss_gb_per_rank"] > 0
    if screen:
        # two sub-batches in flight: the previous step's Viterbi launch shares the SIMDs with the screen; the serial leg has the kernel alone
        assert roof["kernel"].startswith("align_screen3_kernel") and legs["serial"]["roofline"]["kernel"] == "align_screen3_kernel"
        assert roof["co_running"]["ms_under_this_launch_per_step"] > 0 and roof["achieved"] > roof["achieved_own"] > 0
        assert 0 < roof["kernel_alone"]["frac"] < 1 and r["value_serial"] == legs["serial"]["value"] > 0
        assert legs["serial"]["overlap_ms_per_step"]["under_alignment_stage_ms"] == 0 and r["roofline_viterbi"]["co_running_share"] > 0.5
        assert roof["screen_mode"] == "coarse" and roof["flank_rows_per_dp_row"] == 3 and roof["with_windows"] == roof["alignments_screened_per_step"] == 1024
        assert re.match(r"align_forward_seg_kernel<14, 6, false, 1, 2, false, true>", roof["exact_pass"]["kernel"])
        assert roof["window_columns_over_columns_of_the_reads"] < 0.02
    else:
        assert re.match(r"align_forward_seg_kernel<14, 6, false, 4, [34], false, true>", roof["kernel"])
        assert len(roof["overlap_columns_per_step"]) == 3 and roof["overlap_columns_per_step"][-1] < roof["overlap_worst_case"]
    assert 0 < roof["useful_frac"] < roof["frac"] < 1
    assert r["config"]["distinct_batches_per_gpu"] == 3
    assert r["check_ok"] and len(r["check"]) == 2 and {c["batch"] for c in r["check"]} == {0, 1}
    assert r["host_buffers"]["same_rows_as_resident_run"] and r["host_inclusive_reads_per_s"] > 0


def test_bench_launches_its_own_ranks_from_a_clean_environment(tmp_path):
    """`python bench.py --gpus 4 ...` with NO launcher and no RANK / WORLD_SIZE in the environment (how the driver calls the 1-GPU
    line, and so probably the N-GPU one): bench.py starts four fresh rank processes itself, relays rank 0's one JSON line, and the
    collective saw four ranks.  Asking for more GPUs than the box has (without --share-device) exits non-zero and prints no line --
    never an `n_gpus: 1` line for an 8-GPU request."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--share-device", "--backend", "gloo", "--reads", "64", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--check", "1", "--synth-workers", "1"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096 and p.stdout.decode().rstrip().splitlines()[-1] == lines[0]
    c = json.loads(lines[0])
    assert c["collective"]["backend"] == "gloo" and c["collective"]["rows_equal_every_ranks_digest"] is True
    r = json.load(open(c["detail"]))
    assert r["n_gpus"] == 4 == r["world_size_seen_by_the_collective"] and r["check_ok"]
    coll = r["collective"]
    assert coll["backend"] == "gloo" and "bench.py itself" in coll["launched_by"] and coll["rows_on_rank_0"] == 2 * 4 * 64
    assert coll["ranks"]["rows_equal_every_ranks_digest"] is True and max(coll["ranks"]["peak_host_rss_gb_per_rank"]) < 6.0
    # more GPUs than the box has
    import torch
    have = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 7), "--reads", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0 and not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert "HIP device" in p.stderr.decode()


def test_bench_gather_falls_back_to_gloo_when_rccl_refuses():
    """Two ranks on ONE device with `--backend nccl`: RCCL refuses ("Duplicate GPU detected") -- the job agrees on that over its gloo
    control plane, gathers over gloo, and says so in the line (`collective.backend`, `collective.nccl_error`) instead of dying."""
    lines, recs = _run_bench(["--gpus", "2", "--backend", "nccl", "--share-device", "--reads", "64", "--steps", "2", "--warmup", "1",
                              "--no-cpu-baseline", "--check", "1", "--synth-workers", "2"], world=2, port="29547")
    assert len(lines[0]) == 1 and len(lines[1]) == 0
    r = recs[0]; coll = r["collective"]
    assert r["n_gpus"] == 2 and r["world_size_seen_by_the_collective"] == 2 and r["check_ok"]
    assert coll["backend"] == "gloo" and coll["backend_requested"] == "nccl" and coll["nccl_error"]
    assert coll["ranks"]["rows_equal_every_ranks_digest"] is True and coll["rows_on_rank_0"] == 2 * 2 * 64
