"""C-ABI surface (no GPU needed to load the library) and host-side logic."""
import ctypes
import os
import re
import sys
import warnings

import numpy as np
import pytest

from conftest import ROOT


def _lib():
    from strique_amd import build, ffi
    if not os.path.exists(ffi.LIB_PATH):
        build.build_lib()
    return ctypes.CDLL(ffi.LIB_PATH)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "strique_hip.h")).read()
    names = sorted(set(re.findall(r"\b(strq_[a-z_0-9]+)\s*\(", header)))
    assert len(names) >= 15
    lib = _lib()
    for n in names:
        assert hasattr(lib, n), n
    assert lib.strq_abi_version() == 12


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the context cannot be created -- and nothing else is tried."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from strique_amd import ffi
    with pytest.raises(ffi.StriqueHipError):
        ffi.Context(0)


def test_product_does_not_import_oracle():
    """The shipped package never imports, includes, links or loads anything under oracle/."""
    bad = re.compile(r"(^\s*(import|from)\s+oracle\b)|(#include\s+[\"<][^\">]*oracle)|(liboracle)|(oracle[/\\][a-z_]+\.(so|c|py)\b.*(CDLL|dlopen|open))", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "strique_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), (dirpath, f)


def test_shard_indices_cover_everything():
    from strique_amd.dist import shard_indices
    cost = np.random.default_rng(0).integers(1, 100, 37)
    parts = [shard_indices(37, r, 4, cost) for r in range(4)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(37))
    loads = [cost[p].sum() for p in parts]
    assert max(loads) - min(loads) <= cost.max()


def test_synth_is_seeded_and_has_the_repeat(pm, cfg):
    from strique_amd import synth
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    t = synth.KmerTable(pm)
    a, sa = synth.make_read(t, 2, 5, 6000, (repeat, prefix, suffix), 30)
    b2, sb = synth.make_read(t, 2, 5, 6000, (repeat, prefix, suffix), 30)
    assert sa == sb and np.array_equal(a, b2) and a.dtype == np.int16
    assert 6 * 5995 <= len(a) <= 9 * 5995
    idx = t.indices(b"ACGTACGTAC")
    assert np.allclose(t.mean[idx], pm.level_means("ACGTACGTAC"))


def test_oracle_does_not_import_the_product():
    """The oracle stands alone: nothing under oracle/ imports, loads or reads anything of strique_amd."""
    import os
    import re
    from conftest import ROOT
    bad = re.compile(r"^\s*(import|from)\s+strique_amd\b|libstrique_hip", re.M)
    for name in os.listdir(os.path.join(ROOT, "oracle")):
        if name.endswith((".py", ".c", ".h")) or name == "Makefile":
            text = open(os.path.join(ROOT, "oracle", name)).read()
            assert not bad.search(text), name


def test_host_stats_equal_numpy():
    """strq_host_stats (float64 reads: no exact histogram on the GPU) against the numpy calls of the reference:
    scipy medfilt(3), np.median, MAD = np.mean(|x - median|) (STRique.py:142-143), np.percentile([1, 99]) and the
    medians of the two tails (STRique.py:152-160).  Bit for bit, sizes around every blocking boundary of numpy's
    pairwise summation and of the selection."""
    import scipy.signal
    from strique_amd import ffi
    rng = np.random.default_rng(99)
    sizes = [1, 2, 3, 7, 8, 9, 100, 127, 128, 129, 1000, 8191, 8192, 8193, 8200, 16384, 16385, 40000, 100003, 284184]
    sigs = []
    for k, n in enumerate(sizes):
        if k % 3 == 0:
            s = rng.normal(90, 12, n)
        elif k % 3 == 1:
            s = np.round(rng.normal(90, 12, n) * 4) / 4            # many ties
        else:
            s = rng.normal(90, 12, n); s[rng.integers(0, n, max(1, n // 50))] = 300.0
        sigs.append(s)
    sigs.append(np.full(500, 42.0))                                 # constant: empty tails
    sigs.append(np.where(np.arange(9000) % 2, 1.0, 2.0))            # two values
    off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
    got = ffi.host_stats(np.concatenate(sigs), off, want_raw=True)

    def tails(x):
        q_lo, q_hi = np.percentile(x, [1, 99])
        m_lo = np.median(x[x < q_lo]); m_hi = np.median(x[x > q_hi])
        return m_lo + (m_hi - m_lo) / 2, (m_hi - m_lo) / 2
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s, g in zip(sigs, got):
            flt = scipy.signal.medfilt(s, 3)
            med = np.median(flt)
            want = [med, np.mean(np.absolute(np.subtract(flt, med)))] + list(tails(flt)) + list(tails(s))
            assert np.array_equal(np.asarray(want), g, equal_nan=True), (len(s), want, g)
    assert np.array_equal(ffi.host_stats(np.concatenate(sigs), off)[:, 4:], np.tile([0.0, 1.0], (len(sigs), 1)))


def test_one_hip_runtime_whatever_the_import_order():
    """libstrique_hip.so and a PyTorch-ROCm wheel both need `libamdhip64.so.7`; strique_amd.ffi decides which copy serves the
    process BEFORE it opens the library, so that a later `import torch` (RCCL through torch.distributed) binds to the same
    one.  Either order must leave exactly one HIP runtime mapped (round 3: library first, torch second = two runtimes and
    a torch.cuda that could not initialise).  No GPU needed: this is the dynamic linker's doing."""
    import subprocess
    import sys
    prog = (
        "import sys; sys.path.insert(0, %r)\n"
        "order = sys.argv[1]\n"
        "if order == 'torch_first': import torch\n"
        "from strique_amd import ffi\n"
        "ffi.load_library()\n"
        "before = ffi.mapped_hip_runtimes()\n"
        "import torch\n"
        "after = ffi.mapped_hip_runtimes()\n"
        "assert len(before) == 1 and after == before, (order, before, after)\n"
        "print(order, after[0])\n") % ROOT
    env = dict(os.environ); env.pop("STRQ_HIP_RUNTIME", None)
    seen = []
    for order in ("lib_first", "torch_first"):
        r = subprocess.run([sys.executable, "-c", prog, order], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        seen.append(r.stdout.split()[-1])
    assert seen[0] == seen[1]


def test_rank_cpu_share_partitions_the_host():
    """strique_amd.dist.rank_cpu_share: every CPU belongs to exactly one rank, hyper-thread siblings stay together, the
    blocks are contiguous in core order (ranks 0..3 of 8 on the first socket)."""
    from strique_amd import dist as sdist
    allowed = set(range(256))
    import builtins
    real_open = builtins.open

    def fake_open(path, *a, **k):          # a 2 x 64-core host with SMT: cpu c and c + 128 share a core
        if isinstance(path, str) and path.startswith("/sys/devices/system/cpu/cpu") and path.endswith("thread_siblings_list"):
            c = int(path.split("/cpu")[-1].split("/")[0]) % 128
            import io
            return io.StringIO("%d,%d\n" % (c, c + 128))
        return real_open(path, *a, **k)

    builtins.open = fake_open
    try:
        shares = [sdist.rank_cpu_share(r, 8, allowed) for r in range(8)]
    finally:
        builtins.open = real_open
    assert sorted(c for s in shares for c in s) == list(range(256))
    for r, s in enumerate(shares):
        assert s == list(range(16 * r, 16 * r + 16)) + list(range(128 + 16 * r, 128 + 16 * r + 16))
    assert sdist.rank_cpu_share(0, 1, {3, 4}) == [3, 4]
    assert sdist.rank_cpu_share(5, 8, {0, 1}) in ([0], [1])          # fewer cores than ranks: shared round-robin


def test_cpu_quota_of_the_control_group(monkeypatch):
    """strique_amd.dist.cpu_quota: cgroup v2 `cpu.max` ("1600000 100000" = 16 CPUs' worth: what the MI355X boxes give a job that sees
    256 CPUs), "max" = no quota, cgroup v1 files as the fallback; effective_cpus caps the affinity count with it."""
    import builtins
    import io
    from strique_amd import dist as sdist
    files = {}
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup/"):
            if path in files:
                return io.StringIO(files[path])
            raise FileNotFoundError(path)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    assert sdist.cpu_quota() is None
    files["/sys/fs/cgroup/cpu.max"] = "1600000 100000\n"
    assert sdist.cpu_quota() == 16.0 and sdist.effective_cpus() == min(16, len(os.sched_getaffinity(0)))
    files["/sys/fs/cgroup/cpu.max"] = "max 100000\n"
    assert sdist.cpu_quota() is None
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "250000"; files["/sys/fs/cgroup/cpu/cpu.cfs_period_us"] = "100000"
    assert sdist.cpu_quota() == 2.5
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "-1"
    assert sdist.cpu_quota() is None


def test_switches_process_wide_table_and_environment(monkeypatch):
    """strq_set_option(NULL, key, value): the process-wide table sits between a context's own options and the environment variable
    of the same name; "" means "not set" whatever the environment says, NULL hands the decision back.  Read through a switch the
    host-only planning hook consults (STRQ_NO_SCREEN -> no screen frame)."""
    lib = _lib()
    lib.strq_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
    lib.strq_get_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32]
    buf = ctypes.create_string_buffer(64)
    out = (ctypes.c_int32 * 6)()
    arr = (ctypes.c_float * 6)(-1, -1, -16, -16, 16, 0)
    plan = lambda: lib.strq_debug_screen_plan(arr, ctypes.c_int32(6), ctypes.c_int32(400000), out)
    monkeypatch.delenv("STRQ_NO_SCREEN", raising=False)
    assert plan() == 1
    assert lib.strq_set_option(None, b"STRQ_NO_SCREEN", b"1") == 0
    lib.strq_get_option(None, b"STRQ_NO_SCREEN", buf, 64)
    assert buf.value == b"1" and plan() == 0
    monkeypatch.setenv("STRQ_NO_SCREEN", "1")
    assert lib.strq_set_option(None, b"STRQ_NO_SCREEN", b"") == 0          # unset for the process although the environment sets it
    assert plan() == 1
    assert lib.strq_set_option(None, b"STRQ_NO_SCREEN", None) == 0         # back to the environment
    assert plan() == 0
    assert lib.strq_set_option(None, b"NOT_A_SWITCH", b"1") != 0


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` started plainly launches its own rank processes -- and exits non-zero, without a JSON line, when fewer
    than N HIP devices are visible: a request for 8 GPUs must never end as an `n_gpus: 1` measurement (scripts/STRique.py:912 `--t`)."""
    import subprocess
    import sys
    import torch
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 8), "--reads", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 2 and b"HIP device" in p.stderr and b'"n_gpus"' not in p.stdout
    # and an external launcher whose world size disagrees with --gpus is refused as well
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--reads", "8", "--no-cpu-baseline"],
                       env=dict(env, RANK="0", WORLD_SIZE="1"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0 and b"WORLD_SIZE" in p.stderr and b'"n_gpus"' not in p.stdout


def test_empirical_noise_fixture_and_generator(pm, cfg):
    """The degraded-read workload: dwell / level-offset / residual pools taken from the bundled real read (tests/golden/
    make_empirical_noise.py), resampled by strique_amd.synth.EmpiricalNoise -- seeded, and with the real read's statistics."""
    from strique_amd import synth
    z = np.load(os.path.join(ROOT, "strique_amd", "data", "empirical_noise.npz"))
    dw = z["dwell"]
    assert 4000 < len(dw) < 5000 and 8.5 < dw.mean() < 10.0 and np.median(dw) == 7 and (dw == 0).sum() > 10
    assert 1.0 < z["level_offset"].std() < 2.0 and 1.8 < z["resid_z"].std() < 2.3
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    t = synth.KmerTable(pm)
    noise = synth.EmpiricalNoise()
    a, sa = synth.make_read(t, 7, 3, 8000, (repeat, prefix, suffix), 40, noise=noise)
    b2, sb = synth.make_read(t, 7, 3, 8000, (repeat, prefix, suffix), 40, noise=noise)
    assert sa == sb and np.array_equal(a, b2) and a.dtype == np.int16
    assert 7.5 * 8000 < len(a) < 11.0 * 8000


def test_bench_prints_a_compact_last_line(tmp_path):
    """The driver keeps an 8 KB tail of bench.py's stdout and parses its LAST line: round 5's record had grown to 25 KB and could not
    be read.  bench.compact_line() of that very record (profiles/r05_bench_default.json) is under 4 KB, is valid JSON, carries
    `roofline` (bound / achieved / peak / frac / traffic) and `cpu_baseline` (value / cores / kind / sample), and no key called
    `frac` in it exceeds 1; the full record goes to a file (bench.write_detail)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    assert len(json.dumps(full)) > 20000
    path = bench.write_detail(full, str(tmp_path / "d" / "detail.json"))
    assert json.load(open(path)) == full
    line = bench.compact_line(full, path)
    s = json.dumps(line, separators=(",", ":"))
    assert len(s) < 4096 and "\n" not in s
    back = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in back, k
    assert back["config"]["workload"].startswith("BASELINE configs[2]")
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"} <= set(back["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(back["cpu_baseline"])
    assert back["roofline_viterbi"]["kernel"].startswith("viterbi_g2_kernel")

    def fracs(d):
        for k, v in d.items():
            if isinstance(v, dict):
                for f in fracs(v):
                    yield f
            elif k == "frac" or k.endswith("_frac"):
                yield v
    assert all(0 <= f <= 1 for f in fracs(back)), list(fracs(back))
    # a record bloated by future fields still fits: the optional blocks go first
    full["data"] = "x" * 3000
    assert len(json.dumps(bench.compact_line(full, path), separators=(",", ":"))) < 4096
