"""Multi-GPU plumbing: reads shard across ranks, results are gathered once at the end.

Every (read, target) is independent (reference scripts/STRique.py:702-704; the reference itself
spreads reads over worker processes, :743-746), so there is no collective on the data path.  The
only exchange is the final gather of fixed-size result records to rank 0 -- with the "nccl" backend
that is RCCL over xGMI (~64 B per read).
"""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_indices(n_items, rank, world, cost=None):
    """Indices handled by `rank`: items sorted by descending cost (read length) dealt round-robin,
    which balances the sequential DP work per GPU.  Deterministic on every rank."""
    order = np.arange(n_items) if cost is None else np.argsort(-np.asarray(cost), kind="stable")
    return np.sort(order[rank::world])


def cpu_quota():
    """CPUs' worth of time the control group of this process may use (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`), or None
    when there is no quota.  A container can show 256 CPUs and be throttled to 16: thread pools sized by the CPU count then only
    add contention (measured on the MI355X boxes: `tools/cpu_scaling_probe.py`, 16 processes 15.9 x, 128 processes 11.1 x)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1.0, float(quota) / float(period))
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            return max(1.0, quota / period)
    except (OSError, ValueError):
        pass
    return None


def effective_cpus():
    """CPUs this process can really keep busy: its affinity mask, capped by the control group's quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cpu_quota()
    return max(1, int(min(n, q))) if q else n


def _core_groups(allowed):
    """Logical CPUs of `allowed` grouped by physical core (hyper-thread siblings together), cores in ascending order of
    their first CPU.  /sys topology when readable, one CPU per core otherwise."""
    groups, seen = [], set()
    for cpu in sorted(allowed):
        if cpu in seen:
            continue
        sib = {cpu}
        try:
            txt = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu).read().strip()
            for part in txt.split(","):
                lo, _, hi = part.partition("-")
                sib.update(range(int(lo), int(hi or lo) + 1))
        except (OSError, ValueError):
            pass
        sib = sorted(sib & set(allowed))
        seen.update(sib)
        groups.append(sib)
    return groups


def rank_cpu_share(local, local_world, allowed=None):
    """The CPUs of rank `local` out of `local_world` ranks on this host: a contiguous block of physical cores (ranks
    0 .. w/2-1 land on the first socket of a two-socket host, next to GPUs 0 .. w/2-1) with their siblings.
    Every allowed CPU belongs to exactly one rank; a host with fewer cores than ranks shares them round-robin."""
    if allowed is None:
        allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    groups = _core_groups(allowed)
    if local_world <= 1 or not groups:
        return sorted(allowed)
    if len(groups) < local_world:
        return groups[local % len(groups)]
    lo, hi = len(groups) * local // local_world, len(groups) * (local + 1) // local_world
    return sorted(c for g in groups[lo:hi] for c in g)


def pin_rank_cpus(local=None, local_world=None):
    """Pin this rank process -- and with it the library's upload, order-statistic and reader threads -- to its share of
    the host's CPUs (256 CPUs / 8 ranks on the MI355X box: without it eight ranks size their pools for the whole
    machine and migrate across sockets).  sched_setaffinity acts on one thread (threads created later inherit it), so every
    thread the process already has (/proc/self/task) gets the mask as well; callers still pin BEFORE they start
    torch.distributed.  STRQ_NO_PIN=1 leaves the affinity alone.  Returns the CPUs, or None."""
    if os.environ.get("STRQ_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    if local is None:
        local = int(os.environ.get("LOCAL_RANK", "0"))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if local_world <= 1:
        return None
    cpus = rank_cpu_share(local, local_world)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
            except (OSError, ValueError):
                pass
    except OSError:
        pass
    return cpus


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def any_rank(flag, device=None):
    """True on every rank when `flag` is set on at least one (one tiny all_reduce; `count` uses it so that a rank
    whose device failed does not leave the others blocked in the final gather)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(flag)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    if dev == "cuda" and flag:
        dev_ok = False
        try:
            t = torch.tensor([1], dtype=torch.int32, device="cuda"); dev_ok = True
        except Exception:
            pass
        if not dev_ok:
            raise SystemExit(3)      # this rank cannot even talk to its GPU: the launcher tears the job down
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def gather_results(records, index, n_total, mods=None, device=None, group=None):
    """The one collective of a run, used by `bench.py` and by `count` alike: every rank contributes the
    fixed-size result records of the items it processed (structured numpy array), their global
    positions `index` and, optionally, one byte string per record (`mods`: the modification patterns,
    variable length) in a byte pool.  Three all_gathers of padded tensors (sizes, records + positions,
    pool); with the "nccl" backend that is RCCL over xGMI, ~100 B per read.
    `group`: the process group the three all_gathers run on (default: the default group) -- bench.py keeps gloo as its control
    plane and gathers over a second, RCCL group.
    Returns (records, mods) for all n_total items on rank 0, (None, None) elsewhere."""
    import torch
    import torch.distributed as dist
    records = np.ascontiguousarray(records)
    index = np.asarray(index, np.int64)
    if mods is not None:
        blobs = [m.encode() if isinstance(m, str) else bytes(m) for m in mods]
        lens = np.array([len(b) for b in blobs], np.int64)
        pool = np.frombuffer(b"".join(blobs), np.uint8)
    else:
        lens = np.zeros(len(records), np.int64); pool = np.zeros(0, np.uint8)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        out = np.zeros(n_total, dtype=records.dtype)
        out[index] = records
        out_m = None
        if mods is not None:
            out_m = [""] * n_total
            for i, m in zip(index, mods):
                out_m[int(i)] = m if isinstance(m, str) else bytes(m).decode()
        return out, out_m
    world = dist.get_world_size(group)
    itemsize = records.dtype.itemsize
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    sizes = torch.tensor([len(records), len(pool)], dtype=torch.int64, device=dev)
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = [t.cpu().numpy() for t in all_sizes]
    cap = max(1, max(int(t[0]) for t in all_sizes)); pool_cap = max(1, max(int(t[1]) for t in all_sizes))
    # one row per record: [payload bytes | position (8) | pattern length (8)]
    row = itemsize + 16
    payload = np.zeros((cap, row), np.uint8)
    if len(records):
        payload[:len(records), :itemsize] = records.view(np.uint8).reshape(len(records), itemsize)
        payload[:len(records), itemsize:itemsize + 8] = index.view(np.uint8).reshape(-1, 8)
        payload[:len(records), itemsize + 8:] = lens.view(np.uint8).reshape(-1, 8)
    pool_pad = np.zeros(pool_cap, np.uint8); pool_pad[:len(pool)] = pool
    t_pay = torch.from_numpy(payload).to(dev); t_pool = torch.from_numpy(pool_pad).to(dev)
    pays = [torch.empty_like(t_pay) for _ in range(world)]; pools = [torch.empty_like(t_pool) for _ in range(world)]
    dist.all_gather(pays, t_pay, group=group)
    dist.all_gather(pools, t_pool, group=group)
    if dist.get_rank() != 0:
        return None, None
    out = np.zeros(n_total, dtype=records.dtype)
    out_m = [""] * n_total if mods is not None else None
    for r in range(world):
        k = int(all_sizes[r][0])
        if not k:
            continue
        pr = pays[r][:k].cpu().numpy()
        ii = np.ascontiguousarray(pr[:, itemsize:itemsize + 8]).view(np.int64).reshape(-1)
        ll = np.ascontiguousarray(pr[:, itemsize + 8:]).view(np.int64).reshape(-1)
        out[ii] = np.ascontiguousarray(pr[:, :itemsize]).reshape(-1).view(records.dtype)
        if out_m is not None:
            pl = pools[r].cpu().numpy().tobytes()
            pos = 0
            for i, ln in zip(ii, ll):
                out_m[int(i)] = pl[pos:pos + int(ln)].decode(); pos += int(ln)
    return out, out_m


def gather_records(records, index, n_total, device=None, group=None):
    """Records only (no byte strings): see gather_results."""
    return gather_results(records, index, n_total, None, device, group)[0]
