#!/usr/bin/env python3
"""How many CPUs does this box really give one job?  (No GPU involved.)

    python tools/cpu_scaling_probe.py

Prints the CPU count, the cgroup CPU quota if there is one, and the aggregate rate of N processes spinning on a pure
integer loop for N = 1 ... 128: where the rate stops growing is the number of cores the job gets -- which bounds the reader
threads of `count` and the worker sweep of the CPU baseline whatever the code does."""
import multiprocessing as mp
import os
import time


def spin(_):
    t0 = time.perf_counter(); x = 0; n = 0
    while time.perf_counter() - t0 < 1.0:
        for i in range(20000):
            x = (x * 1103515245 + 12345) & 0x7FFFFFFF
        n += 20000
    return n


def main():
    print("os.cpu_count() =", os.cpu_count(), " sched_getaffinity =", len(os.sched_getaffinity(0)))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        try:
            print(f, "=", open(f).read().strip())
        except OSError:
            pass
    base = None
    for n in (1, 8, 16, 24, 32, 48, 64, 96, 128):
        with mp.get_context("fork").Pool(n) as pool:
            pool.map(spin, range(n))
            t0 = time.perf_counter(); tot = sum(pool.map(spin, range(n))); dt = time.perf_counter() - t0
        rate = tot / dt
        base = base or rate
        print("%3d processes: %.1f x one process" % (n, rate / base), flush=True)


if __name__ == "__main__":
    main()
