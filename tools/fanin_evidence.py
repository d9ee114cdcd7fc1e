#!/usr/bin/env python3
"""fanin_evidence.py [out.json] -- GPU box: WHY host fan-in collapses at sixteen host threads.

Runs mbelib-neo_amd/host_bench (queue mode and sessions from 1, 4 and 16 host threads) three times -- unpinned, pinned to 4 cores,
pinned to 16 cores (taskset) -- and records around every run what the kernel's CPU controller did to the container:
  * cgroup cpu.max (the quota) and cpu.stat deltas: nr_periods, nr_throttled, throttled_usec -- whole scheduling periods in which
    the container was stopped because it had used up its quota;
  * the child's voluntary / involuntary context switches (getrusage RUSAGE_CHILDREN deltas) and its user + system CPU time.
If sixteen threads lose against four because of the quota, nr_throttled / throttled_usec grow in the unpinned run and the rate with
16 threads ON 4 CORES is no worse than unpinned; if it were the HIP runtime's lock, pinning would not matter and nothing would be
throttled.  Prints one JSON object (and writes it to out.json)."""
import json
import os
import resource
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_first(paths):
    for p in paths:
        try:
            return open(p).read()
        except OSError:
            continue
    return None


def cpu_stat():
    txt = read_first(["/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"])
    out = {}
    for line in (txt or "").splitlines():
        k, _, v = line.partition(" ")
        try:
            out[k] = int(v)
        except ValueError:
            pass
    return out


def quota():
    v2 = read_first(["/sys/fs/cgroup/cpu.max"])
    if v2:
        return {"cpu.max": v2.strip()}
    q = read_first(["/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu,cpuacct/cpu.cfs_quota_us"])
    p = read_first(["/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu,cpuacct/cpu.cfs_period_us"])
    return {"cfs_quota_us": q.strip() if q else None, "cfs_period_us": p.strip() if p else None}


def main():
    import mbelib_neo_amd  # noqa: F401  (the libraries must be built)
    from mbelib_neo_amd import framegen

    streams = 65536
    frames = framegen.imbe_clean_voiced_frames(streams, framegen.rng_for(0xBE0000))
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(frames.tobytes())
        path = f.name
    exe = os.path.join(ROOT, "mbelib-neo_amd", "host_bench")
    tables = os.path.join(ROOT, "mbelib-neo_amd", "data", "mbx_tables.bin")
    ncpu = len(os.sched_getaffinity(0))
    result = {"host_cpus_visible": ncpu, "quota": quota(), "runs": []}
    pins = [("unpinned", None), ("pinned_4_cores", "0-3"), ("pinned_16_cores", "0-15")]
    try:
        for name, cpus in pins:
            cmd = [exe, tables, path, "0"]
            if cpus:
                cmd = ["taskset", "-c", cpus] + cmd
            s0, r0, t0 = cpu_stat(), resource.getrusage(resource.RUSAGE_CHILDREN), time.time()
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HB_THREADS="1,4,16"))
            s1, r1, t1 = cpu_stat(), resource.getrusage(resource.RUSAGE_CHILDREN), time.time()
            run = {"name": name, "taskset": cpus, "rc": out.returncode, "wall_s": t1 - t0,
                   "cpu_stat_delta": {k: s1.get(k, 0) - s0.get(k, 0) for k in s1},
                   "voluntary_ctx_switches": r1.ru_nvcsw - r0.ru_nvcsw, "involuntary_ctx_switches": r1.ru_nivcsw - r0.ru_nivcsw,
                   "cpu_seconds": (r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime)}
            try:
                hb = json.loads(out.stdout.strip().splitlines()[-1])
                run["threads"] = hb.get("threads")
                run["queue_resident_frames_per_s_by_threads"] = hb.get("queue_resident_frames_per_s_by_threads")
                run["session_pinned_frames_per_s_by_threads"] = hb.get("session_pinned_frames_per_s_by_threads")
            except Exception as e:  # noqa: BLE001
                run["error"] = (str(e) + " | " + out.stderr[-300:])[:500]
            result["runs"].append(run)
    finally:
        os.unlink(path)
    txt = json.dumps(result, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
