#!/bin/bash
# fanin_trace.sh [threads,threads,threads ...] -- GPU box: host_bench's N-thread sections only, with the shim's flush trace on
# (MBE_NEO_TRACE_FLUSH=1: per-thread mean of preparation / issue / wait / scatter per flush on stderr).  Development aid.
cd "$(dirname "$0")/.."
python - "$@" <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from mbelib_neo_amd import framegen
frames = framegen.imbe_clean_voiced_frames(65536, framegen.rng_for(0xBE0000))
f = tempfile.NamedTemporaryFile(suffix=".bin", delete=False); f.write(frames.tobytes()); f.close()
for spec in sys.argv[1:] or ["1,4,16"]:
    envs, _, threads = spec.rpartition(":")
    env = dict(os.environ, HB_THREADS=threads, HB_THREADS_ONLY="1", MBE_NEO_TRACE_FLUSH="1")
    for a in filter(None, envs.split(";")):
        k, _, v = a.partition("="); env[k] = v
    def cpu_stat():
        try:
            return {k: int(v) for k, _, v in (l.partition(" ") for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())}
        except (OSError, ValueError):
            return {}
    s0 = cpu_stat()
    r = subprocess.run(["mbelib-neo_amd/host_bench", "mbelib-neo_amd/data/mbx_tables.bin", f.name, "0"], capture_output=True, text=True, env=env)
    s1 = cpu_stat()
    print("==", spec, r.stdout.strip()[:400])
    print("   cgroup cpu.stat delta:", {k: s1[k] - s0.get(k, 0) for k in ("nr_periods", "nr_throttled", "throttled_usec", "usage_usec") if k in s1})
    lines = [l for l in r.stderr.splitlines() if "flush trace" in l]
    for l in lines[:3] + lines[-3:]:
        print("  ", l)
os.unlink(f.name)
PY
