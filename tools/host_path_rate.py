#!/usr/bin/env python3
"""tools/host_path_rate.py [streams] -- end-to-end host-memory rates of the per-frame shim, its queue mode and the
session API (mbelib-neo_amd/host_bench, a plain C program) on clean all-voiced IMBE frames; prints its JSON line."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(streams=65536, device=0, env=None):
    """env: extra environment of the child (e.g. MBE_NEO_FRAME_SERVER=1 with MBX_HOST_BENCH_SYNC_ONLY=1: only the synchronous call)"""
    import mbelib_neo_amd as m
    from mbelib_neo_amd import framegen

    frames = framegen.imbe_clean_voiced_frames(streams, framegen.rng_for(0xBE0000))
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(frames.tobytes())
        path = f.name
    try:
        exe = os.path.join(ROOT, "mbelib-neo_amd", "host_bench")
        tables = os.path.join(ROOT, "mbelib-neo_amd", "data", "mbx_tables.bin")
        out = subprocess.run([exe, tables, path, str(device)], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, **env) if env else None)
        if out.returncode != 0:
            raise RuntimeError(f"host_bench failed ({out.returncode}): {out.stderr[-500:]}")
        return out.stdout.strip().splitlines()[-1]
    finally:
        os.unlink(path)


if __name__ == "__main__":
    print(run(int(sys.argv[1]) if len(sys.argv) > 1 else 65536))
