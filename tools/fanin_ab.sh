#!/bin/bash
# fanin_ab.sh -- GPU box: host_bench's thread sweep (queue mode / sessions from 1, 4, 16 and 2, 8, 32 host threads) with the
# zero-copy small flush and the pump thread off / on, twice each, interleaved.  (Development aid.)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from mbelib_neo_amd import framegen
frames = framegen.imbe_clean_voiced_frames(65536, framegen.rng_for(0xBE0000))
f = tempfile.NamedTemporaryFile(suffix=".bin", delete=False); f.write(frames.tobytes()); f.close()
exe = "mbelib-neo_amd/host_bench"; tables = "mbelib-neo_amd/data/mbx_tables.bin"
for rep in range(2):
    for threads in ("1,4,16", "2,8,32"):
        for zc, pm in (("0", "0"), ("1", "0"), ("1", "1")):
            e = dict(os.environ, HB_THREADS=threads, MBE_NEO_ZERO_COPY_FLUSH=zc, MBE_NEO_PUMP=pm)
            out = subprocess.run([exe, tables, f.name, "0"], capture_output=True, text=True, env=e)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                print("zero_copy", zc, "pump", pm, d["threads"], "queue", [round(x / 1e6, 1) for x in d["queue_resident_frames_per_s_by_threads"]],
                      "session", [round(x / 1e6, 1) for x in d["session_pinned_frames_per_s_by_threads"]], "1-thread queue", round(d["queue_resident_frames_per_s"] / 1e6, 1), flush=True)
            except Exception as ex:
                print("FAILED", out.stderr[-300:])
PY
