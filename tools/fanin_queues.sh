cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from mbelib_neo_amd import framegen
frames = framegen.imbe_clean_voiced_frames(65536, framegen.rng_for(0xBE0000))
f = tempfile.NamedTemporaryFile(suffix=".bin", delete=False); f.write(frames.tobytes()); f.close()
exe = "mbelib-neo_amd/host_bench"; tables = "mbelib-neo_amd/data/mbx_tables.bin"
for env in ({}, {"GPU_MAX_HW_QUEUES": "8"}, {"GPU_MAX_HW_QUEUES": "16"}, {"HB_THREADS": "2,8,32"}):
    e = dict(os.environ, HB_THREADS="1,4,16"); e.update(env)
    out = subprocess.run([exe, tables, f.name, "0"], capture_output=True, text=True, env=e)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(env, d["threads"], "queue", d["queue_resident_frames_per_s_by_threads"], "session", d["session_pinned_frames_per_s_by_threads"], flush=True)
    except Exception as ex:
        print(env, "FAILED", out.stderr[-300:])
PY
