import json, os, statistics, subprocess, sys
names = sys.argv[2:]; wl = sys.argv[1]
res = {}
for r in range(4):
    for n in names:
        env = dict(os.environ)
        if n != "product":
            env["MBX_HIP_LIBRARY"] = os.path.join(os.getcwd(), "mbelib-neo_amd", "variants", f"libmbx_hip_{n}.so"); env["MBX_HIP_LIBRARY_ALLOW_OLDER"] = "1"
        out = subprocess.run([sys.executable, "bench.py", "--workload", wl, "--steps", "20", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res.setdefault(n, []).append((d["ms_per_step"], d["roofline"]["kernel_ms"]))
for n in names:
    st = [a for a, b in res[n]]; k = [b for a, b in res[n]]
    print(f"{wl} {n:8s} step median {statistics.median(st):.4f} ms  kernel {statistics.median(k):.4f}  front = step - kernel {statistics.median(st) - statistics.median(k):.4f}  {[round(x,4) for x in st]}")
