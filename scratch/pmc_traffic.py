"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one
bench.py iteration; writes gpurun_out/r03_pmc_traffic_cfg<N>.json (copy it to profiles/).

Run on the GPU box (one pass per counter: the TCC block cannot hold both, MI355X_MICROARCH.md):
    python scratch/pmc_traffic.py <config>
The passes are started by this script as child processes `rocprofv3 --kernel-trace --pmc X -- python3
bench.py ...` (this script itself never touches the GPU).  Corrections as the guide's HBM section
prescribes for gfx950: counters are in KB; FETCH_SIZE is doubled (128-B requests tallied at 64 B
for 16 B/lane streaming reads)."""
import json, os, sqlite3, subprocess, sys, glob, shutil

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def demangle(name):
    name = name[:-3] if name.endswith(".kd") else name
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            d = subprocess.run([tool, name], capture_output=True, text=True).stdout.strip()
            if d:
                d = d.split("(")[0]          # drop the parameter list
                return d[5:] if d.startswith("void ") else d   # template arguments may contain spaces
        except OSError:
            pass
    return name


root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join("/tmp", "pmc_%s_cfg%d" % (ctr, cfg))  # raw pass output: large, not worth copying back
    shutil.rmtree(d, ignore_errors=True)
    # PMC_KERNEL_REGEX: instrument only the matching kernels (configs[4]: a full pass over the 2344-tile
    # factorisation's ~10^5 launches does not finish in 18 minutes; the HBM-bound kernels are what is wanted)
    flt = ["--kernel-include-regex", os.environ["PMC_KERNEL_REGEX"]] if os.environ.get("PMC_KERNEL_REGEX") else []
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr] + flt + ["-d", d, "-o", "p", "--",
           "python3", os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-api", "--steps", "1", "--warmup", "0",
           "--config", str(cfg)]
    env = dict(os.environ, TMPDIR="/tmp")
    print("pass", ctr, flush=True)
    # (a pass of the large configs runs for minutes without output: print a heartbeat, the GPU box kills silent runs)
    import time
    with open(os.path.join("/tmp", "pmc_%s_cfg%d.out" % (ctr, cfg)), "w") as fo:
        pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=fo, stderr=subprocess.STDOUT)
        t0 = time.time()
        while pr.poll() is None:
            time.sleep(30)
            print("  ... %s pass running, %d s" % (ctr, time.time() - t0), flush=True)
    if pr.returncode != 0:
        print(open(fo.name).read()[-3000:]); sys.exit(1)
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [x[0] for x in c.execute("select name from sqlite_master where type='table'")]
    sym = [t for t in tabs if "info_kernel_symbol" in t][0]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    info = [t for t in tabs if "info_pmc" in t][0]
    disp = [t for t in tabs if "kernel_dispatch" in t][0]
    q = (f"select s.kernel_name, count(distinct k.id), sum(e.value) from {pmc} e "
         f"join {info} i on e.pmc_id = i.id join {disp} k on e.event_id = k.event_id "
         f"join {sym} s on k.kernel_id = s.id where i.name = '{ctr}' group by s.kernel_name")
    for name, n, tot in c.execute(q):
        short = demangle(name)
        o = out.setdefault(short, {"launches": n})
        o["%s_KB_per_launch" % ctr] = tot / n
    c.close()
    shutil.rmtree(d, ignore_errors=True)
for k, o in out.items():
    f = o.get("FETCH_SIZE_KB_per_launch", 0.0); w = o.get("WRITE_SIZE_KB_per_launch", 0.0)
    o["traffic_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
res = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes, scratch/pmc_traffic.py) -- "
                  "python3 bench.py --no-cpu-baseline --no-api --steps 1 --warmup 0 --config %d" % cfg,
       "corrections": "KB -> bytes; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)",
       "kernels": dict(sorted(out.items()))}
path = os.path.join(root, "gpurun_out", "r03_pmc_traffic_cfg%d.json" % cfg)
json.dump(res, open(path, "w"), indent=1)
print("wrote", path)
for k in sorted(out, key=lambda k: -out[k]["traffic_bytes_per_launch_corrected"] * out[k]["launches"])[:8]:
    print("%-60s launches %5d  %.3f GB/launch" % (k[:60], out[k]["launches"], out[k]["traffic_bytes_per_launch_corrected"] / 1e9))
