"""MFMA utilisation of the bulk trailing-update kernel from rocprofv3 --pmc (scratch; run on the GPU box):
    python scratch/pmc_mfma.py
One bench iteration under `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`."""
import glob, os, shutil, sqlite3, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "gpurun_out", "pmc_mfma")
shutil.rmtree(d, ignore_errors=True)
ctrs = sys.argv[1:] or ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["-d", d, "-o", "p", "--", "python3",
       os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "1", "--warmup", "0"]
r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
if r.returncode != 0:
    print(r.stdout[-1500:], r.stderr[-1500:]); sys.exit(1)
db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
tabs = [x[0] for x in c.execute("select name from sqlite_master where type='table'")]
sym = [t for t in tabs if "info_kernel_symbol" in t][0]
pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
info = [t for t in tabs if "info_pmc" in t][0]
disp = [t for t in tabs if "kernel_dispatch" in t][0]
q = (f"select s.kernel_name, i.name, count(distinct k.id), sum(e.value), sum(k.end - k.start) / 1 from {pmc} e "
     f"join {info} i on e.pmc_id = i.id join {disp} k on e.event_id = k.event_id "
     f"join {sym} s on k.kernel_id = s.id where s.kernel_name like '%k_update128ILb0%' group by s.kernel_name, i.name")
vals = {}
for name, ctr, n, tot, dur in c.execute(q):
    vals[ctr] = (n, tot)
    print(ctr, "launches", n, "sum", tot)
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    m, g = vals["SQ_VALU_MFMA_BUSY_CYCLES"][1], vals["GRBM_GUI_ACTIVE"][1]
    print("MFMA busy / GUI active = %.3f (per SIMD if the counter sums 1024 SIMDs: %.4f)" % (m / g, m / g / 1024))
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals:
    print("MFMA busy / SQ busy = %.4f" % (vals["SQ_VALU_MFMA_BUSY_CYCLES"][1] / vals["SQ_BUSY_CYCLES"][1]))
