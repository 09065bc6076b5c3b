set -u
OUT=$PWD/gpurun_out/prof_c5u
mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace -d "$OUT/trace" --output-format csv -- python3 tools/exp/c5_uniform.py > "$OUT/trace.log" 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/prof_c5u/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"])
out = open("gpurun_out/prof_c5u/last_frame.txt", "w")
t_prev = None
for r in rows[idx:]:
    m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"]); name = m.group(1) if m else r["Kernel_Name"][:30]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - t_prev) / 1e3 if t_prev else 0.0
    out.write(f"{name:18s} dur {(e - s) / 1e3:9.1f} us  gap {gap:8.1f} us\n")
    t_prev = e
out.close()
PY
