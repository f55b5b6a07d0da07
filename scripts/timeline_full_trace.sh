cd /tmp && export TMPDIR=/tmp && rm -rf /root/repo/gpurun_out/tl
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/tl -- python3 /root/repo/scripts/run_workload.py config3 --mode full --reps 3 "$@" > /root/repo/gpurun_out/tl.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/root/repo/gpurun_out/tl/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "ort::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-45]["Start_Timestamp"])
for r in rows[-45:]:
    print("%-28s q=%s start %8.1f us  end %8.1f us  dur %7.1f" % (r["Kernel_Name"].split("(")[0][:28].replace("void ort::",""), r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
