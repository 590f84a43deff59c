#!/bin/bash
# kernel trace (with timestamps) of one bench step, trimmed to the columns the timeline analysis needs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02g -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02g_prof.log 2>&1; echo "prof rc=$?" >> gpurun_out/r02g_prof.log
find gpurun_out/prof_r02g -name '*.db' -delete
F=$(find gpurun_out/prof_r02g -name '*kernel_trace.csv' | head -1)
head -1 $F
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "dispatches")
with open("gpurun_out/r02g_trace_trim.csv", "w") as f:
    f.write("name,start,end,stream,queue\n")
    for r in rows:
        f.write("%s,%s,%s,%s,%s\n" % (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(",", ";")[:60], r["Start_Timestamp"], r["End_Timestamp"], r.get("Stream_Id", ""), r.get("Queue_Id", "")))
PY
find gpurun_out/prof_r02g -name '*kernel_trace.csv' -delete
ls -la gpurun_out/prof_r02g/*/ | head; tail -2 gpurun_out/r02g_prof.log | cut -c1-200
