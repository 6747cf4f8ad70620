#!/bin/bash
# timeline (start offset, duration, gap to the previous kernel's end) of the kernels of the last launch of the driver-shaped bench
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_tl}"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d "$O/kt" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps ${2:-20} --warmup 3 --no-cpu-baseline --no-reupload-leg $BENCH_ARGS > "$O/b.json" 2> "$O/b.err"
python3 - "$O" <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/kt/**/k_kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last launch: from the last k_decompress on
import os
dec=[i for i,r in enumerate(rows) if "k_decompress" in r["Kernel_Name"]]
idx=dec[-int(os.environ.get("LAST_LAUNCHES","1"))]
t0=int(rows[idx]["Start_Timestamp"]); prev_end=t0
for r in rows[idx:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-34s start=%8.1f dur=%8.1f gap=%7.1f us  grid=%s wg=%s" % (r["Kernel_Name"].split("(")[0].replace("void ","").replace("h2v::","")[:34], (s-t0)/1e3,(e-s)/1e3,(s-prev_end)/1e3, r.get("Grid_Size_X","?"), r.get("Workgroup_Size_X","?")))
    prev_end=max(prev_end,e)
PY
rm -rf "$O/kt"
