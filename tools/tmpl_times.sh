#!/bin/bash
# Kernel times of one direct solve of a template group at the C3 size (tools/bench_template.py under rocprofv3):
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/tmpl_times.sh'   -> gpurun_out/tmpl_traceX/, gpurun_out/tmplX.log
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/tmpl_traceX && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tmpl_traceX -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_template.py > $GRAFT_REPO_ROOT/gpurun_out/tmplX.log 2>&1; cd $GRAFT_REPO_ROOT; grep "^direct" gpurun_out/tmplX.log | tail -1; python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/tmpl_traceX/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "schur" in r["Name"] or "k_amp_reg" in r["Name"]:
        print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3), "ms avg")
PY
