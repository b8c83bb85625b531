#!/bin/bash
# Run on the GPU box (via gpurun): SQ issue counters of the default bench command, one counter set
# per rocprofv3 run (gpurun refuses --pmc combined with other trace domains).
#   tools/pmc_valu.sh r01
TAG=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O; cd $R
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/set$i -- python3 bench.py --steps 3 --warmup 1 --quick --repeats 1 --submit stream > $O/set$i.log 2>&1 || echo "set $i failed: $set"
done
python3 - "$O" "$TAG" <<'PY'
import csv, glob, sys, collections
O, TAG = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/set*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void smx::", "")
        if "smx" not in row["Kernel_Name"]:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O + "/summary.txt", "w") as out:
    for k in sorted(acc):
        out.write(k + "\n")
        for c in sorted(acc[k]):
            v = acc[k][c]
            out.write(f"    {c:26s} mean per launch {sum(v)/len(v):16.0f}   ({len(v)} launches)\n")
print(open(O + "/summary.txt").read())
# issue_busy of the dominant kernel: wave-level VALU instructions x the measured mean issue cost of this
# kernel's instruction mix (tools/ubench/issue_rate.hip: 4.93 SIMD-clocks) / SIMD-clocks of the launch
import json
dom = max((k for k in acc if "SQ_INSTS_VALU" in acc[k] and "GRBM_GUI_ACTIVE" in acc[k]), key=lambda k: sum(acc[k]["SQ_INSTS_VALU"]))
iv = sum(acc[dom]["SQ_INSTS_VALU"]) / len(acc[dom]["SQ_INSTS_VALU"])
gui = sum(acc[dom]["GRBM_GUI_ACTIVE"]) / len(acc[dom]["GRBM_GUI_ACTIVE"]) / 8.0      # reported as the sum over 8 XCDs
rec = {"kernel": dom, "insts_valu_per_launch": iv, "gui_active_cycles_per_xcd": gui, "clocks_per_valu_instruction": 4.93,
       "simds": 1024, "issue_busy": iv * 4.93 / (1024 * gui),
       "lds_idx_active_per_cu": sum(acc[dom].get("SQ_LDS_IDX_ACTIVE", [0])) / max(1, len(acc[dom].get("SQ_LDS_IDX_ACTIVE", [0]))) / 256.0,
       "source": "rocprofv3 --pmc passes of `python3 bench.py --steps 3 --warmup 1 --quick --repeats 1 --submit stream` (tools/pmc_valu.sh)"}
json.dump(rec, open(O + "/" + TAG + "_valu.json", "w"), indent=1)
print(json.dumps(rec))
PY
