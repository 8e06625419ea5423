#!/bin/bash
# Diagnostic PMC passes (LDS / vector-memory / L2 side) over a workload; run ON THE GPU BOX:  bash tools/pmc_diag.sh <tag> <command...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/diag_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/pmc_$name" -o p --output-format csv -- "${CMD[@]}" > "$OUT/pmc_$name.log" 2>&1; }
CMD=("$@")
pass SQ1 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES
pass SQ2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES
pass TA TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
pass TCP TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass TCC TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
pass GRBM GRBM_GUI_ACTIVE GRBM_TA_BUSY
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, c in acc.items():
    print("==", k, "launches", len(n[k]), "dur_us", round(dur[k] / 1e3, 1))
    for name in sorted(c): print(f"   {name:42s} {c[name]:16.0f}")
PY
