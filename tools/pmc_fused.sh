#!/bin/bash
# PMC passes over the fused label stage A alone (tools/fused_probe.py): where the waves' cycles go.
# usage (on the GPU box, from the repo root): bash tools/pmc_fused.sh <outdir>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_fused}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY"
P2="SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE"
for p in 1 2 3; do
  v=P$p
  timeout -s KILL 120 rocprofv3 --pmc ${!v} -d $OUT/p$p -o p$p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/fused_probe.py 3 > $OUT/p$p.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2","p3"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if "linear_f16x3_wide16_heads_kernel" in row["Kernel_Name"] or "linear_f16x3_wide_kernel<1>" in row["Kernel_Name"] or "linear_f16x3_dma_kernel<1, false" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(p, k, "n=%d" % len(v), "mean=%.5g" % (sum(v)/len(v)))
PY
