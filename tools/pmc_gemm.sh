#!/bin/bash
# PMC passes over the stage-A shaped GEMM (M=32768, N=13464, K=408) of the dense-layer kernel.
# usage (on the GPU box, from the repo root): bash tools/pmc_gemm.sh <outdir>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_gemm}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY"
P2="SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY"
timeout -s KILL 120 rocprofv3 --pmc $P1 -d $OUT/p1 -o p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py 32768 13464 408 3
timeout -s KILL 120 rocprofv3 --pmc $P2 -d $OUT/p2 -o p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py 32768 13464 408 3
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if "linear" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(p, k, "n=%d" % len(v), "mean=%.4g" % (sum(v)/len(v)))
PY
