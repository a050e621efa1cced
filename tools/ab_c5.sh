#!/bin/bash
# Same-box A/B of two library builds on BASELINE config 5 (B=64, L=512): bash tools/ab_c5.sh <lib A> <lib B> [rounds]
cd $GRAFT_REPO_ROOT
A=${1:-tools/libphonlp_heads_prev.so}; B=${2:-phonlp_amd/lib/libphonlp_heads.so}; R=${3:-2}
run() {
  PHONLP_LIB=$1 PHONLP_BENCH_GRAPH=0 timeout -k 10 300 python tools/ab_bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "FAILED: $1"; tail -3 gpurun_out/ab_tmp.err; return 1; }
  python3 -c "
import json,sys
r=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
c=r['other_configs']['config5_B64_L512']
print('%-40s config5 %.4f ms/step mst %.4f failures %d' % (sys.argv[1], c['ms_per_step'], c['kernels_ms']['mst'], c['mst_failures']))
" "$1"
}
for i in $(seq $R); do run $A && run $B || exit 1; done
