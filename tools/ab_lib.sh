#!/bin/bash
# Same-box A/B of library builds: bash tools/ab_lib.sh <lib A> <lib B> [rounds]   (bench.py headline, without the other configs)
cd $GRAFT_REPO_ROOT
A=${1:-tools/libphonlp_heads_prev.so}; B=${2:-phonlp_amd/lib/libphonlp_heads.so}; R=${3:-2}
run() {
  PHONLP_LIB=$1 PHONLP_BENCH_OTHERS=0 timeout -k 10 200 python tools/ab_bench.py --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "FAILED: $1"; tail -3 gpurun_out/ab_tmp.err; return 1; }
  python3 -c "
import json,sys
r=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
k=r['kernels_ms']
print('%-40s %.4f ms/step streaming %.4f stageA %.4f | mst %.4f gather %.4f reduce %.4f split %.4f pos_tail %.4f viterbi %.4f' % (sys.argv[1], r['ms_per_step'], r['streaming']['ms_per_step'], r['roofline']['mean_ms'], k['mst'], k['gather_heads_t'], k['label_reduce'], k['split_rows'], k['pos_tail'], k['viterbi']))
" "$1"
}
for i in $(seq $R); do run $A && run $B || exit 1; done
