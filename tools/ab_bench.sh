#!/bin/bash
# A/B of bench.py under environment switches: bash tools/ab_bench.sh "VAR=1" "OTHER=0 VAR2=x" ...  (first run: no switch)
cd $GRAFT_REPO_ROOT
run() {
  env $1 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "FAILED: $1"; tail -3 gpurun_out/ab_tmp.err; return 1; }
  python3 -c "
import json,sys
r=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
k=r['kernels_ms']
print('%-40s %.3f ms/step  stageA %.3f  frac %.3f | %s' % (sys.argv[1] or '(default)', r['ms_per_step'], r['roofline']['mean_ms'], r['roofline']['frac'], ' '.join('%s %.3f' % (n.replace('linear:',''), v) for n, v in k.items() if n != 'linear:label_stage_a')))
" "$1"
}
run "" && for v in "$@"; do run "$v" || exit 1; done
