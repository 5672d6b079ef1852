#!/bin/bash
# Developer tool: the default bench step under a list of environment settings, back to back in one process each; prints
# frames/s, ms per step and the per-kernel timing slots.  usage: tools/ab_bench.sh "A=1" "A=0 B=2" ...   (on the GPU box)
for v in "$@"; do
  env $v python bench.py --cpu-sample 0 --no-host-leg --no-secondary --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('%-40s %8.0f frames/s %6.3f ms/step | '%(sys.argv[1],d['value'],d['ms_per_step'])+' '.join('%s %.3f'%(n.replace('k_',''),k[n]['ms_per_step']) for n in sorted(k,key=lambda n:-k[n]['ms_per_step'])[:14]))
" "$v"
done
