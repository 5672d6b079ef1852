"""Pretty-print the JSON line of a bench.py log."""
import json
import sys

line = [l for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
d = json.loads(line)
print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus', 'steps')}, 'found b/d', d['config']['found_bright'],
      d['config']['found_dim'], 'gen_s', d['config']['gen_s'])
print('roofline', d['roofline'])
print('cpu', d.get('cpu_baseline'))
tot = 0
for k, v in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['ms_per_step']):
    print(f"{k:22s} {v['ms_per_step']:8.3f} ms/step  launches {v['launches_per_step']:3d} frames {v['frames_per_step']}")
    tot += v['ms_per_step']
print('sum of kernels %.2f ms/step' % tot)
