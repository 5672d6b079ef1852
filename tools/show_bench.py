"""Pretty-print a bench.py JSON line: python tools/show_bench.py gpurun_out/b.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print({k: v for k, v in d.items() if k not in ('kernels', 'stages', 'config', 'roofline', 'cpu_baseline')})
print(d['config'])
print({k: v for k, v in d['roofline'].items() if k != 'note'})
for k, v in d.get('stages', {}).items():
    print("  stage %-20s %s" % (k, v))
for k, v in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['ms_per_step']):
    print("  %-22s %s" % (k, v))
print(d.get('cpu_baseline'))
