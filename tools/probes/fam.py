import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['roofline']['frac'])
for k,v in d['kernels'].items():
    if True: print(' ',k, v['calls_per_step'], v['avg_us'], v['ms_per_step'])
