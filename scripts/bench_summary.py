import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d = json.loads(line)
        print('fps %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']))
        print({k: round(v['ms'], 3) for k, v in d['stages_per_batch'].items()})
        r = d['roofline']; print('roofline', r['kernel'], 'GB/s %.0f frac %.3f' % (r['achieved'], r['frac']))
        print('pyramid', {k: round(v, 3) for k, v in d['pyramid'].items()})
        if 'knn' in d: print('knn ms %.2f frac %.3f' % (d['knn']['ms'], d['knn']['roofline']['frac']))
        if 'cpu_baseline' in d: print('cpu', d['cpu_baseline']['value'])
