import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d = json.loads(line)
        print('fps %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']))
        print({k: round(v['ms'], 3) for k, v in d['stages_per_batch'].items()})
        r = d['roofline']; print('roofline', r['kernel'], 'GB/s %.0f frac %.3f (moved bytes: %.3f)' % (r['achieved'], r['frac'], r['moved']['frac']), 'traffic', r['traffic'])
        print('pyramid', {k: round(v, 3) for k, v in d['pyramid'].items() if not isinstance(v, dict)})
        for k, v in d.get('kernels', {}).items(): print('  ', k, v['bound'], 'frac %.3f' % v['frac'], 'traffic', v.get('traffic'))
        if 'end_to_end' in d: print('end_to_end fps %.1f' % d['end_to_end']['frames_per_s'])
        if 'knn' in d: print('knn ms %.2f frac %.3f' % (d['knn']['ms'], d['knn']['roofline']['frac']), 'cpu pairs/s %.3g' % d['knn'].get('cpu_baseline', {}).get('value', 0))
        if 'cpu_baseline' in d: print('cpu', d['cpu_baseline']['value'])
