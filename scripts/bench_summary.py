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
        for name, c in d.get('configs', {}).items():
            if 'frames_per_s' in c:
                print('config', name, 'fps %.1f' % c['frames_per_s'], 'pyramid frac %.3f moved %.3f' % (c['pyramid']['frac_of_hbm_peak'], c['pyramid']['moved_frac_of_hbm_peak']),
                      'roofline', c['roofline']['kernel'], '%.3f' % c['roofline']['frac'], 'kp %.0f' % c['keypoints_per_frame_mean'])
            else:
                print('config', name, 'ms %.3f (min %.3f)' % (c['host_to_host_ms_median'], c['host_to_host_ms_min']), c['stage_ms'], 'kp', c['keypoints'],
                      'cpu ms %.0f' % c.get('cpu_baseline', {}).get('value', 0))
        if 'c_group' in d: print('c_group', json.dumps(d['c_group'])[:600])
