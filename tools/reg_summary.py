import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ir=d.get('image_registrator', d)
print('ms_per_run_iteration', ir.get('ms_per_run_iteration'), 'value', ir.get('value'), 'refresh', ir.get('observation_refresh_ms_all_images'))
kg=ir.get('run_phase_profile',{}).get('kernel_groups',{})
for k,v in sorted(kg.items(), key=lambda kv:-kv[1]['ms_per_iteration']):
    print('  %-28s %7.3f ms/it  %5.1f launches  %.4f ms' % (k, v['ms_per_iteration'], v['launches_per_iteration'], v['avg_launch_ms']))
