import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f M  ordered %.3f hinted %.3f | %s | sweep %s" % (d['value']/1e6, d['ordered']['avg_launch_us'], d['ordered']['per_step_calls_hinted_us'], d['legs_us_per_step'], {k:v['us'] for k,v in d['roofline']['frac_out_of_cache'].items()}))
