import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d.get('forms') or {}
print("value %.1f M  step_many %.3f  order off %.3f | %s | sweep %s" % (d['value']/1e6, f.get('step_many',{}).get('avg_launch_us',0), f.get('dispatch_order_off',{}).get('avg_launch_us',0),
      d.get('legs_us_per_step'), {k:(v['us'], v.get('plan')) for k,v in d['roofline'].get('frac_out_of_cache',{}).items()}))
