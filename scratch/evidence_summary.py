"""Prints the numbers of the committed round-3 evidence (profiles/r03_*) the documents quote."""
import json, csv, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
def last(p): return json.loads(open(os.path.join(R, p)).read().strip().splitlines()[-1])
for n in ['llama8b','llama70b_shard','c2']:
    d=last(f'r03_{n}_bench_under_trace.json'); r=d['roofline']
    print('trace',n, d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r['launches_timed'], (r.get('launch_alone') or {}).get('avg_launch_us'))
    tot=0;cnt=0
    for row in csv.reader(open(os.path.join(R, f'r03_{n}_kernel_stats.csv'))):
        if row and ('k_stats_multi' in row[0] or ('k_quant_team' in row[0] and row[0].rstrip().endswith('2>(QuantBatch)'))):
            print('   ', row[0][:50], row[1], round(float(row[3])/1000,1)); tot+=float(row[2]); cnt+=int(row[1])
    print('    mean', round(tot/cnt/1000,1), cnt)
d=last('r03_bench_default.json'); r=d['roofline']
print('default', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r['launches_timed'], r['launch_alone']['avg_launch_us'], r['launch_alone']['frac'], r.get('issue',{}).get('valu_busy'), r['traffic'], r['algorithmic_bytes_per_launch'])
print(' olpp', {k:v for k,v in r['one_launch_per_pass'].items() if k!='note'})
for k,v in d['extra'].items():
    rr=v['roofline']; print(' extra',k,v['value'],v['ms_per_step'],rr['frac'],rr['avg_launch_us'],(rr.get('launch_alone') or {}).get('frac'), (rr.get('launch_alone') or {}).get('avg_launch_us'), rr.get('issue',{}).get('valu_busy'), rr.get('traffic'))
print(' cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['value_1_thread'])
for f in ['r03_bench_llama8b_numpy_order','r03_bench_llama70b_one_gpu','r03_bench_gloo2_rehearsal']:
    d=last(f'{f}.json'); r=d['roofline']; print(f, d['value'], d['ms_per_step'], d['n_gpus'], r['frac'], r['avg_launch_us'], (r.get('launch_alone') or {}).get('frac'))
t=json.load(open(os.path.join(R,'traffic.json'))); print({k:v for k,v in t.items() if not k.endswith('detail')})
i=json.load(open(os.path.join(R,'issue.json'))); print({k:v['valu_busy'] for k,v in i.items()})
