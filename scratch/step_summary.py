"""Summary of the overlapped steps in a rocprofv3 --kernel-trace csv of bench.py: per step the long kernels on the main
stream, gaps between them, and the side streams' busy time."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_store_host' in r['Kernel_Name']]
main_stream = rows[idx[0]]['Stream_Id']
for w in range(len(idx) - 1):
    seg = rows[idx[w]:idx[w + 1] + 1]
    side = [r for r in seg if r['Stream_Id'] != main_stream]
    if not side:
        continue
    t0 = int(seg[0]['End_Timestamp'])
    print('step', w, 'duration %.1f' % ((int(seg[-1]['End_Timestamp']) - t0) / 1000), 'side kernels', len(side))
    if w != int(sys.argv[2]) if len(sys.argv) > 2 else False:
        continue
    prev = t0
    for r in seg[1:]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if r['Stream_Id'] == main_stream and (e - s > 20000 or 'records' in r['Kernel_Name'] or 'store_host' in r['Kernel_Name']):
            print(f"   {(s - t0) / 1000:9.1f} +{(e - s) / 1000:7.1f} gap {(s - prev) / 1000:7.1f}  {r['Kernel_Name'][:50]}")
            prev = e
    print('   side busy %.1f, last side end %.1f' % (sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in side) / 1000,
                                                    (int(side[-1]['End_Timestamp']) - t0) / 1000))
    last_main = max(int(r['End_Timestamp']) for r in seg if 'k_stats_multi' in r['Kernel_Name'] or 'k_stats_' in r['Kernel_Name'])
    print('   after the last statistics kernel:')
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s >= last_main:
            print(f"      {(s - t0) / 1000:9.1f} +{(e - s) / 1000:6.1f} s{r['Stream_Id']} {r['Kernel_Name'][:60]}")
