import re,collections,json,sys
for tag in sys.argv[1:]:
    d=json.loads(open(f'/root/repo/gpurun_out/b_{tag}.json').read().strip().splitlines()[-1])['end_to_end']
    print(tag, round(d['batch']['seconds'],3), d['batch']['stages'], d['batch']['expand_us'], d['batch']['execute_us'], int(d['batch_no_wildcards']['queries_per_s']), round(d['single_query']['median_latency_ms'],4))
    lines=[l for l in open(f'/root/repo/gpurun_out/trace_{tag}.err') if l.startswith('[tetrex]')]
    runs=[];cur=None
    for l in lines:
        if 'busy' in l:
            m=re.match(r'\[tetrex\] busy sum\s+([\d.]+) ms max\s+([\d.]+) ms ops (\d+)',l)
            cur.append(('busy_sum',float(m.group(1))));cur.append(('busy_max',float(m.group(2))));continue
        m=re.match(r'\[tetrex\] (\w+)\s+([\d.]+) ms',l)
        if m.group(1)=='graphs' and (cur is None or len(cur)>1):
            cur=[];runs.append(cur)
        cur.append((m.group(1),float(m.group(2))))
    big=max(runs,key=len)
    tot=collections.Counter()
    for k,v in big: tot[k]+=v
    print('   ',{k:round(v,1) for k,v in tot.items()})
