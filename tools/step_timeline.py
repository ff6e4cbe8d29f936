import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].split('(')[0].split('::')[-1].replace('void ','')
gi=[i for i,r in enumerate(rows) if nm(r)=='k_gen_samples']
q=rows[gi[-3]:gi[-2]]
t0=int(q[0]['Start_Timestamp'])
steps=[];cur=None
for r in q:
    n=nm(r)
    if n.startswith('k_near'):
        cur=[];steps.append(cur)
    if cur is not None: cur.append((n,(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
for si in (2,10,30,60,61,62,63,100,101,107):
    if si>=len(steps): continue
    s=steps[si]; b=s[0][1]
    print(si,' '.join('%s@%.0f+%.0f'%(n[2:9],t-b,d) for n,t,d in s))
print('query span us',(int(q[-1]['End_Timestamp'])-t0)/1e3, 'steps',len(steps))
per=[steps[i+1][0][1]-steps[i][0][1] for i in range(len(steps)-1)]
print('period', ['%.0f'%p for p in per[::6]])
