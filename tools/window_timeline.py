import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].split('(')[0].split('::')[-1].replace('void ','')
gi=[i for i,r in enumerate(rows) if nm(r)=='k_gen_samples']
q=rows[gi[-2]:gi[-1]]
t0=int(q[0]['Start_Timestamp'])
ev=[(nm(r)[:12],(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in q]
# print a window of events around steady state
w=[e for e in ev if 5000<=e[1]<=5400]
for e in w: print('%-13s @%8.1f +%6.1f'%e)
print('span', ev[-1][1]+ev[-1][2])
