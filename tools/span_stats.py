import re,sys
tot=[]
for l in sys.stdin:
    m=re.search(r' total=(\d+)',l)
    if m: tot.append(int(m.group(1)))
    if 'kernel' in l and 'diag' in l: print(l.strip())
if tot:
    t=sorted(tot); n=len(t)
    print("streams %d: span M ticks min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f avg %.2f"%(n,t[0]/1e6,t[n//10]/1e6,t[n//2]/1e6,t[(9*n)//10]/1e6,t[-1]/1e6,sum(t)/n/1e6))
