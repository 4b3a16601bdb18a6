import csv,sys,glob
def load(d):
    f=glob.glob(d+'/*/*kernel_stats.csv')[0]
    out={}
    for r in csv.DictReader(open(f)):
        n=r['Name']
        key=n.split('(')[0][-60:] if 'gemm_h_kernel' not in n else 'gemm<'+n.split('>, ')[-1][:30] if False else n[:140]
        out[n[:150]]=(int(r['Calls']),float(r['AverageNs'])/1e3)
    return out
a,b=load(sys.argv[1]),load(sys.argv[2])
rows=[]
for k in a:
    if k in b and a[k][0]>=8:
        rows.append((a[k][0]*a[k][1], k, a[k], b[k]))
rows.sort(reverse=True)
for tot,k,x,y in rows[:22]:
    short=k.replace('(anonymous namespace)::','').replace('void ','')[:70]
    print(f'{short:70s} n {x[0]:4d}  no-NT {x[1]:8.1f} us   NT {y[1]:8.1f} us   {100*(y[1]/x[1]-1):+5.1f} %')
