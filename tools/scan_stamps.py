import numpy as np, sys
a=np.fromfile(sys.argv[1],dtype=np.uint64).reshape(-1,2).astype(np.float64)
a=a[a[:,1]>0]
t0=a[:,0].min(); s=(a[:,0]-t0)/100.0; e=(a[:,1]-t0)/100.0   # microseconds (100 MHz)
print("waves",len(a),"start us: max",s.max().round(1),"| end us: min",e.min().round(1),"p10",np.percentile(e,10).round(1),"p50",np.percentile(e,50).round(1),"p90",np.percentile(e,90).round(1),"p99",np.percentile(e,99).round(1),"max",e.max().round(1))
d=e-s; print("duration us: min",d.min().round(1),"p50",np.median(d).round(1),"max",d.max().round(1))
# by XCD (block % 8)
blk=np.arange(len(a))//4
for x in range(8):
    m=(blk%8)==x; print(" xcd-group",x,"end p50",np.median(e[m]).round(1),"max",e[m].max().round(1))
