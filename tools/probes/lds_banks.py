import numpy as np
N=8192
def brev(i,b=13):
    r=0
    for k in range(b):
        if i>>k&1: r|=1<<(b-1-k)
    return r
BR=np.array([brev(i) for i in range(N)])
def degree(idx, XI):
    # idx: array of 64 element indices accessed by the lanes of one wave; returns max lanes per 8-byte slot class (ideal 4)
    a=np.array([XI(int(i)) for i in idx])
    cls=a%16
    return np.bincount(cls,minlength=16).max()
def report(name, XI):
    out=[]
    # brev store: lanes i=base+l -> XI(brev(i))
    d=max(degree(BR[b:b+64],XI) for b in range(0,N,64*16))
    out.append(("brev_store",d))
    # passes: R=3 at p_lo=0,3,6,9 ; R=1 at 12
    for p_lo,R in ((0,3),(3,3),(6,3),(9,3),(12,1)):
        worst=0
        for g0 in range(0,N>>R,64*4):
            g=np.arange(g0,g0+64)
            low=g&((1<<p_lo)-1); high=g>>p_lo
            base=(high<<(p_lo+R))|low
            for e in range(1<<R):
                worst=max(worst,degree(base+(e<<p_lo),XI))
        out.append((f"pass{p_lo}",worst))
    # natural order
    out.append(("natural",degree(np.arange(64),XI)))
    size=max(XI(i) for i in range(N))+1
    print(name,size,out)
report("i+(i>>3)", lambda i:i+(i>>3))
report("i+(i>>3)+(i>>7)", lambda i:i+(i>>3)+(i>>7))
report("i+(i>>3)+(i>>6)", lambda i:i+(i>>3)+(i>>6))
report("i+(i>>3)+(i>>7)+(i>>10)", lambda i:i+(i>>3)+(i>>7)+(i>>10))
report("i+(i>>3)+(i>>9)", lambda i:i+(i>>3)+(i>>9))
report("i+(i>>4)+(i>>7)", lambda i:i+(i>>4)+(i>>7))
report("i+(i>>4)+(i>>8)", lambda i:i+(i>>4)+(i>>8))
report("i+(i>>5)+(i>>9)", lambda i:i+(i>>5)+(i>>9))
report("i+(i>>3)+(i>>7)+(i>>11)", lambda i:i+(i>>3)+(i>>7)+(i>>11))
print("---- per 16-lane / 32-lane group worst multiplicity (ideal 1 / 2)")
def degree_g(idx, XI, G):
    a=np.array([XI(int(i)) for i in idx])%16
    return max(np.bincount(a[k:k+G],minlength=16).max() for k in range(0,64,G))
def report2(name, XI):
    out=[]
    for G in (16,32):
        d=max(degree_g(BR[b:b+64],XI,G) for b in range(0,N,64*16))
        row=[("brev",d)]
        for p_lo,R in ((0,3),(3,3),(6,3),(9,3),(12,1)):
            worst=0
            for g0 in range(0,N>>R,64*4):
                g=np.arange(g0,g0+64)
                low=g&((1<<p_lo)-1); high=g>>p_lo
                base=(high<<(p_lo+R))|low
                for e in range(1<<R):
                    worst=max(worst,degree_g(base+(e<<p_lo),XI,G))
            row.append((f"p{p_lo}",worst))
        row.append(("nat",degree_g(np.arange(64),XI,G)))
        out.append((G,[int(x[1]) for x in row]))
    print(name,out)
for nm,f in (("i+(i>>3)", lambda i:i+(i>>3)),("i+(i>>5)+(i>>9)", lambda i:i+(i>>5)+(i>>9)),("i+(i>>4)+(i>>8)", lambda i:i+(i>>4)+(i>>8)),("i+(i>>4)+(i>>7)+(i>>10)", lambda i:i+(i>>4)+(i>>7)+(i>>10)),("i+(i>>4)+(i>>8)+(i>>12)", lambda i:i+(i>>4)+(i>>8)+(i>>12))):
    report2(nm,f)
