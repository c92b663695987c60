"""Host model of dmin3 (graph_kernels.hip) in its spread form - row i at bit 2i, pv with every odd bit set so that the
addition carries through - against the edit-distance recurrence: min(ed(a, b), ed(a, b[:-1]), ed(a[:-1], b)) on 20,000 pairs
(equal, random, and a few edits apart).  python3 tools/myers_spread_check.py"""
import random
M=0xFFFFFFFF; EVEN=0x55555555
def sbfe(x,o): return M if (x>>o)&1 else 0
def new(a,b):
    P0=a&EVEN; P1=(a>>1)&EVEN
    pv=M; mv=0; score=16; s15=0
    for j in range(16):
        m0=sbfe(b,2*j); m1=sbfe(b,2*j+1)
        t1=(~(m0^P0))&EVEN&M
        eq=t1&(~(m1^P1))&M
        xv=eq|mv
        s=((eq&pv)+pv)&M
        xh=((s^pv)|eq)&M
        ph=(mv|(~(xh|pv)))&M
        mh=pv&xh
        score+=(ph>>30)&1; score-=(mh>>30)&1
        ph=((ph<<2)|1)&M; mh=(mh<<2)&M
        pv=(mh|(~(xv|ph)))&M
        mv=ph&xv
        if j==14: s15=score
    d1516=score-((pv>>30)&1)+((mv>>30)&1)
    return min(score,s15,d1516)
def ed(x,y):
    D=list(range(len(y)+1))
    for i in range(1,len(x)+1):
        p=D[0]; D[0]=i
        for j in range(1,len(y)+1):
            c=D[j]; D[j]=min(D[j]+1,D[j-1]+1,p+(x[i-1]!=y[j-1])); p=c
    return D[-1]
def seq(r): return [(r>>(2*i))&3 for i in range(16)]
random.seed(1)
for t in range(20000):
    a=random.getrandbits(32)
    if t%3==0: b=a
    else: b=random.getrandbits(32)
    if t%3==1:
        # few edits
        s=seq(a)
        for _ in range(random.randint(0,3)):
            k=random.random()
            if k<0.4: s[random.randrange(16)]=random.randrange(4)
            elif k<0.7: del s[random.randrange(len(s))]; s.append(random.randrange(4))
            else: s.insert(random.randrange(16),random.randrange(4)); s=s[:16]
        b=sum(c<<(2*i) for i,c in enumerate(s))
    x,y=seq(a),seq(b)
    want=min(ed(x,y),ed(x,y[:-1]),ed(x[:-1],y))
    got=new(a,b)
    assert got==want,(hex(a),hex(b),got,want)
print("ok")
