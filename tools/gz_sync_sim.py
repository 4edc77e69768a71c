#!/usr/bin/env python3
"""How fast does a DEFLATE decoder started at a wrong bit offset fall into step with the true token chain?  (CPU, pure Python: a
small inflate that keeps the token boundaries of zlib-made FASTQ streams, then decoders started W bits before every R-bit boundary.)
The question behind the lane-sequential token decoder of VERDICT r5 item 5; the answer (profiles/ab/r06_gz_sync.txt) is why it was not built."""
import sys, zlib, numpy as np
sys.path.insert(0, "/root/repo")
from varkoder_amd import synth

LBASE=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT=[0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DEXT=[0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]

class Bits:
    def __init__(s, data):
        s.b=np.unpackbits(np.frombuffer(data,dtype=np.uint8),bitorder='little').astype(np.int64); s.n=len(s.b)
        s.b=np.concatenate([s.b,np.zeros(256,dtype=np.int64)])
        s.l=s.b.tolist()
    def get(s,p,n):
        v=0
        for i in range(n): v|=s.l[p+i]<<i
        return v

def build(lens):
    # canonical: returns dict (len, code_msb) -> sym ; decode bit by bit
    maxl=max(lens) if lens else 0
    cnt=[0]*(maxl+2)
    for l in lens:
        if l: cnt[l]+=1
    code=0; nxt=[0]*(maxl+2)
    for l in range(1,maxl+1):
        code=(code+cnt[l-1])<<1; nxt[l]=code
    tab={}
    for s,l in enumerate(lens):
        if l:
            tab[(l,nxt[l])]=s; nxt[l]+=1
    return tab,maxl

def dec(b,p,tab,maxl):
    code=0
    for l in range(1,maxl+1):
        code=(code<<1)|b.get(p+l-1,1)
        if (l,code) in tab: return tab[(l,code)], l
    return None, 0

def token(b,p,lt,lm,dt,dm):
    """returns (bits used, kind) kind: 0 lit/len ok, 1 eob, 2 bad"""
    s,l=dec(b,p,lt,lm)
    if s is None: return 0,2
    if s<256: return l,0
    if s==256: return l,1
    if s>285: return 0,2
    u=l+LEXT[s-257]
    d,l2=dec(b,p+u,dt,dm)
    if d is None or d>29: return 0,2
    return u+l2+DEXT[d],0

def blocks(data):
    b=Bits(data); p=0
    # gzip header (no flags assumed)
    p=80
    out=[]
    while True:
        last=b.get(p,1); typ=b.get(p+1,2); p+=3
        if typ==0:
            p=(p+7)&~7; ln=b.get(p,16); p+=32+8*ln
        else:
            if typ==2:
                hlit=b.get(p,5)+257; hdist=b.get(p+5,5)+1; hclen=b.get(p+10,4)+4; p+=14
                order=[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]
                cl=[0]*19
                for i in range(hclen): cl[order[i]]=b.get(p,3); p+=3
                ct,cm=build(cl)
                lens=[]
                while len(lens)<hlit+hdist:
                    s,l=dec(b,p,ct,cm); p+=l
                    if s<16: lens.append(s)
                    elif s==16: r=3+b.get(p,2); p+=2; lens+= [lens[-1]]*r
                    elif s==17: r=3+b.get(p,3); p+=3; lens+=[0]*r
                    else: r=11+b.get(p,7); p+=7; lens+=[0]*r
                lt,lm=build(lens[:hlit]); dt,dm=build(lens[hlit:])
            else:
                ll=[8]*144+[9]*112+[7]*24+[8]*8; lt,lm=build(ll); dt,dm=build([5]*30)
            starts=[]
            while True:
                starts.append(p)
                u,k=token(b,p,lt,lm,dt,dm)
                assert k!=2
                p+=u
                if k==1: break
            out.append((starts,lt,lm,dt,dm,b))
        if last: break
    return out

def simulate(level, nreads=3000, R=64, W=64, nblocks=3):
    fq=synth.sample_fastq(3,nreads,150,dist=1).tobytes()
    co=zlib.compressobj(level,zlib.DEFLATED,31); data=co.compress(fq)+co.flush()
    bl=[x for x in blocks(data) if len(x[0])>1000][:nblocks]
    hit=tot=0; ntok=0; nbits=0; runs=[]
    for starts,lt,lm,dt,dm,b in bl:
        true=set(starts); first=starts[0]; last=starts[-1]
        ntok+=len(starts); nbits+=last-first
        # lanes at R*i boundaries: start decode at boundary - W, see whether synced at first token start >= boundary
        run=0
        for bd in range(first+R, last-200, R):
            p=bd-W
            ok=None
            steps=0
            while p<bd+R and steps<200:
                u,k=token(b,p,lt,lm,dt,dm); steps+=1
                if k!=0: break
                if p>=bd:
                    ok = p in true; break
                p+=u
            if ok is None:
                # ran past or stopped: check position
                ok = (p in true) if p>=bd else False
            tot+=1; hit+= 1 if ok else 0
            if ok: run+=1
            else: runs.append(run); run=0
    print("level",level,"R",R,"W",W,"bits/token %.2f"%(nbits/ntok),"sync prob %.4f"%(hit/tot), "mean valid run %.1f"%(np.mean(runs) if runs else -1), "n",tot)

for lvl in (1,6):
    for (R,W) in ((64,32),(64,64),(64,128),(128,64)):
        simulate(lvl,R=R,W=W)
