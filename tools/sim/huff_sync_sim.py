"""CPU model of k_jpeg_huff's self-synchronisation (not part of the library): segments decoded from guessed states,
rounds of re-decodes until every segment's entry equals its predecessor's exit.  Reproduces the kernel's per-round
counts (tools/jpeg_rounds.py) exactly, so decoding policies can be tried on the CPU first.
    python3 tools/sim/huff_sync_sim.py"""
import glob, os, sys, struct, random
import numpy as np

def parse(data):
    i=2; dht={}; sos=None; comps=None
    while i < len(data):
        assert data[i]==0xFF
        m=data[i+1]; i+=2
        if m==0xD8 or (0xD0<=m<=0xD7): continue
        L=(data[i]<<8)|data[i+1]; seg=data[i+2:i+L]
        if m==0xC4:
            o=0
            while o < len(seg):
                tc=seg[o]>>4; th=seg[o]&15; bits=list(seg[o+1:o+17]); tot=sum(bits); vals=list(seg[o+17:o+17+tot])
                dht[(tc,th)]=(bits,vals); o+=17+tot
        elif m==0xC0:
            nc=seg[5]; comps=[(seg[6+3*k], seg[7+3*k]>>4, seg[7+3*k]&15) for k in range(nc)]
        elif m==0xDA:
            ns=seg[0]; sel=[(seg[1+2*k], seg[2+2*k]>>4, seg[2+2*k]&15) for k in range(ns)]
            sos=(sel, i+L); break
        i+=L
    scan=bytearray(); j=sos[1]
    while j < len(data):
        b=data[j]
        if b==0xFF:
            n=data[j+1]
            if n==0: scan.append(0xFF); j+=2; continue
            if 0xD0<=n<=0xD7: j+=2; continue
            break
        scan.append(b); j+=1
    return dht, comps, sos[0], bytes(scan)

def build(bits, vals):
    # map (length, code) -> symbol
    code=0; k=0; tab={}
    for l in range(1,17):
        for _ in range(bits[l-1]):
            tab[(l,code)]=vals[k]; k+=1; code+=1
        code<<=1
    return tab

class Dec:
    def __init__(self, data):
        dht, comps, sel, scan = parse(data)
        self.scan=scan; self.nbits=len(scan)*8
        self.bits=np.unpackbits(np.frombuffer(scan,dtype=np.uint8)).tolist()+[1]*64
        # MCU layout for 4:2:0: Y x4, Cb, Cr
        hs=comps[0][1]; vs=comps[0][2]
        self.layout=[]
        for (cid,h,v),(sid,td,ta) in zip(comps, sel):
            n = h*v if len(comps)>1 else 1
            self.layout += [(td,ta)]*n
        self.bpm=len(self.layout)
        self.tabs={k:build(*v) for k,v in dht.items()}
    def sym(self, p, tc, th):
        code=0
        for l in range(1,17):
            code=(code<<1)|self.bits[p+l-1]
            s=self.tabs[(tc,th)].get((l,code))
            if s is not None: return l, s
        return 16, 0   # invalid: treat as EOB / zero
    def run(self, p, blk, k, p_end):
        nblk=0
        while p < p_end:
            td,ta=self.layout[blk]
            if k==0:
                l,s=self.sym(p,0,td); p+=l+s; k=1
            else:
                l,s=self.sym(p,1,ta); r=s>>4; sz=s&15
                p+=l+sz
                if sz==0: k = k+16 if r==15 else 64
                else: k+=r+1
            if k>=64:
                k=0; nblk+=1; blk=(blk+1)%self.bpm
        return (p,blk,k), nblk

def simulate(dec, T=512, guess=None):
    bits=dec.nbits
    S=32*(max(8,(bits+32*T-1)//(32*T))|1)
    nseg=(bits+S-1)//S
    entry=[(i*S, guess(i) if guess else 0, 0) for i in range(nseg)]
    ex=[dec.run(*entry[i], min((i+1)*S, bits+32))[0] for i in range(nseg)]
    rounds=0; work=[]
    while True:
        ch=[i for i in range(1,nseg) if ex[i-1]!=entry[i]]
        if not ch: break
        rounds+=1; work.append(len(ch))
        newex=list(ex)
        for i in ch:
            entry[i]=ex[i-1]
            newex[i]=dec.run(*entry[i], min((i+1)*S, bits+32))[0]
        ex=newex
    return rounds, work, nseg, S, ex

if __name__=='__main__':
    files=sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'sample-images1', '*.jpg')))[:6]
    for f in files:
        d=Dec(open(f,'rb').read())
        r,w,n,S,ex=simulate(d)
        # true states at segment starts
        print(f.split('/')[-1], 'bits',d.nbits,'nseg',n,'S',S,'rounds',r,'work',w)
