"""What the rounds of k_jpeg_huff wait for (CPU model, tools/sim/huff_sync_sim.py): per segment, how many of the six
block-within-MCU phase hypotheses (entry = segment start, k = 0, blk = h) end in the TRUE exit state, and how many rounds a
phase oracle for the guess would save.
    python3 tools/sim/huff_sync_hypotheses.py"""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from huff_sync_sim import *
files=sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'sample-images1', '*.jpg')))[2:5]
for f in files:
    d=Dec(open(f,'rb').read())
    r,w,n,S,ex=simulate(d)
    true_entry=[(0,0,0)]+ex[:-1]
    bits=d.nbits
    # (A) guess = true blk of the entry state (phase oracle), p aligned, k = 0
    rA,wA,_,_,_=simulate(d, guess=lambda i: true_entry[i][1])
    # (A2) guess = true blk + 1 if the entry is mid-block (k>0): the next block boundary's blk
    rA2,wA2,_,_,_=simulate(d, guess=lambda i: (true_entry[i][1] + (1 if true_entry[i][2]>0 else 0)) % d.bpm)
    # (B) hypotheses: for each segment, which h in 0..5 gives the true exit from (iS, h, 0)?
    good=[]
    for i in range(1,n):
        hs=[h for h in range(d.bpm) if d.run(i*S,h,0,min((i+1)*S,bits+32))[0]==ex[i]]
        good.append(len(hs))
    import collections
    print(f.split('/')[-1], 'base rounds',r,'sum work',sum(w), '| phase-oracle rounds',rA,'work',wA[:8],'| next-boundary oracle',rA2,wA2[:8],'| #hyp giving true exit:',sorted(collections.Counter(good).items()))
