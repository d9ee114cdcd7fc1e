#!/usr/bin/env python3
"""Development aid (CPU, uses the oracle): how far is the REFERENCE's own voiced-bank arithmetic from exact evaluation?
The reference (and the oracle restating it) advances each harmonic's oscillator by 160 float rotations; its rounding
error grows with the sample index.  This script takes one frame of a seeded random-bit stream, replays the windowed
oscillators once as that float recurrence and once exactly (float64, direct cosine), and prints the difference.  On the
loudest frames (harmonic amplitudes >10x the clip level) it reaches 0.4-0.5 at the end of the frame -- the size and
the place of the largest HIP-vs-oracle differences tools/err_tail.py finds (the HIP path evaluates the phasors
directly and sits near the exact value).
usage: tools/ref_rounding_noise.py [codec S T seed stream frame]   (defaults: the worst frame of err_tail.py 2 4096 16)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle_lib
from mbelib_neo_amd import framegen
from mbelib_neo_amd.layout import load_tables_blob, table_views
o=oracle_lib.load()
a_=[int(x) for x in sys.argv[1:]]
codec,S,T,seed=(a_+[2,4096,16,4242][len(a_):])[:4] if len(a_)<4 else a_[:4]
rng=framegen.rng_for(seed)
frames=framegen.random_frames(codec,S*T,rng).reshape(S,T,-1)
s,t=(a_[4],a_[5]) if len(a_)>=6 else (2251,13)
fr=frames[s]
def run(Tn):
    st=o.init_state(1); r=o.rng_seeded([99+7*s])
    return o.process_batch(codec,1,Tn,fr[:Tn].reshape(Tn,-1).copy(),st,r)
a=run(t); b=run(t+1)
st13=a["state"].reshape(3); st14=b["state"].reshape(3)
pcm=np.asarray(b["pcmf"]).reshape(-1,160)[t]
cur=st14[2]; prev=st13[2]   # enhanced slots: final cur of frame t, enhanced model of frame t-1
print("cur L",cur["L"],"w0",cur["w0"],"prev L",prev["L"],"w0",prev["w0"], "peak", np.abs(pcm).max())
Ws=table_views(load_tables_blob())["ws"]
N=160
f32=np.float32
cw0=f32(cur["w0"]); pw0=f32(prev["w0"])
cL=int(cur["L"]); pL=int(prev["L"]); maxl=max(cL,pL)
cM=cur["Ml"].astype(f32).copy(); pM=prev["Ml"].astype(f32).copy()
cV=cur["Vl"].copy(); pV=prev["Vl"].copy()
if cL>pL:
    pM[pL+1:maxl+1]=0; pV[pL+1:maxl+1]=1
else:
    cM[cL+1:maxl+1]=0; cV[cL+1:maxl+1]=1
cPHI=cur["PHIl"].astype(f32); pPHI=prev["PHIl"].astype(f32)
out32=np.zeros(N,f32); out64=np.zeros(N)
n=np.arange(N)
big=[]
for l in range(1,maxl+1):
    cv=cV[l]==1; pv=pV[l]==1
    if not cv and not pv: continue
    cw0l=f32(cw0*f32(l)); pw0l=f32(pw0*f32(l))
    if l<8 and cv and pv and abs(cw0-pw0)<f32(0.1)*cw0:
        continue   # interpolated branch: same direct evaluation in both
    for (v,g,th,ph0,w) in ((pv,f32(2)*pM[l],pw0l,pPHI[l],Ws[n+N]),(cv,f32(2)*cM[l],cw0l,f32(cPHI[l]-f32(cw0l*f32(N))),Ws[n])):
        if not v: continue
        big.append(float(g))
        # float32 recurrence as in the reference
        sd=f32(np.sin(f32(th))); cd=f32(np.cos(f32(th))); sp=f32(np.sin(f32(ph0))); cp=f32(np.cos(f32(ph0)))
        for i in range(N):
            out32[i]=f32(out32[i]+f32(f32(g*w[i])*cp))
            c2=f32(f32(cp*cd)-f32(sp*sd)); s2=f32(f32(sp*cd)+f32(cp*sd)); cp,sp=c2,s2
        out64+=float(g)*w.astype(np.float64)*np.cos(float(ph0)+float(th)*n)
d=out32.astype(np.float64)-out64
print("harmonics",len(big),"max 2M",max(big),"sum 2M",sum(big))
print("voiced bank: float recurrence - exact: max |d| %.4f at n=%d ; d[153]=%.4f ; rms %.4f" % (np.abs(d).max(), int(np.argmax(np.abs(d))), d[153], np.sqrt(np.mean(d*d))))
print("|d| by segment:", [round(float(np.abs(d[i:i+20]).max()),4) for i in range(0,160,20)])
