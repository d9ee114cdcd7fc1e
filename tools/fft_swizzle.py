#!/usr/bin/env python3
"""fft_swizzle.py -- (CPU) LDS bank-conflict count of the unvoiced path's 256-point radix-4 transform pair
(mbelib-neo_amd/csrc/mbx_stream.hip, synth_core) for candidate index maps, under the gfx950 banking rules of
MI355X_MICROARCH.md: ds_read_b64 = 2 groups of 32 lanes on 64 dword banks, ds_write_b64 = 4 groups of 16 lanes on 32.
Prints (LDS-array cycles, conflict-free minimum) for the identity map, the best padded maps and the best XOR swizzles;
the kernel uses e ^ ((e >> 2) & 3) ^ (((e >> 4) & 7) << 2): 200 of 200 (identity: 504)."""
import itertools
def accesses():
    acc=[]  # (kind, [index per lane])
    for r in range(4): acc.append(('w',[l+64*r for l in range(64)]))  # first stage writes
    for q in (16,4,1):
        for r in range(4):
            idx=[(l//q)*4*q+(l%q)+r*q for l in range(64)]
            acc.append(('r',idx)); acc.append(('w',idx))
    for r in range(4): acc.append(('r',[l+64*r for l in range(64)]))  # bins
    for r in range(4):
        idx=[l+64*r for l in range(64)]; acc.append(('r',idx)); acc.append(('w',idx))  # scale
    for q in (1,4,16):
        for r in range(4):
            idx=[(l//q)*4*q+(l%q)+r*q for l in range(64)]
            acc.append(('r',idx)); acc.append(('w',idx))
    for r in range(4): acc.append(('r',[l+64*r for l in range(64)]))
    return acc
ACC=accesses()
def cost(m):
    tot=0; base=0
    for kind,idx in ACC:
        if kind=='r':
            groups=[range(0,32),range(32,64)]; nb=32
        else:
            groups=[range(g*16,g*16+16) for g in range(4)]; nb=16
        for g in groups:
            banks={}
            for l in g:
                e=m(idx[l]); banks.setdefault(e%nb,set()).add(e)
            tot+=max(len(v) for v in banks.values()); base+=1
    return tot,base
def kernel_map(e):
    """the map mbx_stream.hip uses (synth_core, fsw)"""
    return e ^ ((e >> 2) & 3) ^ (((e >> 4) & 7) << 2)


def main():
    print("identity",cost(lambda e:e))
    print("kernel map",cost(kernel_map))
    search()


def search():
    best=[]
    for c4,c5,c6,c7 in itertools.product(range(0,5),repeat=4):
        m=lambda e:e+c4*(e>>4)+c5*(e>>5)+c6*(e>>6)+c7*(e>>7)
        t,b=cost(m); size=m(255)+1
        best.append((t,size,(c4,c5,c6,c7)))
    best.sort(); print(best[:10])
    # xor swizzles
    res=[]
    for s1,t1,m1 in itertools.product(range(1,8),range(0,8),(1,3,7,15)):
      for s2,t2,m2 in itertools.product(range(1,8),range(0,8),(0,1,3,7)):
        def m(e,s1=s1,t1=t1,m1=m1,s2=s2,t2=t2,m2=m2):
            return (e ^ (((e>>s1)&m1)<<t1) ^ (((e>>s2)&m2)<<t2)) & 255
        if len({m(e) for e in range(256)})!=256: continue
        t,b=cost(m); res.append((t,(s1,t1,m1,s2,t2,m2)))
    res.sort(); print(res[:10])


if __name__ == "__main__":
    main()
