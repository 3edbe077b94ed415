#!/usr/bin/env python3
"""LDS bank-conflict model of conv2_pool_x3_kernel's B-operand gather (robustbnns_amd/csrc/rbnn_conv.hip, ConvX3Img): for every (tap,
16-position tile, 16-lane service group of a ds_read_b128) count how many lanes hit the same 16-byte slot of the 256-byte bank row;
the cost is the mean number of passes per read (1.0 = conflict-free).  Used to pick the image pitch and the octet swizzle:
1x28x28 (O2W 8, P1W 12): pitch 12 + row parity in octet bit 0 -> 1.0 (was 2.0); 3x32x32 (O2W 10, P1W 14): pitch 18, idle lanes spread
over positions 0..11 -> 1.14 (was 2.86).  Service groups as measured for rbnn_common.hpp's row_swz: {0-3,12-15,20-27},{4-11,16-19,28-31},+32."""
import random, sys
GROUPS=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
GROUPS=GROUPS+[[l+32 for l in g] for g in GROUPS]
def build(O2W,P1W,NPOS,pitch,wrap):
    NPT=(NPOS+15)//16
    reads=[]
    for tap in range(25):
        for pt in range(NPT):
            for g in GROUPS:
                r=[]
                for lane in g:
                    li=lane&15; lg=lane>>4
                    pos=pt*16+li
                    if pos>=NPOS: pos=(pos-NPOS) if wrap else 0
                    y,x=pos//O2W+tap//5, pos%O2W+tap%5
                    r.append((y*pitch+x,lg))
                reads.append(r)
    return reads
def cost(reads,sig):
    tot=0
    for r in reads:
        cnt={}
        for p,lg in r:
            slot=(p*4+(lg^sig[p]))%16
            cnt[slot]=cnt.get(slot,0)+1
        tot+=max(cnt.values())
    return tot/len(reads)
O2W,P1W,NPOS=10,14,100
for pitch in (14,15,16,18):
  for wrap in (0,1):
    reads=build(O2W,P1W,NPOS,pitch,wrap)
    n=pitch*P1W+8
    forms={"cur":lambda p:((p>>2)&1)<<1, "row^":lambda p:(((p>>2)&1)<<1)^((p//pitch)&1), "row&3":lambda p:(p//pitch)&3,
           "x>>2&1<<1 ^ y&1":lambda p:((((p%pitch)>>2)&1)<<1)^((p//pitch)&1), "(x>>2 + y)&3":lambda p:(((p%pitch)>>2)+(p//pitch))&3,
           "(x>>2)&3":lambda p:((p%pitch)>>2)&3, "((x>>2)^y)&3": lambda p:(((p%pitch)>>2)^(p//pitch))&3, "(x>>2)*? +2y":lambda p:(((p%pitch)>>2)+2*(p//pitch))&3,
           "(x>>1 ^ y)&3":lambda p:(((p%pitch)>>1)^(p//pitch))&3}
    res={k:round(cost(reads,[f(p) for p in range(n)]),3) for k,f in forms.items()}
    # local search over table
    random.seed(1)
    sig=[forms["row^"](p) for p in range(n)]
    c=cost(reads,sig)
    for it in range(6000):
        i=random.randrange(n); old=sig[i]; sig[i]=random.randrange(4)
        c2=cost(reads,sig)
        if c2<=c: c=c2
        else: sig[i]=old
    print("cifar pitch",pitch,"wrap",wrap,res,"local-search table:",round(c,3))

# 1x28x28
reads=build(8,12,64,12,1)
n=12*12+8
print("mnist pitch 12: current", cost(reads,[((p>>2)&1)<<1 for p in range(n)]), " + row parity", cost(reads,[(((p>>2)&1)<<1)^((p//12)&1) for p in range(n)]))
