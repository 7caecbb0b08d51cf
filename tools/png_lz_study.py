#!/usr/bin/env python
"""Round 6: what would LZ77 matches buy the GPU PNG encoder, and which GPU-friendly restrictions cost what?  (CPU only; renders
its sample with the oracle.)  Per config-2 tile, with ONE prefix code shared by all tiles (entropy / Huffman-coded bytes):

    runs only (the encoder of rounds 2-5)                                   47.5 / 48.2 kB
    + hash matches searched only at "burst starts", 4 bands, one per 12-byte lane span, 4096 slots     43.7 / 43.9 kB
    the same with 2048 slots / minimum length 5 / every burst of a span / one band  43.76 / 43.66 / 43.65 / 43.60 kB
    (zlib -1 on the same filtered bytes 45.7 kB, zlib -6 42.2 kB)

    python tools/png_lz_study.py [tiles=8]

The kernel (csrc/osmt_pngenc.hip) and its byte-exact model (tests/_png_model.py) implement the first-burst-per-lane variant with a
256-slot table and a ring of eight rows in LDS; the window / table-size sweep behind that choice is in DESIGN.md 3.7."""
import os, sys, zlib, math, collections, heapq
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _png_model as M
from oracle import oracle_py
from osm_renderer_amd import synth
LBASE=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT=[0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE=[1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT=[0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
def len_sym(l):
    for i in range(28,-1,-1):
        if l>=LBASE[i]: return i, LEXT[i]
def dist_sym(d):
    for i in range(29,-1,-1):
        if d>=DBASE[i]: return i, DEXT[i]
def entropy_bits(cnt):
    tot=sum(cnt.values()); return sum(-c*math.log2(c/tot) for c in cnt.values() if c)
def huff_bits(cnt, maxlen=15):
    items=[(c,[s]) for s,c in cnt.items() if c]
    if len(items)<=1: return sum(cnt.values())
    L=collections.Counter(); h=[(c,i,syms) for i,(c,syms) in enumerate(items)]; heapq.heapify(h); k=len(h)
    while len(h)>1:
        a=heapq.heappop(h); b=heapq.heappop(h)
        for s_ in a[2]+b[2]: L[s_]+=1
        heapq.heappush(h,(a[0]+b[0],k,a[2]+b[2])); k+=1
    return sum(cnt[s_]*min(L[s_],maxlen) for s_ in L)

def hash4(b, hbits):
    w=int.from_bytes(b,'little'); return ((w*2654435761)&0xFFFFFFFF)>>(32-hbits)

def encode_tile(rows, span, hbits, minlen, bands, one_per_span=True, maxdist=32768):
    H,NB=rows.shape
    data=rows.tobytes()
    lit=collections.Counter(); dist=collections.Counter(); extra=0; nm=0
    def emit_match(l,d):
        nonlocal extra
        s,e=len_sym(l); lit[257+s]+=1; extra+=e
        s2,e2=dist_sym(d); dist[s2]+=1; extra+=e2
    rows_per_band=H//bands
    for b in range(bands):
        table={}
        for y in range(b*rows_per_band,(b+1)*rows_per_band):
            base=y*NB
            r=data[base:base+NB]
            bursts=[k for k in range(1,NB) if r[k]!=0 and (k==1 or r[k-1]==0)]
            # candidate matches
            cands=[]  # (k, len, dist)
            seen_span=set()
            for k in bursts:
                sp=(k-1)//span
                if one_per_span and sp in seen_span: continue
                if k+4>NB: continue
                h=hash4(r[k:k+4],hbits)
                c=table.get(h)
                if c is None: continue
                d=base+k-c
                if d>maxdist: continue
                ck=c%NB
                mx=min(258,NB-k,NB-ck)
                l=0
                while l<mx and data[c+l]==r[k+l]: l+=1
                if l>=minlen:
                    cands.append((k,l,d)); seen_span.add(sp)
            # greedy selection
            sel=[]; cover=0
            for (k,l,d) in cands:
                if k>=cover: sel.append((k,l,d)); cover=k+l
            # tokens
            i=0; si=0
            while i<NB:
                if si<len(sel) and sel[si][0]==i:
                    emit_match(sel[si][1],sel[si][2]); nm+=1; i+=sel[si][1]; si+=1; continue
                v=r[i]
                lim=sel[si][0] if si<len(sel) else NB
                j=i+1
                while j<lim and r[j]==v: j+=1
                run=j-i; lit[v]+=1; rem=run-1
                while rem>=3:
                    l=min(rem,258); emit_match(l,1); rem-=l
                lit[v]+=rem
                i=j
            for k in bursts:
                if k+4<=NB: table[hash4(r[k:k+4],hbits)]=base+k
    lit[256]+=1
    return lit,dist,extra,nm

def main():
    n=int(sys.argv[1]) if len(sys.argv)>1 else 8
    img=oracle_py.render_batch(synth.config2(n), threads=8)
    cfgs={'runs':dict(span=12,hbits=12,minlen=999,bands=4),
          'gpu':dict(span=12,hbits=12,minlen=4,bands=4),
          'gpu_m5':dict(span=12,hbits=12,minlen=5,bands=4),
          'gpu_h11':dict(span=12,hbits=11,minlen=4,bands=4),
          'gpu_allbursts':dict(span=12,hbits=12,minlen=4,bands=4,one_per_span=False),
          'gpu_1band':dict(span=12,hbits=12,minlen=4,bands=1),
          }
    G={k:[collections.Counter(),collections.Counter(),0,0] for k in cfgs}
    for t in range(n):
        rgb=img[t][...,:3]; f=M.paeth_filter(rgb)
        rows=np.concatenate([np.full((f.shape[0],1),4,np.uint8),f],axis=1)
        for name,c in cfgs.items():
            lit,dist,extra,nm=encode_tile(rows,**c)
            G[name][0].update(lit); G[name][1].update(dist); G[name][2]+=extra; G[name][3]+=nm
    for name,(lit,dist,extra,nm) in G.items():
        print(name,'shared-code bytes/tile: entropy', round((entropy_bits(lit)+entropy_bits(dist)+extra)/8/n,1),'huffman', round((huff_bits(lit)+huff_bits(dist)+extra)/8/n,1),'matches/tile',nm/n)
main()
