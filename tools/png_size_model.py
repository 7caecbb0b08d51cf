#!/usr/bin/env python
"""What would per-tile dynamic Huffman tables buy the GPU PNG encoder?  (CPU only, uses the oracle to render.)

For 8 config-2 tiles: the Paeth-filtered scanlines are tokenised the way k_png_encode does (literals + distance-1
runs), then sized (a) with the fixed Huffman code the kernel emits, (b) with the optimal prefix code for the tile's
own literal/length histogram (+ header), and (c) compressed by zlib level 6, which also finds real LZ77 matches.
Result (round 2): fixed 59 KB, dynamic 47 KB, zlib-6 42 KB per tile — dynamic tables are worth 20 %, not the 45 % the
gap to a full encoder suggests; see DESIGN.md 3.7.
(Round 3 built the cheaper cousin: ONE code fitted to map tiles and shared by every tile, tools/make_png_huffman.py.)"""
import heapq
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osm_renderer_amd import synth
from oracle import oracle_py
dl = synth.config2(8)
img = oracle_py.render_batch(dl, threads=8)
print(img.shape, img.dtype)
def paeth_filter(rgb):
    H,W,C = rgb.shape
    out = np.zeros((H, W*C), np.uint8)
    a = rgb.astype(np.int32)
    left = np.zeros_like(a); left[:,1:] = a[:,:-1]
    up = np.zeros_like(a); up[1:] = a[:-1]
    ul = np.zeros_like(a); ul[1:,1:] = a[:-1,:-1]
    p = left+up-ul
    pa=np.abs(p-left); pb=np.abs(p-up); pc=np.abs(p-ul)
    pred = np.where((pa<=pb)&(pa<=pc), left, np.where(pb<=pc, up, ul))
    return ((a-pred)&255).astype(np.uint8).reshape(H, W*C)
LBASE=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT=[0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
def len_sym(l):
    for i in range(28,-1,-1):
        if l>=LBASE[i]: return 257+i, LEXT[i]
def tokens(data):
    # literals + dist-1 runs (greedy, like the GPU encoder: run of equal bytes after the first)
    lit=np.zeros(286,int); extra=0; nmatch=0
    i=0; n=len(data)
    while i<n:
        j=i+1
        while j<n and data[j]==data[i]: j+=1
        run=j-i
        lit[data[i]]+=1
        rem=run-1
        while rem>=3:
            l=min(rem,258)
            if rem-l in (1,2): l-=3-(rem-l) if l-(3-(rem-l))>=3 else 0
            s,e=len_sym(l); lit[s]+=1; extra+=e; nmatch+=1; rem-=l
        lit[data[i]]+=rem
        i=j
    lit[256]+=1
    return lit, extra, nmatch
def fixed_bits(lit, extra, nmatch):
    b=0
    for s,c in enumerate(lit):
        if s<144: b+=8*c
        elif s<256: b+=9*c
        elif s<280: b+=7*c
        else: b+=8*c
    return b+extra+5*nmatch+3
def huff_bits(freq, maxlen=15):
    h=[(f,i) for i,f in enumerate(freq) if f>0]
    if len(h)==1: return h[0][0]
    heapq.heapify(h); depth={}
    nodes={i:[i] for _,i in h}; nid=1000
    while len(h)>1:
        f1,a=heapq.heappop(h); f2,b=heapq.heappop(h)
        for s in nodes[a]+nodes[b]: depth[s]=depth.get(s,0)+1
        nodes[nid]=nodes[a]+nodes[b]; heapq.heappush(h,(f1+f2,nid)); nid+=1
    return sum(freq[s]*d for s,d in depth.items()), max(depth.values())
tot_fixed=tot_dyn=tot_z=0
for t in range(img.shape[0]):
    rgb = img[t][...,:3]
    f = paeth_filter(rgb)
    rows = np.concatenate([np.full((f.shape[0],1),4,np.uint8), f],axis=1).reshape(-1)
    lit,extra,nm = tokens(rows.tolist())
    fb = fixed_bits(lit,extra,nm)
    db,ml = huff_bits(lit)
    db += extra + nm*1 + 3 + 14 + 19*3 + 300  # one dist symbol (1 bit), header approx
    z = len(zlib.compress(rows.tobytes(),6))
    print(t, "fixed KB", fb/8/1024, "dynamic KB", db/8/1024, "maxlen", ml, "zlib6 KB", z/1024)
