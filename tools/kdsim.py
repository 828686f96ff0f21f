"""Round-2 experiment (CPU, numpy): how many 64-signature blocks of a tile window pass a bounding-box test, with the signatures in
numeric order and in a k-d order (boxes as tight as a median split makes them), for 6 / 7 / 8 signature groups.  Needs /tmp/Q.npy,
/tmp/L.npy (count vectors and lengths of the bench queries, tools/tilesim.py).  Result: 78-87 % of the blocks pass either way."""
import sys, collections, time
sys.path.insert(0, "/root/repo")
import numpy as np
from analiticcl_amd import synth
from oracle import twin as T
d = synth.materialize_golden("/tmp/anxdata")
alpha = T.read_alphabet(d["alphabet"]); A=len(alpha)
words = synth.load_lexicon_words(d["eng"])
cls = {}
for w in words:
    codes = T.normalize_to_alphabet(w, alpha)
    cv=[0]*(A+1)
    for c in codes: cv[c if c<A else A]+=1
    cls[tuple(cv)] = cls.get(tuple(cv),0)+1
C = np.array(list(cls.keys()), dtype=np.int16); LC = C.sum(axis=1)
slot_freq = C.sum(axis=0)
def groups(ng):
    order = sorted(range(A+1), key=lambda s: -slot_freq[s])
    weight=[0]*ng; sg=[0]*(A+1)
    for s in order:
        g=min(range(ng), key=lambda i:(weight[i],i)); sg[s]=g; weight[g]+=slot_freq[s]
    return np.array(sg)
Q = np.load("/tmp/Q.npy").astype(np.int16); L = np.load("/tmp/L.npy")
rng = np.random.default_rng(0)
idx = rng.choice(len(Q), 1500, replace=False)
def kd_order(P, lo, hi, perm, base):
    # order points perm[lo:hi] (absolute positions base+lo..) so that 64-aligned blocks are tight
    a0 = base + lo; a1 = base + hi
    s = ((a0 + a1)//2 + 32)//64*64
    if s <= a0 or s >= a1: return
    sub = perm[lo:hi]
    pts = P[sub]
    spread = pts.max(axis=0) - pts.min(axis=0)
    dim = int(np.argmax(spread))
    k = s - a0
    part = np.argpartition(pts[:,dim], k-1)  # elements < k are the k smallest
    perm[lo:hi] = sub[part]
    kd_order(P, lo, lo+k, perm, base); kd_order(P, lo+k, hi, perm, base)
for ng in (6,7,8):
    sg = groups(ng)
    S = np.zeros((len(C), ng), dtype=np.int16)
    for g in range(ng): S[:,g] = C[:, sg==g].sum(axis=1)
    # distinct signatures per charcount
    out = {}
    for mode in ("numeric","kd"):
        sigs_all=[]; lens_all=[]
        pos=0
        for c in range(1, 40):
            m = LC==c
            if not m.any(): continue
            u = np.unique(S[m], axis=0)
            # numeric order: group ng-1 most significant
            key = np.zeros(len(u), dtype=np.int64)
            for g in range(ng-1,-1,-1): key = key*256 + u[:,g]
            u = u[np.argsort(key)]
            if mode=="kd":
                perm = np.arange(len(u))
                kd_order(u, 0, len(u), perm, pos)
                u = u[perm]
            sigs_all.append(u); lens_all.append(np.full(len(u), c)); pos += len(u)
        SIG = np.concatenate(sigs_all); SL = np.concatenate(lens_all)
        nb = (len(SIG)+63)//64
        pad = nb*64 - len(SIG)
        SIGp = np.concatenate([SIG, np.full((pad,ng), 999, dtype=np.int16)]); 
        bmin = SIGp.reshape(nb,64,ng).min(axis=1); bmax = np.where(SIGp.reshape(nb,64,ng)==999, -1, SIGp.reshape(nb,64,ng)).max(axis=1)
        starts = np.searchsorted(SL, np.arange(0,42))  # first index with len>=c
        tot_blocks=0; tot_pass=0; tot_match=0
        for i in idx:
            q = Q[i]; lq=int(L[i]); k=min(3, lq//2)
            qs = np.array([q[sg==g].sum() for g in range(ng)])
            s0 = starts[max(1,lq-k)]; s1 = starts[min(41,lq+k+1)]
            b0 = s0//64; b1=(s1+63)//64
            dist = (np.maximum(bmin[b0:b1]-qs,0) + np.maximum(qs-bmax[b0:b1],0)).sum(axis=1)
            tot_blocks += b1-b0; tot_pass += int((dist<=k).sum())
            tot_match += int((np.abs(SIG[s0:s1]-qs).sum(axis=1)<=k).sum())
        print(f"groups {ng} {mode}: sigs {len(SIG)} blocks/query {tot_blocks/len(idx):.0f} pass {tot_pass/len(idx):.0f} matched sigs/query {tot_match/len(idx):.0f}")
