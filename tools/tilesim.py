import sys, time, collections
sys.path.insert(0, "/root/repo")
import numpy as np
from analiticcl_amd import synth
from oracle import twin as T
d = synth.materialize_golden("/tmp/anxdata")
alpha = T.read_alphabet(d["alphabet"])
A = len(alpha)
words = synth.load_lexicon_words(d["eng"])
def cv_of(text):
    codes = T.normalize_to_alphabet(text, alpha)
    cv = [0]*(A+1)
    for c in codes:
        cv[c if c < A else A] += 1   # UNK code A+1 in norm -> slot A
    return cv, len(codes)
t=time.time()
lexcv = {}
for w in words:
    cv,l = cv_of(w); lexcv[tuple(cv)] = l
slot_freq = np.zeros(A+1, dtype=np.int64)
for cv in lexcv: slot_freq += np.array(cv)
def groups(ng):
    order = sorted(range(A+1), key=lambda s: -slot_freq[s])
    weight=[0]*ng; sg=[0]*(A+1)
    for s in order:
        g=min(range(ng), key=lambda i:(weight[i],i)); sg[s]=g; weight[g]+=slot_freq[s]
    return sg
qs = synth.make_queries(words, 1_000_000, max_len=16, seed=synth.SEED)
print("gen", time.time()-t); t=time.time()
Q = np.zeros((len(qs), A+1), dtype=np.uint8); L=np.zeros(len(qs),dtype=np.int32)
for i,q in enumerate(qs):
    cv,l = cv_of(q); Q[i]=cv; L[i]=l
print("enc", time.time()-t)
kind = Q.max(axis=1); kind[kind>4]=0
np.save("/tmp/Q.npy", Q); np.save("/tmp/L.npy", L)
for ng in (6,):
    sg = np.array(groups(ng))
    S = np.zeros((len(qs), ng), dtype=np.int64)
    for g in range(ng): S[:,g] = Q[:, sg==g].sum(axis=1)
    key = L.astype(np.int64)
    for g in range(ng): key = key*64 + S[:,g]
    k1 = key*8 + kind
    for name,k in (("kind,len,sig",k1),("len,sig",key)):
        u,c = np.unique(k, return_counts=True)
        tiles = ((c+63)//64).sum()
        print(ng, name, "distinct", len(u), "tiles", tiles, "avg q/tile", len(qs)/tiles)
    print("kind hist", np.bincount(kind))
