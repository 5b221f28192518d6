"""What two-hop relays through idle peers would buy the halo exchange (host-side simulation on the traffic matrices of
profiles/traffic_matrix_L32.md): every message is cut into 16 chunks, each chunk takes the path -- direct, or owner -> k -> receiver --
that keeps max(round-1 link load) + max(round-2 link load) smallest (greedy, largest chunks first).  Rounds are sequential (a relay
forwards what it has fully received).  python profiles/relay_sim.py"""
import re, sys, itertools
import numpy as np
import os
txt=open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'traffic_matrix_L32.md')).read()
def mats():
    out={}
    for blk in txt.split('## ')[1:]:
        head=blk.splitlines()[0]
        m=re.match(r"L=(\d+) P=(\d+) mode=(\w+)", head)
        L,P,mode=int(m.group(1)),int(m.group(2)),m.group(3)
        rows=[l for l in blk.splitlines() if re.match(r"\| \d+ \|", l)]
        M=np.zeros((P,P))
        for l in rows:
            c=[x.strip() for x in l.strip().strip('|').split('|')]
            r=int(c[0])
            for q,x in enumerate(c[1:]): M[r,q]=0.0 if x=='·' else float(x)
        out[(L,P,mode)]=M
    return out
def route(M, C=16):
    P=len(M)
    l1=np.zeros((P,P)); l2=np.zeros((P,P))   # load[src][dst] per round
    pairs=sorted([(M[r,o],o,r) for r in range(P) for o in range(P) if M[r,o]>0], reverse=True)
    chunks=[]
    for sz,o,r in pairs:
        for c in range(C): chunks.append((sz/C,o,r))
    chunks.sort(reverse=True)
    plan={}
    for sz,o,r in chunks:
        best=None
        def cost(a1,a2):
            return a1.max()+a2.max()
        # direct
        l1[o,r]+=sz; cd=(cost(l1,l2), l1[o,r], -1); l1[o,r]-=sz
        best=cd
        for k in range(P):
            if k in (o,r): continue
            l1[o,k]+=sz; l2[k,r]+=sz
            ck=(cost(l1,l2), max(l1[o,k],l2[k,r]), k)
            l1[o,k]-=sz; l2[k,r]-=sz
            if ck[:2]<best[:2]: best=ck
        k=best[2]
        if k<0: l1[o,r]+=sz
        else: l1[o,k]+=sz; l2[k,r]+=sz
        plan.setdefault((o,r),[]).append(k)
    return l1,l2,plan
for key,M in mats().items():
    if key[2]!='class': continue
    l1,l2,plan=route(M)
    relayed=sum(1 for v in plan.values() for k in v if k>=0)/max(1,sum(len(v) for v in plan.values()))
    print(key, "direct max %.3f GB -> rounds %.3f + %.3f = %.3f GB; relayed chunks %.0f%%; total bytes x%.2f" % (M.max(), l1.max(), l2.max(), l1.max()+l2.max(), 100*relayed, (l1.sum()+l2.sum())/M.sum()))
