"""trips of the iter_long sweep per haystack and per group of 64 lanes, on the CPU: the round-4 form (raw records) against the compact form with and
without kind 3 / `below` (profiles/r5_experiments.md §3).  python tools/long_trips.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import numpy as np
import test_iter_long_plan_cpu as T
from helpers import build_pair
from pyahocorasick_amd.workloads import dna_workload
from oracle import orc
keys, reads = dna_workload(100_000, 64*24, 150, seed=0)
A,_ = build_pair(keys)
dk, dv, reals, longest = T.long_dictionary(A)
OD = orc.Oracle()
for k,v in zip(dk,dv): OD.add_word(k,v)
OD.make_automaton()
def trips_old(ends, vals):
    # copy of sweep_records_lockstep counting trips
    INT_MIN=-(1<<31); r=0;k=0;n=len(ends);prev_e=INT_MIN;prev_len=0;path=p=last_e=last_i=k_last=0;reach=longest-1;t=0
    if n==0: return 0
    while True:
        eor=1 if k>=n else 0
        if eor&(path^1): break
        t+=1
        kk=n-1 if eor else k
        e,v=ends[kk],vals[kk]&0xFFFFFFFF
        kind,ln=v>>30,(v>>24)&63
        start=e-ln+1
        is_fe,is_ev=int(kind==2),int(kind!=0)
        p_end=path&(eor|int(e>p+reach)); p_hit=path&(p_end^1)&is_ev&int(start==p); p_fe,p_e=p_hit&is_fe,p_hit&(is_fe^1)
        s_act=(path^1)&(eor^1); longer_in=int(prev_e==e)&int(e-prev_len+1>=r)
        fires=s_act&is_ev&int(start>=r)&(longer_in^1); s_fe,s_e=fires&is_fe,fires&(is_fe^1)
        emit=p_end|p_fe|s_fe
        ox=last_e if p_end else e
        r=ox+1 if emit else r
        keep=p_e|s_e
        if keep: last_e=e
        k_next=k_last+1 if p_end else k+1
        if keep: k_last=k
        if s_e: p=start
        seen=s_act|p_fe
        prev_e=INT_MIN if p_end else (e if seen else prev_e); prev_len=ln if seen else prev_len
        path=(path&(p_end^1)&(p_fe^1))|s_e
        k=k_next
    return t
def trips_new(ends, vals, use_below=True, use_k3=True):
    INT_MIN=-(1<<31); comp=[]
    for i,(e,v) in enumerate(zip(ends,vals)):
        v&=0xFFFFFFFF
        if v>>30==0: continue
        st=e-((v>>24)&63)+1; up=INT_MIN
        if i>0 and ends[i-1]==e: up=e-(((vals[i-1]&0xFFFFFFFF)>>24)&63)+1
        comp.append((st,up,v))
    r=0;k=0;n=len(comp);path=p=last_e=k_last=limit=0;reach=longest-1;t=0
    if n==0: return 0,0
    while True:
        eor=1 if k>=n else 0
        if eor&(path^1): break
        t+=1
        kk=n-1 if eor else k
        start,up,v=comp[kk]; kind=v>>30; e=start+((v>>24)&63)-1
        now=(kind>>1) if use_k3 else int(kind==2)
        p_end=path&(eor|int(e>limit)); p_hit=path&(p_end^1)&int(start==p)
        fires=(path^1)&(eor^1)&int(start>=r)&int(up<r); hit=p_hit|fires
        emit=p_end|(hit&now); keep=hit&(now^1)
        ox=last_e if p_end else e
        r=ox+1 if emit else r
        if keep: last_e=e; limit=(e+((v>>18)&63)) if use_below else start+reach
        k_next=k_last+1 if p_end else k+1
        if keep: k_last=k
        if fires&keep: p=start
        path=(path&(emit^1))|(fires&keep)
        k=k_next
    return t,n
to=[];tn=[];tn2=[];tn3=[];nc=[];nr=[]
for h in range(len(reads)):
    recs=OD.iter(bytes(reads[h])); e=[a for a,_ in recs]; v=[b for _,b in recs]
    to.append(trips_old(e,v)); a,n=trips_new(e,v); tn.append(a); nc.append(n); nr.append(len(e))
    tn2.append(trips_new(e,v,use_below=False)[0]); tn3.append(trips_new(e,v,use_below=False,use_k3=False)[0])
to=np.array(to).reshape(-1,64); tn=np.array(tn).reshape(-1,64); tn2=np.array(tn2).reshape(-1,64); tn3=np.array(tn3).reshape(-1,64)
print("raw records/read %.1f compact %.1f"%(np.mean(nr),np.mean(nc)))
for name,t in (("old",to),("new(no k3,no below)",tn3),("new(k3)",tn2),("new(k3+below)",tn)):
    print("%-22s mean trips/haystack %.1f   mean over groups of max-over-64 lanes %.1f"%(name,t.mean(),t.max(axis=1).mean()))
