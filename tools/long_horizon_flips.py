import importlib, sys, os, time, hashlib, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, ctypes as C
import oracle_py as O
syn = importlib.import_module("a-loam_amd.synthetic")
g = np.load('/root/repo/tests/golden/reflong_hdl64_c512_seed51.npz')
kw = json.loads(str(g["kwargs"]))
N=int(sys.argv[1]) if len(sys.argv)>1 else 80
scans,R,t,model = syn.make_sequence(str(g["sensor"]), N, seed=int(g["seed"]), **kw)
xs=[s.numpy() for s in scans]
orcs=[O.Oracle(n_scans=64,min_range=5.0,canonical_order=c) for c in (False,True)]
for o in orcs: o.map_config(0.4,0.8)
def counts(o,cls):
    cnt=np.zeros(4851,np.int32); O.lib().orc_map_cube_counts(o.h,cls,O._p(cnt)); return cnt
first={}
for k,x in enumerate(xs):
    dec=[];poses=[];cnts=[];stacks=[]
    for o in orcs:
        O.decision_log(True)
        o.scan_register(x); po=o.odometry_step()
        pm=o.mapping_step(po["q_w"],po["t_w"],o.cloud(O.CLOUD_CORNER_LAST),o.cloud(O.CLOUD_SURF_LAST),o.cloud(O.CLOUD_FULL))
        kd,v,th=O.decisions(); O.decision_log(False)
        dec.append((kd,v,th)); poses.append(pm); cnts.append((counts(o,0),counts(o,1)))
        stacks.append((o.map_cloud(O.MAP_CORNER_STACK), o.map_cloud(O.MAP_SURF_STACK)))
    dt=np.abs(poses[0]["t_w"]-poses[1]["t_w"]).max()
    # per kind: compare outcome sequences
    msgs=[]
    for kind in range(len(O.DECISION_KINDS)):
        a=dec[0][0]==kind; b=dec[1][0]==kind
        oa=dec[0][1][a] > dec[0][2][a]; ob=dec[1][1][b] > dec[1][2][b]
        if a.sum()!=b.sum(): msgs.append(f"{O.DECISION_KINDS[kind]}: count {a.sum()} vs {b.sum()}")
        elif (oa!=ob).any():
            i=int(np.argmax(oa!=ob)); msgs.append(f"{O.DECISION_KINDS[kind]}: outcome #{i} differs: {dec[0][1][a][i]!r} vs {dec[1][1][b][i]!r} thr {dec[0][2][a][i]!r}")
    popdiff=[int((cnts[0][c]!=cnts[1][c]).sum()) for c in (0,1)]
    stackdiff=[(len(stacks[0][c]),len(stacks[1][c])) for c in (0,1)]
    if msgs or any(popdiff) or k%10==0:
        print(k, "dt %.3e"%dt, "cubes with different population", popdiff, "stack sizes", stackdiff, "|", "; ".join(msgs[:4]))
