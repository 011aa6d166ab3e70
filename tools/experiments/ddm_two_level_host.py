"""Host prototype (numpy / scipy, no GPU): CG on the assembled DDM operator of an n^3 BCC cantilever preconditioned by the
diagonal, by the 6 x 6 node blocks, and by either plus a rigid-body (6) or rigid + strain (12) coarse space on aggregates of
agg^3 cells.  Usage: python tools/experiments/ddm_two_level_host.py [n = 12] [agg = 4]"""
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl, sys, time
n=int(sys.argv[1]) if len(sys.argv)>1 else 12
agg=float(sys.argv[2]) if len(sys.argv)>2 else 4.0
d=np.load('/root/repo/tests/golden/Schur_complement_BCC.npz')
S=d['schur_matrices'][4]   # r=0.05
S=0.5*(S+S.T)
# node numbering of the (n+1)^3 corner grid; cell node order: which order does S use? use a guess consistent ordering: we only need a SPD operator of the right structure
idx=lambda i,j,k:(i*(n+1)+j)*(n+1)+k
N=(n+1)**3
xyz=np.array([(i,j,k) for i in range(n+1) for j in range(n+1) for k in range(n+1)],float)
# reference order of the 8 corners (cell.py define_node_order_to_simulate): take from the repo
sys.path.insert(0,'/root/repo')
import json
from pylatticedso_amd.lattice_sim import LatticeSim
base=json.loads(str(np.load('/root/repo/tests/golden/ddm_bcc_4x2x2.npz')['preset_json']))
base["geometry"]["number_of_cells"]=dict(x=n,y=n,z=n)
L=LatticeSim(base, enable_domain_decomposition_solver=True, data_roots=['/root/repo/tests/golden'])
rows=L.cell_boundary_nodes()            # (C, 8) node ids (lattice node ids)
lat=L.lattice
used=np.unique(rows); remap=-np.ones(lat.n_nodes,int); remap[used]=np.arange(len(used))
rows=remap[rows]; xyz=lat.node_xyz[used]; N=len(used)
C=len(rows); m=48
dof=(6*rows[:,:,None]+np.arange(6)[None,None,:]).reshape(C,m)
I=np.repeat(dof,m,axis=1).ravel(); J=np.tile(dof,(1,m)).ravel()
A=sp.coo_matrix((np.tile(S.ravel(),C),(I,J)),shape=(6*N,6*N)).tocsr()
fixed=np.zeros((N,6),bool); fixed[xyz[:,0]==xyz[:,0].min()]=True
free=~fixed.ravel()
Af=A[free][:,free]
b=np.zeros((N,6)); b[xyz[:,0]==xyz[:,0].max(),2]=-1.0; b=b.ravel()[free]
def cg(M,name):
    it=[0]
    def cb(x): it[0]+=1
    t=time.time(); x,info=spl.cg(Af,b,rtol=1e-8,maxiter=20000,M=M,callback=cb); print(f"{name:40s} iterations {it[0]:5d} info {info} {time.time()-t:.1f}s",flush=True)
print('n',n,'free dofs',Af.shape[0])
cg(None,'plain')
D=Af.diagonal(); cg(spl.LinearOperator(Af.shape,lambda r:r/D),'Jacobi')
# node-block Jacobi
nb6=Af.shape[0]
Afull=A.tolil()
blocks=np.zeros((N,6,6))
Ad=A.tocsr()
for i in range(N):
    blocks[i]=Ad[6*i:6*i+6,6*i:6*i+6].toarray()
fm=fixed
for i in range(N):
    for k in range(6):
        if fm[i,k]: blocks[i,k,:]=0; blocks[i,:,k]=0; blocks[i,k,k]=1
binv=np.linalg.inv(blocks)
def bj(r):
    full=np.zeros(6*N); full[free]=r
    z=np.einsum('nij,nj->ni',binv,full.reshape(N,6)).ravel()
    return z[free]
cg(spl.LinearOperator(Af.shape,bj),'node-block Jacobi')
# coarse space: aggregates of agg^3 cells worth of nodes, rigid body modes
lo=xyz.min(0); a=np.floor((xyz-lo)/agg-1e-9).astype(int); a=np.maximum(a,0)
na=a.max(0)+1; aid=(a[:,0]*na[1]+a[:,1])*na[2]+a[:,2]; nagg=aid.max()+1
cen=np.array([xyz[aid==q].mean(0) for q in range(nagg)])
rr=xyz-cen[aid]
def Zrows(modes):
    r_,c_,v_=[],[],[]
    for i in range(N):
        x,y,z=rr[i]; base=modes*aid[i]
        ent=[(0,0,1),(1,1,1),(2,2,1),(3,3,1),(4,4,1),(5,5,1),(1,3,-z),(2,3,y),(0,4,z),(2,4,-x),(0,5,-y),(1,5,x)]
        if modes==12:
            ent+= [(0,6,x),(1,7,y),(2,8,z),(0,9,.5*y),(1,9,.5*x),(1,10,.5*z),(2,10,.5*y),(0,11,.5*z),(2,11,.5*x)]
        for dd,mm,v in ent:
            r_.append(6*i+dd); c_.append(base+mm); v_.append(v)
    return sp.coo_matrix((v_,(r_,c_)),shape=(6*N,modes*nagg)).tocsr()
for modes in (6,12):
    Z=Zrows(modes)[free]
    Ac=(Z.T@Af@Z).toarray(); Ac+=1e-12*np.eye(len(Ac))*np.trace(Ac)/len(Ac)
    Aci=np.linalg.pinv(Ac)
    cg(spl.LinearOperator(Af.shape,lambda r:r/D+Z@(Aci@(Z.T@r))),f'Jacobi + {modes}-mode coarse ({nagg} aggregates)')
    cg(spl.LinearOperator(Af.shape,lambda r:bj(r)+Z@(Aci@(Z.T@r))),f'block Jacobi + {modes}-mode coarse')
