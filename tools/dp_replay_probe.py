import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/hybrid-ode-neurips-2021_amd'); sys.path.insert(0,'/root/repo/tests')
import test_hip_dopri5 as T
from oracle.solvers import odeint_dopri5_replay
from oracle.rhs import THETA_NAMES
from hode import adaptive
adaptive.keep_workspace=True
dev=torch.device("cuda:0")
def replay(inp,f,rtol,atol,cot,tape,first):
    f.set_action(inp["actions"]); y0=inp["z0"].clone().requires_grad_(True); f.zero_grad(); st={}
    h=odeint_dopri5_replay(f,y0,inp["t"],rtol,atol,list(zip(tape["t"],tape["dt"])),first,stats=st)
    (h*cot).sum().backward()
    names=list(THETA_NAMES)+(["theta_1","theta_2"] if f.ablate else [])
    out={"h":h.detach(),"gy0":y0.grad,"gtheta":torch.stack([getattr(f,n).grad if getattr(f,n).grad is not None else torch.zeros(()) for n in names])}
    if f.ml_dim>0: out["gw"],out["gb"]=f.ml_net[0].weight.grad,f.ml_net[0].bias.grad
    out["sigma"]=float(st["dt0"].grad) if "dt0" in st else 0.0
    return out
for (D,lanes,N,T_,ablate) in [(12,4,21,20,False),(12,1,21,20,False),(8,4,21,20,False),(4,1,21,20,False),(6,1,21,20,False),(12,0,300,30,False),(8,4,23,14,True)]:
    inp,f=T._setup(N,T_,D,seed=40+D,ablate=ablate)
    cot=torch.randn(T_,N,D,generator=torch.Generator().manual_seed(3))
    hip=T._hip(inp,f,dev,lanes,1e-7,1e-8,cot)
    tape=adaptive.read_tape()
    rep=replay(inp,f,1e-7,1e-8,cot,tape,bool(tape["init"]["first_accepted"]))
    print(D,lanes,N,"acc",hip["stats"],"first",tape["init"]["first_accepted"],"sigma hip %.6e rep %.6e"%(tape["init"]["sigma"],rep["sigma"]),"dh %.2e"%float((hip["h"]-rep["h"]).abs().max()))
    print("   ",{k:"%.2e"%T._rel(hip[k],rep[k]) for k in ("gy0","gw","gb","gtheta") if k in rep})
    # detached
    import hode
    from hode.solver import pack_theta
    from oracle.rhs import dose_schedule
    rep2=replay(inp,f,1e-7,1e-8,cot,tape,False)
    names=list(THETA_NAMES)+(["theta_1","theta_2"] if f.ablate else [])
    scal=[getattr(f,n).detach().clone().to(dev).requires_grad_(True) for n in names]
    y0=inp["z0"].to(dev).requires_grad_(True)
    w=b=None
    if f.ml_dim>0:
        w=f.ml_net[0].weight.detach().clone().to(dev).requires_grad_(True); b=f.ml_net[0].bias.detach().clone().to(dev).requires_grad_(True)
    dosage,times=dose_schedule(inp["actions"],f.step_size)
    h=adaptive.roche_dopri5(y0,pack_theta(scal,dev),w,b,inp["t"].to(dev),dosage.to(dev),times.to(dev),rtol=1e-7,atol=1e-8,ablate=f.ablate,lanes_per_patient=lanes,detach_first_step=True)
    (h*cot.to(dev)).sum().backward()
    print("    detached: gy0 %.2e"%T._rel(y0.grad.cpu(),rep2["gy0"]), "gtheta %.2e"%T._rel(torch.stack([s.grad for s in scal]).cpu(),rep2["gtheta"]), "| effect of term on gy0: %.2e"%T._rel(rep2["gy0"],rep["gy0"]))
