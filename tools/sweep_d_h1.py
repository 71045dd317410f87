import os, sys, ctypes, torch
sys.path.insert(0, os.getcwd())
from gan_amd import _lib as L, ops
sys.path.insert(0, 'tools')
import kernel_bench as kb
n,h,w_,cb,cs,k,s = (64,32,32,64,128,5,2)
g = ops._conv_geom(n,h,w_,cb,cs,k,k,s,s,'SAME')
big=torch.randn(n,h,w_,cb,device='cuda'); small=torch.randn(n,g.hs,g.ws,cs,device='cuda'); w=torch.randn(k,k,cb,cs,device='cuda')*0.05
dw=torch.empty_like(w); db=torch.empty(cs,device='cuda')
ws=torch.empty(1<<30,dtype=torch.uint8,device='cuda')
flops=2.0*n*g.hs*g.ws*cs*k*k*cb
for p in ['fwd','dgrad','wgrad']:
  for cfg in range(3):
    row=[]
    for sp in (1,2,3,4,6,8,12,16):
        L.knob_set("force_cfg", int(str(cfg))); L.knob_set("force_splits", int(str(sp)))
        us=kb.time_pass(g,p,(big,small,w,dw,db),ws.data_ptr(),1<<30,10)
        row.append('%d:%.0f'%(sp,us))
    print(p,'cfg',cfg,' '.join(row),flush=True)
