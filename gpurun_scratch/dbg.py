import ctypes as C, numpy as np, sys, torch
sys.path.insert(0,".")
from mtscomp_amd import hip
L=hip.lib(); nc=385; rate=30000; n=8
raw=torch.empty((n*rate,nc),dtype=torch.int16,device="cuda")
for k in range(n): L.mts_dev_synth_int16(0,None,C.c_void_p(raw[k*rate:].data_ptr()),k*rate,(k+1)*rate,nc,0)
bound=(hip.compress_bound(rate*nc*2)+255)//256*256
cbuf=torch.empty(n*bound,dtype=torch.uint8,device="cuda")
b=np.arange(n+1,dtype=np.int64)*rate; sl=np.arange(n,dtype=np.int64)*bound; sz=np.zeros(n,dtype=np.int64)
lp=lambda a:a.ctypes.data_as(C.POINTER(C.c_long))
rc=L.mts_dev_compress_chunks(0,None,C.c_void_p(raw.data_ptr()),nc,2,lp(b),n,5,6,C.c_void_p(cbuf.data_ptr()),lp(sl),lp(sz))
h=(C.c_ulonglong*16)(); L.mts_dbg_read(h)
g=h[0]; print("groups",g,"iters/group",h[1]/g,"pops/lane",h[2]/g/64,"maxpops/group",h[3]/g,"improvements/lane",h[4]/g/64)
