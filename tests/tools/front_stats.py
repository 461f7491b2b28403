import sys, os, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT+'/tests', ROOT+'/tests/golden', ROOT+'/tests/host_emu'): sys.path.insert(0,p)
import chd_amd, emu, numpy as np
import make_bench_parity_golden as G
from chd_amd.phys_capi import default_config
from chd_amd.synth import make_walk
L=emu.lib()
L.emu_front_profile.restype=C.c_int; L.emu_front_profile.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_void_p,C.c_int]
mxs=[]
def prof(seq, tag):
    e=emu.EmuProblem(seq, default_config(max_iter=G.CAPS))
    for st in range(5):
        sz=e.sizes(st)
        if not sz['valid']: continue
        out=(C.c_int*(3*2048))()
        n=L.emu_front_profile(e.h, st, 32, out, 2048)
        a=np.array(out[:3*n]).reshape(n,3)
        front=a[:,0]+a[:,1]+a[:,2]
        print(tag, 'stage',st,'N',sz['n']+sz['m'],'Nb',sz['Nb'],'bc',sz['bc'],'w',sz['w'],'panels',n,'front max',front.max(),'mean %.0f'%front.mean(),'band act max',a[:,0].max(),'border act max',a[:,1].max())
        mxs.append(front.max())
for s in [0,1,2,3,5,8,13,21,34,55,89,100,1688,88]:
    prof(G.make_case(s,90,0.0),'s%d'%s)
print('max over all', max(mxs))
from chd_amd.io_formats import read_inputs
for v in (0,64,100,150):
    prof(read_inputs(os.path.join(os.environ.get('CHD_CLIPS', '/tmp/pipe192'), 'video_%03d/phys_optim_in_combined' % v), 100),'clip%d'%v)
prof(G.make_case(0,60,0.0),'F60')
