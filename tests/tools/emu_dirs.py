"""Iteration counts of the physics solver on a tree of phys_optim_in_* directories (e.g. what tests/tools/pipe_phys_stats.py copied off the GPU box), on the CPU through the
host emulation of the kernel source, one process per core.

    python tests/tools/emu_dirs.py ROOT FRAMES [video ...]        (W=<workers> in the environment; CHD_EMU_* study switches are passed on to the emulation)
"""
import sys, os, time, json, numpy as np, multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_emu'))
def work(args):
    d,F=args
    import chd_amd, emu
    from chd_amd.phys_capi import default_config
    from chd_amd.io_formats import read_inputs
    seq=read_inputs(d,F)
    e=emu.EmuProblem(seq, default_config())
    t0=time.time(); e.solve(0,4); st,_=e.results()
    fb=int(st[4][0])!=0
    if fb and e.rebuild_fallback(): e.solve(5,5); st,_=e.results()
    return os.path.basename(d), [(int(s[0]),int(s[1])) for s in st[:6 if fb else 5]], [float(s[4]) for s in st[:6 if fb else 5]], time.time()-t0
if __name__=='__main__':
    root=sys.argv[1]; F=int(sys.argv[2]); vids=sys.argv[3:] or sorted(v for v in os.listdir(root) if v.startswith('video'))
    import emu; emu.build()
    with mp.get_context('spawn').Pool(int(os.environ.get('W','6'))) as pool:
        res=pool.map(work,[(os.path.join(root,v),F) for v in vids],chunksize=1)
    tot=0
    for k,st,obj,dt in sorted(res,key=lambda r:-sum(s[1] for s in r[1])):
        it=sum(s[1] for s in st); tot+=it
        print(k,it,st,'obj %.5f'%obj[-1],'%.0fs'%dt)
    print('total',tot,'max',max(sum(s[1] for s in r[1]) for r in res))
