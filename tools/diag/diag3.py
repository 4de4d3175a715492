import numpy as np, sys, os
os.environ['SK_DEBUG_SKIP_GLOBAL_PASS']='1'
sys.path.insert(0,'.')
from strelka_amd import capi, synth
capi.init(0)
rng=np.random.default_rng(1000)
pb=synth.pileups(1<<16,rng)
out,_=capi.site_digt_call_fused(pb)
bad=np.where(out['is_called']==0xffffffff)[0]
print('n sentinel',len(bad),'of',pb.n_loci)
for l in bad[:6]:
    c=pb.calls[pb.call_off[l]:pb.call_off[l+1]]
    grp=((c>>10)&1)+2*((c>>6)&15)
    print(l,'depth',len(c),'groups',np.bincount(grp,minlength=8).tolist(),'q<3',int(((c&63)<3).sum()), 'block pos', l%128)
print(np.bincount(bad%128,minlength=128)[:20])
