import numpy as np, sys
sys.path.insert(0,'.')
from strelka_amd import capi, synth
from oracle import pyoracle
from tests.test_gpu_parity import _varied_pileups
capi.init(0)
rng=np.random.default_rng(201)
pb=_varied_pileups(rng)
got=capi.dependent_eprob(pb); want=pyoracle.adjust_joint_eprob(pb)
rel=np.abs(got-want)/np.abs(want)
bad=np.where(got!=want)[0]
print('n mismatch',len(bad))
for b in bad[:8]:
    l=np.searchsorted(pb.call_off,b,side='right')-1
    s,e=pb.call_off[l],pb.call_off[l+1]
    c=pb.calls[s:e]
    grp=((c>>10)&1)+2*((c>>6)&15)
    g=grp[b-s]
    m=(grp==g)&(((c>>12)&1)==0)&((c&63)>=3)
    idx=np.where(m)[0]
    print('locus',l,'call',b-s,'group',g,'q',(c[idx]&63).tolist(),'nmm',((c[idx]>>11)&1).tolist())
    print(' got ',got[s:e][idx].tolist())
    print(' want',want[s:e][idx].tolist())
