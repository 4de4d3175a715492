import numpy as np, sys
sys.path.insert(0,'.')
from strelka_amd import capi, synth
from oracle import pyoracle
sys.path.insert(0,'tests')
from tests.test_gpu_parity import _varied_pileups
capi.init(0)
rng=np.random.default_rng(201)
pb=_varied_pileups(rng)
got=capi.dependent_eprob(pb); want=pyoracle.adjust_joint_eprob(pb)
rel=np.abs(got-want)/np.abs(want)
print('de: max rel',rel.max(),'n>2e-6',(rel>2e-6).sum(),'n>1e-5',(rel>1e-5).sum(),'n>1e-3',(rel>1e-3).sum(),'exact frac',np.mean(got==want))
bad=np.argsort(-rel)[:10]
locus=np.searchsorted(pb.call_off,bad,side='right')-1
for b,l in zip(bad,locus): print(b,l,got[b],want[b],rel[b],'depth',pb.call_off[l+1]-pb.call_off[l],'q',pb.calls[b]&63)
rng=np.random.default_rng(202)
pb=_varied_pileups(rng); pb.de=pyoracle.adjust_joint_eprob(pb)
pb.ploidy=rng.choice(np.array([1,2,2,2],np.uint8),pb.n_loci); pb.ref_base[::97]=4
g=capi.site_digt_call(pb); w=pyoracle.site_digt_call(pb,pb.de)
d=np.abs(g['strand_bias']-w['strand_bias']); s=np.maximum(1,np.abs(w['strand_bias']))
i=np.argsort(-(d/s))[:8]
for k in i: print('sb',k,g['strand_bias'][k],w['strand_bias'][k],'max_gt',g['genome']['max_gt'][k],w['genome']['max_gt'][k],'lh',g['lhood'][k][:4],w['lhood'][k][:4], 'snpq', g['genome']['snp_qphred'][k], w['genome']['snp_qphred'][k])
dl=np.abs(g['lhood']-w['lhood']); print('lhood max abs',np.nanmax(dl), 'max rel', np.nanmax(dl/np.maximum(1,np.abs(w['lhood']))))
