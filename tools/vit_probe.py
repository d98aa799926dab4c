import sys, os, json, time, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
from strique_amd.pore_model import pore_model
from strique_amd import hmm, ffi
t=np.load(os.path.join(R,'tests/golden/pore_tables.npz'))
pm=pore_model(table=(t['base_kmer'],t['base_mean'],t['base_stdv']))
cfg=json.load(open(os.path.join(R,'tests/golden/config.json')))
chrom,b,e,repeat,prefix,suffix=cfg['repeat']['c9orf72']
ctx=ffi.Context(0)
fm=hmm.FlankedRepeatModel(repeat,prefix[-50:],suffix[:50],pm,cfg['HMM'])
mid=ctx.model_create(fm.baked)
rng=np.random.default_rng(7)
seq=prefix[-50:]+repeat*500+suffix[:50]
sig=np.clip(pm.generate_signal(seq,noise=True,rng=rng),pm.model_min+.5,pm.model_max-.5)
for nb in (1,2048,4096):
    lg,cg,sg,_=ctx.viterbi_batch(mid,[sig]*nb)
    tm=ctx.last_timing()
    print('batch',nb,'T',len(sig),'ms',tm[0],'per-step us',tm[0]*1e3/len(sig),'reads/s',nb/(tm[0]/1e3),'count',cg[0]+fm.count_bias)
