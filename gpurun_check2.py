import sys, os, json, time, numpy as np
R=os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0,R)
from strique_amd.pore_model import pore_model
from strique_amd import hmm, ffi
from oracle import strique_oracle as orc
t=np.load(os.path.join(R,'tests/golden/pore_tables.npz'))
pm=pore_model(table=(t['base_kmer'],t['base_mean'],t['base_stdv']))
cfg=json.load(open(os.path.join(R,'tests/golden/config.json')))
chrom,b,e,repeat,prefix,suffix=cfg['repeat']['c9orf72']
ctx=ffi.Context(0)
params=orc.align_params(cfg['align'])
ctx.set_align_params(*[float(v) for v in params])
fm=hmm.FlankedRepeatModel(repeat,prefix[-50:],suffix[:50],pm,cfg['HMM'])
mid=ctx.model_create(fm.baked)
rng=np.random.default_rng(7)
ok=True
# ---- viterbi parity
for nrep,noise in ((5,False),(30,True),(100,True),(400,True)):
    seq=prefix[-50:]+repeat*nrep+suffix[:50]
    sig=pm.generate_signal(seq,samples=8,noise=noise,rng=rng)
    sig=np.clip(sig,pm.model_min+.5,pm.model_max-.5)
    t0=time.time(); lo,po,co=orc.viterbi(fm.baked,sig); to=time.time()-t0
    t0=time.time(); lg,cg,sg,pg=ctx.viterbi(mid,sig,want_path=True); tg=time.time()-t0
    lg2,cg2,sg2,_=ctx.viterbi(mid,sig,want_path=False)
    same=(np.float64(lo).tobytes()==np.float64(lg).tobytes(), co==cg, po is not None and bool((po==pg).all()), lg2==lg and cg2==cg)
    print('viterbi',nrep,noise,len(sig),lo,lg,co+fm.count_bias,cg+fm.count_bias,same,'t_or %.2f t_gpu %.3f'%(to,tg), ctx.last_timing()[:2])
    ok&=all(same)
# no-path case
lg,cg,sg,pg=ctx.viterbi(mid,np.full(50,1e6),want_path=True); lo,po,co=orc.viterbi(fm.baked,np.full(50,1e6))
print('nopath',lg,sg,lo,po is None); ok&= (sg==1 and po is None)
# batch
seqs=[np.clip(pm.generate_signal(prefix[-50:]+repeat*int(k)+suffix[:50],noise=True,rng=rng),pm.model_min+.5,pm.model_max-.5) for k in rng.integers(3,60,40)]
lg,cg,sg,_=ctx.viterbi_batch(mid,seqs)
for i,s in enumerate(seqs):
    lo,po,co=orc.viterbi(fm.baked,s,want_path=False)
    if lo!=lg[i] or co!=cg[i]: ok=False; print('batch mismatch',i,lo,lg[i],co,cg[i])
print('viterbi batch ok', ok, 'ms', ctx.last_timing()[:2], 'total T', sum(map(len,seqs)))
# ---- DP throughput on synthetic reads (host conditioning via oracle for now)
opm=orc.PoreModel.__new__(orc.PoreModel); opm.means=pm._means; opm.model_min=pm.model_min; opm.model_max=pm.model_max
pre_ext=pm.generate_signal(prefix.upper(),samples=6).astype(np.float32); suf_ext=pm.generate_signal(suffix.upper(),samples=6).astype(np.float32)
def make_read(L_nt,nrep):
    left=int(rng.integers(1000,L_nt-300-6*nrep-1000))
    bb=''.join(rng.choice(list('ACGT'),L_nt-300-6*nrep))
    seq=bb[:left]+prefix+repeat*nrep+suffix+bb[left:]
    pa=pm.generate_signal(seq,noise=True,rng=rng)
    return np.round(pa*(8192/1400.0)-10).astype(np.int16)
def cond(raw):
    flt,u8,morph,fltn=orc.condition(raw,opm)
    lv=np.zeros(256,np.float32)
    vals=opm.normalize_minmax(np.concatenate([u8.astype(np.float64),np.arange(256.)]))  # not exact: percentiles change
    return u8
for (L_nt,nrep,nreads) in ((10000,30,64),(50000,1000,16)):
    reads=[make_read(L_nt,nrep) for _ in range(nreads)]
    levels=[];lvals=[]
    for raw in reads:
        flt,u8,morph,fltn=orc.condition(raw,opm)
        lv=np.zeros(256,np.float32)
        # level -> value map of this read, taken from the oracle's own morph output
        lv[:]=np.float32(opm.model_max); uq,idx=np.unique(u8,return_index=True); lv[uq]=morph[idx].astype(np.float32)
        # fill unused levels monotonically so they are harmless
        levels.append(u8); lvals.append(lv)
    # replicate reads to fill the GPU
    rep=max(1,2048//nreads)
    levels_all=np.concatenate(levels*rep); off=np.concatenate([[0],np.cumsum([len(x) for x in levels*rep])]).astype(np.int64)
    lval_all=np.stack(lvals*rep)
    nr=nreads*rep
    align_read=np.repeat(np.arange(nr,dtype=np.int32),2)
    flank=np.concatenate([pre_ext,suf_ext]*nr); foff=(np.arange(2*nr+1)*870).astype(np.int64)
    t0=time.time(); sc,je,j0,rec=ctx.align_batch(levels_all,off,lval_all,align_read,flank,foff); tw=time.time()-t0
    tm=ctx.last_timing()
    cells=sum(871*(len(x)+1) for x in levels*rep)*2
    print('DP',L_nt,'reads',nr,'N~',len(levels[0]),'lut %.1f fwd %.1f trace %.1f ms'%(tm[0],tm[1],tm[2]),'hard',tm[4],'wall %.2f'%tw,' reads/s(kernels) %.0f'%(nr/(tm[3]/1e3)),' GCUPS %.1f'%(cells/(tm[1]+tm[2])/1e6))
    # parity on the first few
    for i in range(3):
        a=lvals[i][levels[i]]
        for f,(fl) in enumerate((pre_ext,suf_ext)):
            o=orc.align_overlap(a,fl,params,want_idx=False)
            g=(sc[2*i+f],rec[(2*i+f)*870:(2*i+f+1)*870],je[2*i+f],j0[2*i+f])
            same=(np.float32(o[0]).tobytes()==np.float32(g[0]).tobytes(), o[4]==g[2], o[5]==g[3], bool((o[3]==g[1]).all()))
            ok&=all(same)
            print('  parity read',i,'flank',f,o[0],g[0],o[4],o[5],same)
print('ALL OK' if ok else 'MISMATCH')
