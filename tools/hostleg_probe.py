import os,sys,time,numpy as np
sys.path.insert(0,'/root/repo'); os.chdir('/root/repo')
import bench
from strique_amd.counter import repeatCounter
pm,cfg=bench.load_inputs()
counter=repeatCounter(pm,align_config=cfg["align"],HMM_config=cfg["HMM"],device=0)
counter.add_target("c9orf72",*cfg["repeat"]["c9orf72"][3:6])
sigs,strands,nreps=bench.make_batch(pm,cfg,1024,50000,0)
sigs=sigs*4; strands=strands*4
tids=[counter._classifier_for("c9orf72",s).target_id for s in strands]
reps=3
big=np.concatenate(sigs*reps); off2=np.zeros(reps*len(sigs)+1,np.int64); off2[1:]=np.cumsum([len(s) for s in sigs]*reps)
tids2=np.array(list(tids)*reps,np.int32)
ctx=counter.ctx
ctx.detect_batch(big,off2,tids2,None)
os.environ['STRQ_DEBUG']='1'
t=time.time(); ctx.detect_batch(big,off2,tids2,None); print('host leg',time.time()-t)
