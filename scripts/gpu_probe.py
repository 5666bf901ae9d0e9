import sys, time, numpy as np
sys.path.insert(0,'.')
from debwt_amd import api, synth
for name in ("pan_16M_4","pan_100M_4","chr1_250M"):
    t0=time.time(); recs = synth.make_workload(name); t1=time.time()
    d = api.DeBWT(k=32); d.load_records(recs); t2=time.time()
    for it in range(3):
        ta=time.time(); d.build(); tb=time.time()
        st=d.stats()
        print(name, "it",it,"wall %.1f ms"%((tb-ta)*1e3), {k:(round(v,2) if isinstance(v,float) else v) for k,v in st.items()}, flush=True)
    w,h,dr = d.fetch()
    if name!="chr1_250M":
        rc,inv = api.verify_inverse(w, st['n'], h, dr); print("inverse rc",rc, "gen %.1fs load %.1fs"%(t1-t0,t2-t1))
    d.close()
