import sys; sys.path.insert(0,".")
import numpy as np, mpr_amd as m
for name,dim,S in (("bear",3,1024),("architecture",3,1024),("prospero",2,1024)):
    c=m.Context(S, flags=m.CTX_COUNTERS); T=np.eye(4,dtype=np.float32); T[3,2]=0.3
    t=m.Tape(m.model(name))
    (c.render3D(t,T) if dim==3 else c.render2D(t))
    print(name, flush=True); c.counters(); c.close()
