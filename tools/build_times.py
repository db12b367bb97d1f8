import time, os, sys
sys.path.insert(0, os.getcwd())
import tray_racing_amd as T
v,c=T.gen_scene("bistro",0,1)
for rep in range(2):
    for label,fn in (("medium_build", lambda: T.flat_build(v,c,use_tlas=False,preset="medium_build")),("ploc default", lambda: T.flat_build_params(v,c,T.build_params(),use_tlas=False))):
        t0=time.time(); flat=fn(); print(label,"total %.2f s"%(time.time()-t0), "nodes",flat.n_nodes, flush=True)
