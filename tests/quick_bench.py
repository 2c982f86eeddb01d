import sys, time; sys.path.insert(0,'.')
import numpy as np
from tests import scenes
import raytracer_amd as ra
def run(spec,W,H,N,depth,reps=3):
    r=spec.build_product(options=dict(maxDepth=depth))
    f=r.flat(); print(spec.name,"tris",f['info'].numTris,"nodes",f['info'].numNodes,"depth",f['info'].maxDepthBlas,"build %.2fs"%f['build_seconds'],flush=True)
    cam=spec.product_camera()
    for rep in range(reps):
        r.begin_frame(cam,W,H,N)
        t=time.time(); ms=r.render_passes(0,N); wall=(time.time()-t)*1e3
        c=r.counters(); rays=c['extendPrimary']+c['extendSecondary']+c['shadow']
        print("  %dx%d %dspp depth%d: dev %.2f ms wall %.2f ms  rays %.1fM  -> %.1f Mrays/s"%(W,H,N,depth,ms,wall,rays/1e6,rays/ms/1e3), c['hits'][:6],flush=True)
    r.set_option("kernelTiming",1)
    r.begin_frame(cam,W,H,N); r.render_passes(0,N); print("  kernels",r.kernel_times(),flush=True)
    r.close()
run(scenes.cube_scene(),1920,1080,16,12)
run(scenes.cube_scene(),1920,1080,16,1)
run(scenes.bunny_scene(),1920,1080,4,12)
s=scenes.soup_scene(1<<20,0x5EED0001,0.02); run(s,1920,1080,1,4)
