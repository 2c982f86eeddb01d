import sys, time; sys.path.insert(0,'.')
from tests import scenes
def run(spec,W,H,N,depth):
    r=spec.build_product(options=dict(maxDepth=depth))
    cam=spec.product_camera()
    best=1e9
    for rep in range(4):
        r.begin_frame(cam,W,H,N); ms=r.render_passes(0,N); best=min(best,ms)
    c=r.counters(); rays=c['extendPrimary']+c['extendSecondary']+c['shadow']
    r.set_option("kernelTiming",1); r.begin_frame(cam,W,H,N); r.render_passes(0,N); kt=r.kernel_times()
    print("  %s %dspp: %.2f ms %.0f Mrays/s  extend %.2f shadow %.2f"%(spec.name,N,best,rays/best/1e3,kt['extend'][0],kt['shadow'][0]),flush=True)
    r.close()
import os
print("TUNE",os.environ.get("RTHIP_TUNE"),flush=True)
run(scenes.soup_scene(1<<20,0x5EED0001,0.02),1920,1080,1,4)
run(scenes.soup_scene(1<<20,0x5EED0001,0.02),1920,1080,8,4)
run(scenes.bunny_scene(),1920,1080,4,12)
