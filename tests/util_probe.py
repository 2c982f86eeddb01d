import sys; sys.path.insert(0,'.')
from tests import scenes
def run(spec,W,H,N,depth):
    r=spec.build_product(options=dict(maxDepth=depth))
    r.set_option("countWork",1)
    r.begin_frame(spec.product_camera(),W,H,N); r.render_passes(0,N)
    c=r.counters()
    rays=c['extendPrimary']+c['extendSecondary']+c['shadow']
    print(spec.name,"rays",rays,"nodes/ray %.1f tris/ray %.2f"%(c['nodesVisited']/rays,c['trisTested']/rays),
      "node-step util %.3f"%(c['nodesVisited']/(64*c['waveNodeSteps'])), "leaf steps",c['waveLeafSteps'],"tris per leafstep-lane %.3f"%(c['trisTested']/(64*max(1,c['waveLeafSteps']))), "nodeSteps",c['waveNodeSteps'])
    r.close()
run(scenes.soup_scene(1<<20,0x5EED0001,0.02),1920,1080,1,4)
run(scenes.bunny_scene(),1920,1080,4,12)
