import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.scene import FlatScene
from raytracer_amd.tracer import DeviceWorld, GpuImageTracer, RenderMode
scene = FlatScene.from_json(os.path.join(ROOT, "tests/golden/demo_path.scene.json"))
world = DeviceWorld(scene)
for (w,h,aa,n,rr,md,seq) in [(256,192,1,10,3,2,54),(256,192,1,10,3,2,99),(64,48,1,10,3,5,54)]:
    tr = GpuImageTracer(w,h,scene.camera,aa)
    img = tr.fire_all_rays(world, RenderMode("path", n_rays=n, roulette=rr, max_depth=md, seq_number=seq))
    want, ost = O.render(scene, O.make_cfg(w,h,"path",spp_edge=aa,n_rays=n,roulette=rr,max_depth=md,seq_number=seq))
    d = np.abs(img-want).max(-1)
    print((w,h,aa,n,rr,md,seq), "rays gpu/oracle", tr.last_stats.rays, ost.rays, "exact", (d==0).sum(), ">1e-6", (d>1e-6).sum(), ">1e-4", (d>1e-4).sum(), "max", d.max(), "means", img.mean(), want.mean())
    sd = (img-want).sum(-1); print("   brighter", (sd>1e-4).sum(), "dimmer", (sd<-1e-4).sum())
    idx = np.argsort(d.ravel())[::-1][:5]
    for i in idx:
        r,c = divmod(i, w); print("   px", r, c, img[r,c], want[r,c])
