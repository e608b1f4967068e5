import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import wgpu_renderer_amd as W
from wgpu_renderer_amd import abi, scene
from oracle_lib import Oracle
from parity import compare
detail=float(sys.argv[1]) if len(sys.argv)>1 else 0.25
w,h,S = 480,270,512
sc = scene.sponza_standin(detail=detail, tex_size=64)
print('tris', sc.geometry.triangle_count, flush=True)
cam = abi.camera_uniform(scene.REFERENCE_CAMERA['position'],0.0,0.0,55.0,w,h)
light = abi.light_at_time(2.0)
for passes, use in ((abi.PASS_LIGHT_DEBUG,0),(abi.PASS_ALL,1)):
    g = abi.global_uniforms(2.0,use)
    o = Oracle(w,h,S); o.upload_scene(sc)
    t0=time.time(); o.render_frame(cam, light, g, passes, threads=16); print('oracle s', time.time()-t0, o.counters(), flush=True)
    r = abi.Renderer(w,h,S); r.upload_scene(sc)
    r.set_counting(True)
    t0=time.time(); r.render_frame(cam, light, g, passes); r.sync(); print('gpu s', time.time()-t0, flush=True)
    st=r.stats(); print({k:getattr(st,k) for k,_ in st._fields_}, flush=True)
    m = compare(o.read_visibility(), r.read_visibility(), o.read_color(), r.read_color())
    print(passes, m, flush=True)
    print('depth exact', float((o.read_depth()==r.read_depth()).mean()), flush=True)
    if passes & abi.PASS_SHADOWS:
        a,b=o.read_shadow_map(), r.read_shadow_map()
        print('shadow exact', float((a==b).mean()), 'maxdiff', float(np.abs(a-b).max()), flush=True)
    np.save(f'gpurun_out/gpu_color_{passes}.npy', r.read_color()); np.save(f'gpurun_out/ora_color_{passes}.npy', o.read_color())
    r.close(); o.close()
