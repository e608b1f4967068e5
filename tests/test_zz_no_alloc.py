"""wr_render_frame_static (the entry point that allocates nothing) against wr_render_frame,
through the C ABI: bit-identical frames on two calls in a row, the second with a smaller
input (smaller target, shadow maps and scene), then the first input again; sizes beyond the
static work buffer are refused with WR_ERR_CAPACITY."""
import numpy as np
import pytest

import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import abi, scene

pytestmark = pytest.mark.gpu


def _inputs(w, h, t, pose=None):
    rc = scene.REFERENCE_CAMERA
    pos, pitch, yaw = pose if pose else (rc["position"], rc["pitch"], rc["yaw"])
    return abi.camera_uniform(pos, pitch, yaw, rc["fovy"], w, h), abi.light_at_time(t), abi.global_uniforms(t, 1)


def _both(r, w, h, cam, light, glob, passes):
    import torch
    r.render_frame(cam, light, glob, passes)
    want = r.read_color()
    out = torch.full((h, w, 4), 77, dtype=torch.uint8, device="cuda")
    r.render_frame_static(cam, light, glob, passes, out.data_ptr())
    r.sync()
    torch.cuda.synchronize()
    return want, out.cpu().numpy()


def test_static_frame_equals_allocating_frame():
    big_scene = scene.sponza_standin(detail=0.1, tex_size=32)
    small_scene = scene.sponza_standin(detail=0.05, tex_size=16)
    wa, ha, sa = 320, 180, 256
    wb, hb, sb = 200, 112, 128
    ra = abi.Renderer(wa, ha, sa)
    ra.upload_scene(big_scene)
    rb = abi.Renderer(wb, hb, sb)
    rb.upload_scene(small_scene)
    cam_a, light_a, glob_a = _inputs(wa, ha, 2.0)
    # the scene must be resident before the call that may not allocate
    with pytest.raises(abi.WrError) as e:
        import torch
        ra.render_frame_static(cam_a, light_a, glob_a, abi.PASS_ALL, torch.zeros(wa * ha, dtype=torch.int32, device="cuda").data_ptr())
    assert e.value.code == abi.WR_ERR_NO_SCENE
    want, got = _both(ra, wa, ha, cam_a, light_a, glob_a, abi.PASS_ALL)
    assert (want[..., :3].max(axis=2) > 0).mean() > 0.5
    assert np.array_equal(want, got)
    # second call in a row, smaller input: nothing of the first call's contents may show
    cam_b, light_b, glob_b = _inputs(wb, hb, 5.5, pose=((-1400.0, 40.0, -600.0), 20.0, 35.0))
    want_b, got_b = _both(rb, wb, hb, cam_b, light_b, glob_b, abi.PASS_ALL)
    assert np.array_equal(want_b, got_b)
    # without the shadow passes the static call must not see the previous call's shadow maps
    want_c, got_c = _both(rb, wb, hb, cam_b, light_b, abi.global_uniforms(5.5, 0), abi.PASS_FOG | abi.PASS_LIGHT_DEBUG)
    assert np.array_equal(want_c, got_c)
    # and the first input again
    want2, got2 = _both(ra, wa, ha, cam_a, light_a, glob_a, abi.PASS_ALL)
    assert np.array_equal(want2, want) and np.array_equal(got2, want)
    ra.close()
    rb.close()


def test_static_frame_refuses_sizes_beyond_its_buffer():
    import torch
    r = abi.Renderer(1920, 1080, 2048)
    r.upload_scene(scene.sponza_standin(detail=0.05, tex_size=16))
    cam, light, glob = _inputs(1920, 1080, 2.0)
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    out = torch.zeros(1920 * 1080, dtype=torch.int32, device="cuda")
    with pytest.raises(abi.WrError) as e:
        r.render_frame_static(cam, light, glob, abi.PASS_ALL, out.data_ptr())
    assert e.value.code == abi.WR_ERR_CAPACITY and "bytes" in str(e.value)
    r.close()
