"""wr_render_frame_static (the entry point that allocates nothing) against wr_render_frame,
through the C ABI: bit-identical frames on two calls in a row, the second with a smaller
input (smaller target, shadow maps and scene), then the first input again, with the device's
free memory unchanged across every call; sizes beyond the static work buffer are refused with
WR_ERR_CAPACITY; static frames of two contexts queued without any wait between them, and
static frames between the frames of a context with three frames in flight, all equal the
frames a waiting context renders."""
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
    torch.cuda.synchronize()
    # the property the entry point exists for: the device's free memory is the same before the
    # call, after it has been queued and after it has run (hipMemGetInfo; no hipMalloc / hipFree)
    free_before = torch.cuda.mem_get_info()[0]
    r.render_frame_static(cam, light, glob, passes, out.data_ptr())
    free_queued = torch.cuda.mem_get_info()[0]
    r.sync()
    torch.cuda.synchronize()
    assert free_before == free_queued == torch.cuda.mem_get_info()[0]
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


def test_static_frames_of_two_contexts_without_a_wait_between_them():
    """Two contexts on one GPU share the GPU's one static array.  Their static frames are queued
    back to back with no wr_sync anywhere; the library orders them on the device, so each equals
    the frame its context renders the ordinary way."""
    import torch
    sc_a = scene.sponza_standin(detail=0.1, tex_size=32)
    sc_b = scene.sponza_standin(detail=0.05, tex_size=16)
    wa, ha, sa = 320, 180, 256
    wb, hb, sb = 256, 144, 256
    ra = abi.Renderer(wa, ha, sa)
    ra.upload_scene(sc_a)
    rb = abi.Renderer(wb, hb, sb, frames_in_flight=2)
    rb.upload_scene(sc_b)
    jobs = []
    for k in range(6):
        r, w, h = (ra, wa, ha) if k % 2 == 0 else (rb, wb, hb)
        cam, light, glob = _inputs(w, h, 2.0 + 0.7 * k, pose=scene.flythrough_pose(5 * k + 1, 64))
        passes = abi.PASS_ALL if k != 3 else abi.PASS_FOG | abi.PASS_LIGHT_DEBUG
        r.render_frame(cam, light, glob, passes)
        jobs.append((r, w, h, cam, light, glob, passes, r.read_color()))
    outs = [torch.full((h, w, 4), 13, dtype=torch.uint8, device="cuda") for (_, w, h, *_rest) in jobs]
    torch.cuda.synchronize()
    for rounds in range(2):  # the second round finds the array full of the first one's data
        for (r, w, h, cam, light, glob, passes, _), out in zip(jobs, outs):
            r.render_frame_static(cam, light, glob, passes, out.data_ptr())
    ra.sync(); rb.sync()
    torch.cuda.synchronize()
    for k, ((_, _, _, _, _, _, _, want), out) in enumerate(zip(jobs, outs)):
        assert np.array_equal(out.cpu().numpy(), want), k
    ra.close()
    rb.close()


def test_static_frames_between_the_frames_of_a_context_with_three_in_flight():
    """frames_in_flight = 3: a static frame borrows the streams of the first set of frame buffers
    while frames of that set may still be running.  Ordinary and static frames alternate without a
    wait; every static frame and every ordinary frame that is looked at equals the frame of a
    context that waits after every call."""
    import torch
    sc = scene.sponza_standin(detail=0.1, tex_size=32)
    w, h, s = 320, 180, 256
    ref = abi.Renderer(w, h, s)
    ref.upload_scene(sc)
    r = abi.Renderer(w, h, s, frames_in_flight=3)
    r.upload_scene(sc)

    def frame(k):
        cam, light, glob = _inputs(w, h, 1.5 + 0.4 * k, pose=scene.flythrough_pose(3 * k + 2, 64))
        passes = [abi.PASS_ALL, abi.PASS_ALL, abi.PASS_SHADOWS | abi.PASS_LIGHT_DEBUG, abi.PASS_FOG | abi.PASS_LIGHT_DEBUG][k % 4]
        return cam, light, glob, passes

    def want(k):
        ref.render_frame(*frame(k))
        return ref.read_color()

    r.render_frame(*frame(0))
    r.sync()
    plan = "fsffsfsssffsf"  # f: wr_render_frame, s: wr_render_frame_static
    outs = {}
    for look in (False, True):  # first nothing is read until the end, then every ordinary frame is
        for k, op in enumerate(plan):
            if op == "s":
                outs[k] = torch.full((h, w, 4), 91, dtype=torch.uint8, device="cuda")
                r.render_frame_static(*frame(k), outs[k].data_ptr())
            else:
                r.render_frame(*frame(k))
                if look:
                    assert np.array_equal(r.read_color(), want(k)), k
        last = max(k for k, op in enumerate(plan) if op == "f")
        assert np.array_equal(r.read_color(), want(last))
        torch.cuda.synchronize()
        for k, out in outs.items():
            assert np.array_equal(out.cpu().numpy(), want(k)), (look, k)
    ref.close()
    r.close()
