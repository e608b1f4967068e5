"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the
same inputs.

Bar (BASELINE.json north_star): triangle-id coverage equal on >= 99.9 % of pixels with
differences only on edges, colour PSNR >= 45 dB and max abs error <= 2/255 on non-edge
pixels.  Because both sides follow one arithmetic contract the tests first assert the
stronger property that holds today -- visibility, depth and shadow maps bit-identical --
and then the tolerance on colour.
"""
import os

import numpy as np
import pytest

import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import abi, scene
from golden_inputs import GOLDEN, cube_config1, standin_small
from oracle_lib import Oracle
from parity import compare

pytestmark = pytest.mark.gpu

PSNR_MIN = 45.0      # dB, north_star
MAX_ERR_NON_EDGE = 2  # /255, north_star
ID_MATCH_MIN = 0.999


def render_gpu(sc, cam, light, glob, passes, w, h, s, counting=False, **kw):
    r = abi.Renderer(w, h, s, **kw)
    r.upload_scene(sc)
    r.set_counting(counting)
    r.render_frame(cam, light, glob, passes)
    r.sync()
    out = dict(color=r.read_color(), depth=r.read_depth(), vis=r.read_visibility())
    if passes & abi.PASS_SHADOWS:
        out["shadow"] = r.read_shadow_map()
    if counting:
        st = r.stats()
        out["counters"] = dict(fog_fragments=int(st.fog_fragments), fog_noise_evals=int(st.fog_noise_evals),
                               replay_pixels=int(st.replay_pixels), replay_slow_pixels=int(st.replay_slow_pixels))
    r.close()
    return out


def render_oracle(sc, cam, light, glob, passes, w, h, s, threads=16):
    o = Oracle(w, h, s)
    o.upload_scene(sc)
    o.render_frame(cam, light, glob, passes, threads=threads)
    out = dict(color=o.read_color(), depth=o.read_depth(), vis=o.read_visibility(), counters=o.counters())
    if passes & abi.PASS_SHADOWS:
        out["shadow"] = o.read_shadow_map()
    o.close()
    return out


def live_evals(o):
    """fog_noise evaluations whose result reaches the frame: the oracle runs every march of
    fog.wgsl and counts separately the light marches of fog points in shadow, whose result is
    never read; the GPU path does not run those."""
    return o["counters"]["fog_noise_evals"] - o["counters"]["fog_noise_evals_unused"]


def assert_parity(g, o, exact=True):
    m = compare(o["vis"], g["vis"], o["color"], g["color"])
    assert m["id_match"] >= ID_MATCH_MIN and m["id_mismatch_off_edge"] == 0, m
    assert m["psnr"] >= PSNR_MIN and m["max_err_non_edge"] <= MAX_ERR_NON_EDGE, m
    if exact:
        assert np.array_equal(g["vis"], o["vis"]), m
        assert np.array_equal(g["depth"], o["depth"])
        if "shadow" in g:
            assert np.array_equal(g["shadow"], o["shadow"])
    return m


def default_inputs(w, h, t=2.0, use_shadows=1, pose=None):
    rc = scene.REFERENCE_CAMERA
    pos, pitch, yaw = pose if pose else (rc["position"], rc["pitch"], rc["yaw"])
    return abi.camera_uniform(pos, pitch, yaw, rc["fovy"], w, h), abi.light_at_time(t), abi.global_uniforms(t, use_shadows)


def test_all_passes_small_scene(small_scene):
    w, h, s = 480, 270, 512
    cam, light, glob = default_inputs(w, h)
    g = render_gpu(small_scene, cam, light, glob, abi.PASS_ALL, w, h, s, counting=True)
    o = render_oracle(small_scene, cam, light, glob, abi.PASS_ALL, w, h, s)
    assert_parity(g, o)
    # executed march steps are integer work: must agree exactly
    assert g["counters"]["fog_fragments"] == o["counters"]["fog_fragments"] > 0
    assert g["counters"]["fog_noise_evals"] == live_evals(o) > 0
    assert o["counters"]["blended_lt1"] > 0  # the ordered-blend path was exercised


def test_pbs_only_small_scene(small_scene):
    # BASELINE configs[1] shape: no shadow passes, use_shadowmaps = 0, no fog (the wasm32 build)
    w, h, s = 480, 270, 128
    cam, light, glob = default_inputs(w, h, use_shadows=0)
    g = render_gpu(small_scene, cam, light, glob, abi.PASS_LIGHT_DEBUG, w, h, s)
    o = render_oracle(small_scene, cam, light, glob, abi.PASS_LIGHT_DEBUG, w, h, s)
    assert_parity(g, o)


def test_cube_config1_full_size():
    # BASELINE configs[0]: Cube.glb at 1280x720, PBS only
    sc, cam, light, glob, passes, w, h, s = cube_config1(1280, 720)
    g = render_gpu(sc, cam, light, glob, passes, w, h, s)
    o = render_oracle(sc, cam, light, glob, passes, w, h, s)
    m = assert_parity(g, o)
    assert (g["vis"] != 0xFFFFFFFF).sum() > 20000, m


@pytest.mark.parametrize("name,maker", [("cube_frame_160x90", lambda: cube_config1(160, 90)),
                                        ("standin_frame_192x108", lambda: standin_small(192, 108))])
def test_committed_golden_frames(name, maker):
    sc, cam, light, glob, passes, w, h, s = maker()
    g = render_gpu(sc, cam, light, glob, passes, w, h, s)
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    o = dict(vis=gold["visibility"], depth=gold["depth"], color=gold["color"])
    assert_parity(dict(vis=g["vis"], depth=g["depth"], color=g["color"]), o)
    if passes & abi.PASS_SHADOWS:
        assert np.array_equal(g["shadow"][3], gold["shadow_face3"])


@pytest.mark.parametrize("w,h,s", [(333, 207, 200), (64, 64, 64), (65, 17, 96), (1, 1, 1)])
def test_ragged_target_sizes(tiny_scene, w, h, s):
    cam, light, glob = default_inputs(w, h)
    g = render_gpu(tiny_scene, cam, light, glob, abi.PASS_ALL, w, h, s)
    o = render_oracle(tiny_scene, cam, light, glob, abi.PASS_ALL, w, h, s, threads=4)
    assert_parity(g, o)


POSES = [
    ((-1400.0, 40.0, -600.0), 20.0, 35.0),     # corner of the hall, looking up and across
    ((0.0, 900.0, 0.0), -80.0, 0.0),           # from the ceiling straight down (fog top face fills the view)
    ((-500.0, 30.0, 0.0), 0.0, 0.0),           # inside the fog box: no fog fragments
    ((-1250.0, 100.0, -262.0), 0.0, 90.0),     # inside a column: near-plane clipping everywhere
    ((1400.0, 500.0, 0.0), -10.0, 180.0),      # upper tier looking back down the nave
    ((-500.0, 150.0, 0.0), 0.0, 180.0),        # facing the near wall
]


@pytest.mark.parametrize("pose", POSES)
def test_camera_poses(small_scene, pose):
    w, h, s = 320, 180, 256
    cam, light, glob = default_inputs(w, h, pose=pose)
    g = render_gpu(small_scene, cam, light, glob, abi.PASS_ALL, w, h, s)
    o = render_oracle(small_scene, cam, light, glob, abi.PASS_ALL, w, h, s)
    assert_parity(g, o)


@pytest.mark.parametrize("t", [0.0, 7.3, 123.0, 400.0, 90000.0])
def test_light_animation_times(tiny_scene, t):
    # t = 0 is the black light (|sin 0| = 0, state.rs:615-617); larger t moves light and fog drift.
    # The fog drift 0.2 t also moves the noise domain: at t = 90000 the finest octaves' hash
    # arguments pass 2^30, where the sine rule returns 0, the range-checked instantiation of the
    # noise runs and the domain is beyond what the hash tables index.
    w, h, s = 256, 144, 256
    cam, light, glob = default_inputs(w, h, t=t)
    g = render_gpu(tiny_scene, cam, light, glob, abi.PASS_ALL, w, h, s)
    o = render_oracle(tiny_scene, cam, light, glob, abi.PASS_ALL, w, h, s, threads=8)
    assert_parity(g, o)


def _quad(x0, x1, y0, y1, z, uv=1.0):
    v = np.zeros((4, 14), np.float32)
    v[:, 0:3] = [[x0, y0, z], [x1, y0, z], [x1, y1, z], [x0, y1, z]]
    v[:, 3:5] = [[0, 0], [uv, 0], [uv, uv], [0, uv]]
    v[:, 5:8] = [0, 0, 1]
    v[:, 8:11] = [1, 0, 0]
    v[:, 11:14] = [0, 1, 0]
    return v, np.array([0, 1, 2, 0, 2, 3], np.uint32)


def blend_order_scene(order):
    """Opaque back wall plus three overlapping translucent / cut-out quads drawn in `order`
    (far-to-near, near-to-far or mixed): the final colour depends on which fragments passed
    the depth test when they were drawn (state.rs:466 blending + depth writes, no sorting)."""
    rng = np.random.default_rng(3)
    tex = []
    for color, alpha in (((200, 60, 40), 255), ((40, 200, 60), 128), ((40, 60, 220), 64), ((230, 230, 60), None)):
        a = np.empty((16, 16, 4), np.uint8)
        a[..., :3] = color
        a[..., 3] = alpha if alpha is not None else np.where(rng.random((16, 16)) > 0.5, 255, 0)
        flat = np.zeros((16, 16, 4), np.uint8)
        flat[..., :] = (128, 128, 255, 255)
        mr = np.zeros((16, 16, 4), np.uint8)
        mr[..., 1], mr[..., 3] = 180, 255
        tex += [(a, True), (flat, False), (mr, False)]
    mats = [scene.Material(3 * i, 3 * i + 1, 3 * i + 2, 1.0, 1.0) for i in range(4)]
    quads = {
        0: (_quad(-300, 300, -200, 200, -100.0), 0),      # opaque wall
        1: (_quad(-200, 100, -150, 100, -50.0, 3.0), 1),  # alpha 128
        2: (_quad(-50, 250, -100, 150, 0.0, 3.0), 2),     # alpha 64, nearest
        3: (_quad(-150, 150, -50, 180, -25.0, 5.0), 3),   # cut-out, bilinear taps give fractional alpha
    }
    meshes = [scene.Mesh(quads[k][0][0], quads[k][0][1], quads[k][1]) for k in order]
    cube = scene.cube_model()
    return scene.Scene(tex, scene.Model(meshes, mats), cube, scene.cube_model(),
                       abi.instance_raw((0.0, 0.0, 0.0)), abi.instance_raw((0.0, -500.0, 0.0), (1, 0, 0, 0), (10.0, 10.0, 10.0)))


@pytest.mark.parametrize("order", [(0, 1, 3, 2), (2, 3, 1, 0), (1, 0, 2, 3), (3, 2, 0, 1)])
def test_ordered_blending_follows_draw_order(order):
    w, h, s = 320, 240, 64
    sc = blend_order_scene(order)
    cam = abi.camera_uniform((0.0, 0.0, 500.0), 0.0, -90.0, 55.0, w, h)  # looking down -z
    light = abi.light_uniform((0.0, 100.0, 300.0), (1.0, 1.0, 1.0, 250000.0))
    glob = abi.global_uniforms(1.0, 0)
    g = render_gpu(sc, cam, light, glob, abi.PASS_LIGHT_DEBUG, w, h, s)
    o = render_oracle(sc, cam, light, glob, abi.PASS_LIGHT_DEBUG, w, h, s, threads=2)
    assert_parity(g, o)
    assert np.array_equal(g["color"], o["color"])  # blending is 8-bit round trips: exact
    assert o["counters"]["blended_lt1"] > 1000
    assert (o["color"][..., 3] < 255).any()  # alpha is blended too (pass.rs:79-82)


def test_ordered_blending_many_layers():
    """More translucent layers on a pixel than the per-pixel fragment list holds (16): those
    pixels go through the general replay, the others through the list path; both must agree
    with the oracle exactly."""
    w, h, s = 200, 150, 64
    base = blend_order_scene((0,))
    rng = np.random.default_rng(11)
    order = rng.permutation(22)
    meshes = [base.geometry.meshes[0]]
    for k in order:
        x0 = -250 + 12 * int(k)
        q, idx = _quad(x0, x0 + 260, -180 + 5 * int(k), 150, -90.0 + 4.0 * int(k), 2.0)
        meshes.append(scene.Mesh(q, idx, 1 + int(k) % 2))
    sc = scene.Scene(base.textures, scene.Model(meshes, base.geometry.materials), base.fog, base.marker,
                     base.geometry_instance, base.fog_instance)
    cam = abi.camera_uniform((0.0, 0.0, 500.0), 0.0, -90.0, 55.0, w, h)
    light = abi.light_uniform((0.0, 100.0, 300.0), (1.0, 1.0, 1.0, 250000.0))
    glob = abi.global_uniforms(1.0, 0)
    g = render_gpu(sc, cam, light, glob, abi.PASS_LIGHT_DEBUG, w, h, s, counting=True)
    o = render_oracle(sc, cam, light, glob, abi.PASS_LIGHT_DEBUG, w, h, s, threads=2)
    assert_parity(g, o)
    assert np.array_equal(g["color"], o["color"])
    assert g["counters"]["replay_slow_pixels"] > 100
    assert g["counters"]["replay_pixels"] > g["counters"]["replay_slow_pixels"]


def test_fog_volume_with_overlapping_parts(tiny_scene):
    """A fog model that is not convex (a small slab floating above the box): pixels covered by
    two fog fragments are marched and blended once per fragment, in draw order."""
    w, h, s = 240, 135, 128
    cube = scene.cube_model().meshes[0]
    slab = cube.vertices.copy()
    slab[:, 0:3] = slab[:, 0:3] * np.float32(0.4) + np.array([0.0, 1.8, 0.0], np.float32)
    fog = scene.Model([cube, scene.Mesh(slab, cube.indices.copy(), 0)], [])
    sc = scene.Scene(tiny_scene.textures, tiny_scene.geometry, fog, tiny_scene.marker,
                     tiny_scene.geometry_instance, tiny_scene.fog_instance)
    for pose in (((0.0, 900.0, 0.0), -80.0, 0.0), ((-900.0, 500.0, 100.0), -25.0, 0.0)):
        cam, light, glob = default_inputs(w, h, pose=pose)
        g = render_gpu(sc, cam, light, glob, abi.PASS_ALL, w, h, s, counting=True)
        o = render_oracle(sc, cam, light, glob, abi.PASS_ALL, w, h, s, threads=4)
        assert_parity(g, o)
        assert np.array_equal(g["color"], o["color"])
        assert g["counters"]["fog_fragments"] == o["counters"]["fog_fragments"] > 0
        assert g["counters"]["fog_noise_evals"] == live_evals(o)
    g1 = render_gpu(tiny_scene, cam, light, glob, abi.PASS_ALL, w, h, s, counting=True)
    assert g["counters"]["fog_fragments"] > g1["counters"]["fog_fragments"]  # the slab added fragments


@pytest.mark.parametrize("in_flight", [2, 3])
def test_frames_in_flight_match_one(small_scene, in_flight):
    """Consecutive frames on rotating sets of frame buffers, overlapping on the device, give
    the same pixels as one frame at a time; the light moves between frames, so every frame
    needs its own shadow maps."""
    import torch
    w, h, s = 400, 225, 256
    poses = [scene.flythrough_pose(k, 64) for k in (0, 9, 21, 40, 63)]
    cams = [default_inputs(w, h, pose=p)[0] for p in poses]
    lights = [abi.light_at_time(1.0 + 0.7 * k) for k in range(len(cams))]
    globs = [abi.global_uniforms(1.0 + 0.7 * k, 1) for k in range(len(cams))]
    one = abi.Renderer(w, h, s)
    one.upload_scene(small_scene)
    ref = []
    for c, l, g in zip(cams, lights, globs):
        one.render_frame(c, l, g, abi.PASS_ALL)
        ref.append((one.read_color(), one.read_depth(), one.read_shadow_map()))
    one.close()
    two = abi.Renderer(w, h, s, frames_in_flight=in_flight)
    two.upload_scene(small_scene)
    for n in (2, 3, 5):  # submit n frames back to back, then look at the last one
        for c, l, g in list(zip(cams, lights, globs))[:n]:
            two.render_frame(c, l, g, abi.PASS_ALL)
        assert np.array_equal(two.read_color(), ref[n - 1][0])
        assert np.array_equal(two.read_depth(), ref[n - 1][1])
        assert np.array_equal(two.read_shadow_map(), ref[n - 1][2])
    # every frame of a batch, delivered to device memory
    frames = torch.zeros((len(cams), h, w, 4), dtype=torch.uint8, device="cuda")
    two.render_batch(cams, lights[0], globs[0], abi.PASS_ALL, shadows_every_frame=True, device_frames=frames.data_ptr())
    two.sync()
    torch.cuda.synchronize()
    got = frames.cpu().numpy()
    chk = abi.Renderer(w, h, s)
    chk.upload_scene(small_scene)
    for k, c in enumerate(cams):
        chk.render_frame(c, lights[0], globs[0], abi.PASS_ALL)
        assert np.array_equal(chk.read_color(), got[k]), k
    chk.close()
    two.close()


@pytest.mark.parametrize("size", ["small", "full"])
def test_back_to_back_frames_on_one_set_of_buffers(small_scene, full_scene, size):
    """One frame in flight, frames submitted back to back without a host synchronisation: the
    library lets the next frame's vertex and triangle set-up run next to the previous frame's fog
    tail (its two streams exchange roles with every frame).  Every frame must still be the frame
    rendered alone: camera, light and time change from frame to frame, and so do the passes."""
    import torch
    sc, (w, h, s) = (small_scene, (400, 225, 256)) if size == "small" else (full_scene, (1920, 1080, 2048))
    ks = (0, 9, 21, 40, 63, 33, 50)
    cams = [default_inputs(w, h, pose=scene.flythrough_pose(k, 64))[0] for k in ks]
    lights = [abi.light_at_time(1.0 + 0.7 * k) for k in range(len(cams))]
    globs = [abi.global_uniforms(1.0 + 0.7 * k, 1) for k in range(len(cams))]
    passes = [abi.PASS_ALL, abi.PASS_ALL, abi.PASS_SHADOWS | abi.PASS_LIGHT_DEBUG, abi.PASS_ALL, abi.PASS_FOG, abi.PASS_ALL, abi.PASS_ALL]
    one = abi.Renderer(w, h, s)
    one.upload_scene(sc)
    ref = []
    for c, l, g, p in zip(cams, lights, globs, passes):
        one.render_frame(c, l, g, p)
        one.sync()
        ref.append((one.read_color(), one.read_depth(), one.read_shadow_map()))
    assert len({r[0].tobytes() for r in ref}) == len(ref)  # (the frames differ)
    r = abi.Renderer(w, h, s, frames_in_flight=1)
    r.upload_scene(sc)
    for n in (2, 3, 4, 5, 7):  # n frames back to back, then the last one is looked at: both stream roles end a run
        for c, l, g, p in list(zip(cams, lights, globs, passes))[:n]:
            r.render_frame(c, l, g, p)
        assert np.array_equal(r.read_color(), ref[n - 1][0]), n
        assert np.array_equal(r.read_depth(), ref[n - 1][1]), n
        assert np.array_equal(r.read_shadow_map(), ref[n - 1][2]), n
    # every frame of a batch, copied to device memory behind its frame
    frames = torch.zeros((len(cams), h, w, 4), dtype=torch.uint8, device="cuda")
    for rep in range(2):
        r.render_batch(cams, lights[0], globs[0], abi.PASS_ALL, shadows_every_frame=True, device_frames=frames.data_ptr())
    r.sync()
    torch.cuda.synchronize()
    got = frames.cpu().numpy()
    for k, c in enumerate(cams):
        one.render_frame(c, lights[0], globs[0], abi.PASS_ALL)
        assert np.array_equal(one.read_color(), got[k]), k
    one.close()
    r.close()


@pytest.mark.parametrize("seed,size", [(1, "small"), (2, "small"), (3, "small"), (4, "full")])
def test_random_call_sequences_on_one_set_of_buffers(small_scene, full_scene, seed, size):
    """One frame in flight: a random sequence of frames (random pose, time, passes), counted
    frames, batches, frames into caller memory through the entry point that allocates nothing,
    and read-backs at random places.  Whatever overlaps on the device, every frame that is looked
    at equals the same frame rendered by a context that waits after every call."""
    import random
    import torch
    rng = random.Random(seed)
    sc, (w, h, s) = (small_scene, (400, 225, 256)) if size == "small" else (full_scene, (1920, 1080, 2048))
    combos = [abi.PASS_ALL, abi.PASS_ALL, abi.PASS_ALL, abi.PASS_SHADOWS | abi.PASS_LIGHT_DEBUG, abi.PASS_FOG | abi.PASS_LIGHT_DEBUG,
              abi.PASS_SHADOWS | abi.PASS_FOG, abi.PASS_LIGHT_DEBUG]

    def inputs():
        k = rng.randrange(64)
        t = 1.0 + 0.1 * rng.randrange(60)
        return default_inputs(w, h, pose=scene.flythrough_pose(k, 64))[0], abi.light_at_time(t), abi.global_uniforms(t, 1), rng.choice(combos)

    ref = abi.Renderer(w, h, s)
    ref.upload_scene(sc)
    r = abi.Renderer(w, h, s, frames_in_flight=1)
    r.upload_scene(sc)
    # (a frame without the shadow passes samples cleared maps, whatever was drawn before; both
    # contexts start from a frame with every pass all the same, so that every buffer has been used)
    first = inputs()[:3] + (abi.PASS_ALL,)
    ref.render_frame(*first); ref.sync()
    r.render_frame(*first)
    target = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    checked = 0
    for step in range(60):
        op = rng.choice(["frame"] * 6 + ["counted", "batch", "static" if size == "small" else "frame", "read"])
        if op == "frame":
            args = inputs()
            ref.render_frame(*args); ref.sync()
            r.render_frame(*args)
        elif op == "counted":
            args = inputs()
            ref.set_counting(True); ref.render_frame(*args); a = ref.stats(); ref.set_counting(False)
            r.set_counting(True); r.render_frame(*args); b = r.stats(); r.set_counting(False)
            assert (a.fog_fragments, a.fog_noise_evals, a.shaded_fragments, a.replay_pixels) == \
                   (b.fog_fragments, b.fog_noise_evals, b.shaded_fragments, b.replay_pixels), step
        elif op == "batch":
            cams = [inputs()[0] for _ in range(3)]
            _, light, glob, passes = inputs()
            for c in cams:
                ref.render_frame(c, light, glob, passes); ref.sync()
            r.render_batch(cams, light, glob, passes, shadows_every_frame=True)
        elif op == "static":
            args = inputs()
            ref.render_frame(*args); ref.sync()
            r.render_frame_static(*args, target.data_ptr())
            r.sync(); torch.cuda.synchronize()
            assert np.array_equal(target.cpu().numpy(), ref.read_color()), step
            checked += 1
            # (the shadow maps of a static frame live in its own buffer: both contexts redraw theirs)
            again = inputs()[:3] + (abi.PASS_ALL,)
            ref.render_frame(*again); ref.sync()
            r.render_frame(*again)
        if op == "read" or step % 9 == 8:
            assert np.array_equal(r.read_color(), ref.read_color()), (step, op)
            assert np.array_equal(r.read_depth(), ref.read_depth()), (step, op)
            assert np.array_equal(r.read_shadow_map(), ref.read_shadow_map()), (step, op)
            checked += 1
    assert checked >= 6
    ref.close()
    r.close()


@pytest.mark.parametrize("in_flight", [1, 3])
def test_state_shared_between_frames_in_flight(small_scene, in_flight):
    """What frames share across the sets of frame buffers: a frame without shadow passes sees
    cleared maps whichever set it lands on, the counters of a counted frame are its own although
    other frames are still running, and an empty batch changes nothing."""
    w, h, s = 400, 225, 256
    cam, light, glob = default_inputs(w, h)
    no_shadow_pass = abi.PASS_FOG | abi.PASS_LIGHT_DEBUG
    fresh = render_gpu(small_scene, cam, light, glob, no_shadow_pass, w, h, s)
    full = render_gpu(small_scene, cam, light, glob, abi.PASS_ALL, w, h, s, counting=True)
    assert not np.array_equal(fresh["color"], full["color"])
    r = abi.Renderer(w, h, s, frames_in_flight=in_flight)
    r.upload_scene(small_scene)
    for _ in range(in_flight):  # every set has drawn shadow maps
        r.render_frame(cam, light, glob, abi.PASS_ALL)
    for _ in range(in_flight + 1):  # a frame without the passes, on every set
        r.render_frame(cam, light, glob, no_shadow_pass)
        assert np.array_equal(r.read_color(), fresh["color"])
    assert (r.read_shadow_map() == 1.0).all()
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    assert np.array_equal(r.read_color(), full["color"])
    # counted frame behind uncounted frames still in flight
    other = default_inputs(w, h, pose=scene.flythrough_pose(5, 64))[0]
    for _ in range(4):
        r.render_frame(other, light, glob, abi.PASS_ALL)
    r.set_counting(True)
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    st = r.stats()
    r.set_counting(False)
    assert dict(fog_fragments=int(st.fog_fragments), fog_noise_evals=int(st.fog_noise_evals), replay_pixels=int(st.replay_pixels),
                replay_slow_pixels=int(st.replay_slow_pixels)) == full["counters"]
    assert full["counters"]["fog_noise_evals"] > 0
    # an empty batch
    before = r.read_color()
    r.render_batch([], light, glob, abi.PASS_ALL)
    assert np.array_equal(r.read_color(), before) and np.array_equal(before, full["color"])
    r.close()


def test_scene_uploaded_through_the_native_glb_importer(tmp_path):
    """wr_upload_glb (C++ importer) and the Python loader + wr_upload_* give the same frame."""
    from test_abi_and_host import _synthetic_glb
    from wgpu_renderer_amd import gltf_loader
    w, h, s = 200, 120, 64
    glb, _ = _synthetic_glb(np.random.default_rng(5), True)
    p = tmp_path / "s.glb"
    p.write_bytes(glb)
    model, textures = gltf_loader.load_glb(str(p))
    cube = scene.cube_model()
    inst = abi.instance_raw((0.0, 0.0, -6.0), (1, 0, 0, 0), (2.0, 2.0, 2.0))
    fog_inst = abi.instance_raw((0.0, -50.0, 0.0), (1, 0, 0, 0), (10.0, 10.0, 10.0))
    cam = abi.camera_uniform((0.0, 0.0, 0.0), 0.0, -90.0, 55.0, w, h)
    light = abi.light_uniform((1.0, 3.0, 0.0), (1.0, 0.9, 0.8, 50.0))
    glob = abi.global_uniforms(1.5, 1)
    sc = scene.Scene(textures, model, cube, scene.cube_model(), inst, fog_inst)
    ref = render_gpu(sc, cam, light, glob, abi.PASS_ALL, w, h, s)
    assert (ref["vis"] != 0xFFFFFFFF).mean() > 0.02
    o = render_oracle(sc, cam, light, glob, abi.PASS_ALL, w, h, s, threads=2)
    assert_parity(ref, o)
    r = abi.Renderer(w, h, s)
    g = abi.Glb(data=glb)
    r.upload_glb(abi.MODEL_GEOMETRY, g)
    g.close()
    r.set_instance(abi.MODEL_GEOMETRY, inst)
    for mesh in cube.meshes:
        r.upload_mesh(abi.MODEL_FOG_VOLUME, mesh.vertices, mesh.indices, mesh.material)
        r.upload_mesh(abi.MODEL_LIGHT_MARKER, mesh.vertices, mesh.indices, mesh.material)
    r.set_instance(abi.MODEL_FOG_VOLUME, fog_inst)
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    assert np.array_equal(r.read_color(), ref["color"]) and np.array_equal(r.read_visibility(), ref["vis"])
    r.close()


def test_draw_orders_give_different_images():
    w, h, s = 160, 120, 64
    cam = abi.camera_uniform((0.0, 0.0, 500.0), 0.0, -90.0, 55.0, w, h)
    light = abi.light_uniform((0.0, 100.0, 300.0), (1.0, 1.0, 1.0, 250000.0))
    glob = abi.global_uniforms(1.0, 0)
    a = render_gpu(blend_order_scene((0, 1, 3, 2)), cam, light, glob, abi.PASS_LIGHT_DEBUG, w, h, s)["color"]
    b = render_gpu(blend_order_scene((2, 3, 1, 0)), cam, light, glob, abi.PASS_LIGHT_DEBUG, w, h, s)["color"]
    assert (a != b).any()


def test_degenerate_and_extreme_triangles():
    """Zero-area, duplicated-vertex, sliver, screen-filling and far-outside triangles, a
    triangle through the camera, and NaN tangents (a GLB without TANGENT, mesh.rs:67-69)."""
    w, h, s = 200, 150, 64
    v = np.zeros((0, 14), np.float32)
    tris = [
        [(0, 0, -100), (0, 0, -100), (0, 0, -100)],                  # point
        [(-50, 0, -100), (50, 0, -100), (0, 0, -100)],               # collinear
        [(-1e5, -1e5, -300), (1e5, -1e5, -300), (0, 1e5, -300)],     # covers everything, guard-band clipped
        [(-40, -40, -200), (40, -40, -200), (0, 1e-4, -200)],        # sliver
        [(-30, -30, 50), (30, -30, -400), (0, 40, -400)],            # crosses the camera plane
        [(1e4, 1e4, -50), (1.1e4, 1e4, -50), (1e4, 1.1e4, -50)],     # far outside the frustum
        [(-20, -20, -150), (20, -20, -150), (0, 20, -150)],          # ordinary, NaN tangents below
        [(-20, -20, -2990), (20, -20, -3010), (0, 20, -2990)],       # crosses the far plane
    ]
    verts, idx = [], []
    for t in tris:
        for p in t:
            row = np.zeros(14, np.float32)
            row[0:3] = p
            row[3:5] = (p[0] * 0.01, p[1] * 0.01)
            row[5:8] = (0, 0, 1)
            row[8:11] = (1, 0, 0)
            row[11:14] = (0, 1, 0)
            verts.append(row)
        idx += [len(verts) - 3, len(verts) - 2, len(verts) - 1]
    v = np.array(verts, np.float32)
    v[18:21, 8:14] = np.nan
    sc0 = blend_order_scene((0,))
    sc = scene.Scene(sc0.textures, scene.Model([scene.Mesh(v, np.array(idx, np.uint32), 0)], sc0.geometry.materials),
                     scene.cube_model(), scene.cube_model(), abi.instance_raw((0.0, 0.0, 0.0)),
                     abi.instance_raw((0.0, 0.0, -200.0), (1, 0, 0, 0), (60.0, 20.0, 60.0)))
    cam = abi.camera_uniform((0.0, 0.0, 0.0), 0.0, -90.0, 55.0, w, h)
    light = abi.light_uniform((10.0, 40.0, -100.0), (1.0, 0.9, 0.8, 250000.0))
    glob = abi.global_uniforms(3.0, 1)
    g = render_gpu(sc, cam, light, glob, abi.PASS_ALL, w, h, s)
    o = render_oracle(sc, cam, light, glob, abi.PASS_ALL, w, h, s, threads=2)
    assert_parity(g, o)
    assert (o["vis"] != 0xFFFFFFFF).mean() > 0.9  # the huge triangle was clipped, not dropped


def test_bgra_target_is_the_rgba_frame_with_red_and_blue_exchanged(tiny_scene):
    # state.rs:106 takes the surface's first format, Bgra8UnormSrgb on most Vulkan surfaces
    w, h, s = 333, 207, 128
    cam, light, glob = default_inputs(w, h)
    rgba = render_gpu(tiny_scene, cam, light, glob, abi.PASS_ALL, w, h, s)["color"]
    for in_flight in (1, 2):
        r = abi.Renderer(w, h, s, color_format=abi.FORMAT_BGRA8_UNORM_SRGB, frames_in_flight=in_flight)
        r.upload_scene(tiny_scene)
        for _ in range(3):  # the same buffers frame after frame: the exchange happens once per frame
            r.render_frame(cam, light, glob, abi.PASS_ALL)
        bgra = r.read_color()
        r.close()
        assert np.array_equal(bgra[..., [2, 1, 0, 3]], rgba) and (rgba[..., 0] != rgba[..., 2]).any()
    with pytest.raises(abi.WrError):
        abi.Renderer(w, h, s, color_format=7)


def test_resize_and_scene_reuse(tiny_scene):
    cam, light, glob = default_inputs(200, 120)
    r = abi.Renderer(200, 120, 128)
    r.upload_scene(tiny_scene)
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    a = r.read_color()
    r.resize(0, 0)  # ignored like state.rs:576
    r.resize(320, 200)
    cam2, _, _ = default_inputs(320, 200)
    r.render_frame(cam2, light, glob, abi.PASS_ALL)
    b = r.read_color()
    r.resize(200, 120)
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    assert np.array_equal(r.read_color(), a) and b.shape == (200, 320, 4)
    o = render_oracle(tiny_scene, cam2, light, glob, abi.PASS_ALL, 320, 200, 128, threads=4)
    assert np.array_equal(b, o["color"])
    r.close()


@pytest.mark.parametrize("in_flight", [1, 2])
def test_profiling_levels(small_scene, in_flight):
    """Level 1 times every stage, level 2 only the dominant kernel, the view noise or the march (what
    a timed run can afford); neither changes a pixel."""
    w, h, s = 400, 225, 256
    cam, light, glob = default_inputs(w, h)
    plain = render_gpu(small_scene, cam, light, glob, abi.PASS_ALL, w, h, s)["color"]
    r = abi.Renderer(w, h, s, frames_in_flight=in_flight)
    r.upload_scene(small_scene)
    r.set_profiling(1)
    for _ in range(5):
        r.render_frame(cam, light, glob, abi.PASS_ALL)
    st = r.stats()
    assert st.profiled_frames == 5
    assert min(st.ms_shadow, st.ms_geometry, st.ms_shade, st.ms_fog, st.ms_fog_march) > 0.0
    assert st.ms_total >= max(st.ms_shadow, st.ms_geometry, st.ms_fog_march)
    if os.environ.get("WR_DEBUG_SCHEDULE"):  # (a forced fog schedule: either of the two below)
        pass
    elif in_flight == 1:  # the split march has a light-noise kernel of its own
        assert st.ms_fog_light_march > 0.2 * st.ms_fog_march
    else:
        assert st.ms_fog_light_march < 0.2 * st.ms_fog_march
    r.set_profiling(2)
    for _ in range(8):
        r.render_frame(cam, light, glob, abi.PASS_ALL)
    st = r.stats()
    assert st.profiled_frames == (2 if in_flight == 1 else 8) and st.ms_fog_march > 0.0  # (one in flight: every fourth frame)
    assert st.ms_shadow == st.ms_geometry == st.ms_shade == st.ms_fog == st.ms_total == st.ms_fog_light_march == 0.0
    assert np.array_equal(r.read_color(), plain)
    r.set_profiling(0)
    r.close()


def test_list_overflow_is_reported_with_its_frame(tiny_scene, monkeypatch):
    """A frame whose bin item list overflows is reported at the next wr_sync, by its number in
    submission order (each set of frame buffers has its own error word), and the context goes
    on working."""
    w, h, s = 256, 144, 128
    cam, light, glob = default_inputs(w, h)
    good = render_gpu(tiny_scene, cam, light, glob, abi.PASS_ALL, w, h, s)["color"]
    monkeypatch.setenv("WR_DEBUG_BIN_ITEMS_CAP", "100")
    r = abi.Renderer(w, h, s, frames_in_flight=2)
    r.upload_scene(tiny_scene)
    for _ in range(3):
        r.render_frame(cam, light, glob, abi.PASS_ALL)
    with pytest.raises(abi.WrError) as e:
        r.sync()
    assert e.value.code == abi.WR_ERR_CAPACITY and "bin item capacity exceeded by frame 2 " in str(e.value), str(e.value)
    r.sync()  # reported once
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    with pytest.raises(abi.WrError) as e:
        r.read_color()
    assert "by frame 4 " in str(e.value), str(e.value)
    r.close()
    monkeypatch.delenv("WR_DEBUG_BIN_ITEMS_CAP")
    assert np.array_equal(render_gpu(tiny_scene, cam, light, glob, abi.PASS_ALL, w, h, s)["color"], good)


def test_error_paths():
    r = abi.Renderer(64, 64, 64)
    cam, light, glob = default_inputs(64, 64)
    with pytest.raises(abi.WrError) as e:
        r.render_frame(cam, light, glob, abi.PASS_ALL)
    assert e.value.code == abi.WR_ERR_NO_SCENE
    with pytest.raises(abi.WrError):
        r.upload_material(0, 5, 5, 5, 1.0, 1.0)  # unknown textures
    tex = np.zeros((4, 4, 4), np.uint8)
    r.upload_texture(tex, True)
    r.upload_material(0, 0, 0, 0, 1.0, 1.0)
    v, i = _quad(-1, 1, -1, 1, 0)
    with pytest.raises(abi.WrError):
        r.upload_mesh(0, v, np.array([0, 1, 9], np.uint32), 0)  # index out of range
    with pytest.raises(abi.WrError):
        r.upload_mesh(0, v, i, 3)  # unknown material
    r.close()
    with pytest.raises(abi.WrError):
        abi.Renderer(64, 64, 64, tile_rank=2, tile_world=2)
    with pytest.raises(abi.WrError):
        abi.Renderer(0, 64, 64)


# ------------------------------------------------------------------ full size
@pytest.fixture(scope="module")
def full_scene():
    return scene.sponza_standin(detail=1.0, tex_size=1024)  # the scene bench.py times


def test_full_size_frame_against_oracle(full_scene):
    """BASELINE configs[2] at its real size: 1920x1080, six 2048^2 shadow faces, fog."""
    w, h, s = 1920, 1080, 2048
    cam, light, glob = default_inputs(w, h)
    g = render_gpu(full_scene, cam, light, glob, abi.PASS_ALL, w, h, s, counting=True)
    o = render_oracle(full_scene, cam, light, glob, abi.PASS_ALL, w, h, s, threads=16)
    assert_parity(g, o)
    assert g["counters"]["fog_noise_evals"] == live_evals(o)


def test_full_size_properties(full_scene):
    """Size-independent properties at 1920x1080: determinism, pose batch == single frames,
    shadow reuse == recomputation, bin-sharded rendering == whole-frame rendering."""
    import torch
    w, h, s = 1920, 1080, 2048
    cam, light, glob = default_inputs(w, h)
    r = abi.Renderer(w, h, s)
    r.upload_scene(full_scene)
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    ref = r.read_color()
    ref_vis = r.read_visibility()
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    assert np.array_equal(r.read_color(), ref) and np.array_equal(r.read_visibility(), ref_vis)
    # batch of three poses, shadows computed once, frames delivered to device memory
    cams = [cam, default_inputs(w, h, pose=scene.flythrough_pose(40))[0], default_inputs(w, h, pose=scene.flythrough_pose(300))[0]]
    frames = torch.zeros((3, h, w, 4), dtype=torch.uint8, device="cuda")
    r.render_batch(cams, light, glob, abi.PASS_ALL, shadows_every_frame=False, device_frames=frames.data_ptr())
    r.sync()
    torch.cuda.synchronize()
    fr = frames.cpu().numpy()
    assert np.array_equal(fr[0], ref)
    for k in (1, 2):
        r.render_frame(cams[k], light, glob, abi.PASS_ALL)
        assert np.array_equal(r.read_color(), fr[k])
    r.close()
    # screen-bin sharding over 3 contexts, gathered through packed bin buffers
    world = 3
    nb = ((w + 63) // 64) * ((h + 63) // 64)
    per = (nb + world - 1) // world
    gathered = abi.Renderer(w, h, s)
    for rank in range(world):
        rr = abi.Renderer(w, h, s, tile_rank=rank, tile_world=world)
        rr.upload_scene(full_scene)
        rr.render_frame(cam, light, glob, abi.PASS_ALL)
        packed = torch.zeros(per * 64 * 64, dtype=torch.int32, device="cuda")
        n = rr.pack_owned_bins(packed.data_ptr())
        rr.sync()
        assert n == len(range(rank, nb, world))
        gathered.scatter_bins(packed.data_ptr(), rank, world)
        gathered.sync()
        rr.close()
    assert np.array_equal(gathered.read_color(), ref)
    gathered.close()


def test_512_pose_flythrough_batch_at_full_size(full_scene):
    """BASELINE configs[4] at its size: the 512-pose flythrough at 1920x1080 with shadows and fog
    through wr_render_batch into 4.25 GB of device frames, with the light fixed (shadow maps
    rendered once, then once per pose) and with t = 2 + k/60 (light, fog drift and shadow maps
    change every frame, frames rotating through four sets of buffers).  Sampled frames equal
    single wr_render_frame calls bit for bit, one of them is checked against the oracle."""
    import torch
    w, h, s, n = 1920, 1080, 2048, 512
    poses = [scene.flythrough_pose(k, n) for k in range(n)]
    cams = [default_inputs(w, h, pose=p)[0] for p in poses]
    light, glob = abi.light_at_time(2.0), abi.global_uniforms(2.0, 1)
    sample = [0, 1, 63, 64, 200, 255, 300, 511]
    one = abi.Renderer(w, h, s)
    one.upload_scene(full_scene)
    ref = {}
    for k in sample:
        one.render_frame(cams[k], light, glob, abi.PASS_ALL)
        ref[k] = one.read_color()
    o = render_oracle(full_scene, cams[300], light, glob, abi.PASS_ALL, w, h, s, threads=16)
    assert np.array_equal(ref[300], o["color"])
    assert len({ref[k].tobytes() for k in sample}) == len(sample)  # the poses really differ
    r = abi.Renderer(w, h, s, frames_in_flight=4)
    r.upload_scene(full_scene)
    frames = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda")
    for every in (False, True):
        frames.zero_()
        r.render_batch(cams, light, glob, abi.PASS_ALL, shadows_every_frame=every, device_frames=frames.data_ptr())
        r.sync()
        torch.cuda.synchronize()
        for k in sample:
            assert np.array_equal(frames[k].cpu().numpy(), ref[k]), (every, k)
        assert int((frames.view(n, -1)[:, 3::4] != 0).all(dim=1).sum()) == n  # every frame was delivered (alpha is never 0)
    del frames
    # animated: one wr_render_frame per pose, four frames in flight
    for k in range(n):
        t = 2.0 + k / 60.0
        r.render_frame(cams[k], abi.light_at_time(t), abi.global_uniforms(t, 1), abi.PASS_ALL)
        if k in sample:
            got = r.read_color()
            one.render_frame(cams[k], abi.light_at_time(t), abi.global_uniforms(t, 1), abi.PASS_ALL)
            assert np.array_equal(got, one.read_color()), k
    r.close()
    one.close()


def test_4k_bin_sharding_over_eight_contexts(full_scene):
    """BASELINE configs[3]: 3840x2160 with the 64x64 bins interleaved over eight ranks; the
    gathered frame equals the frame one context renders alone."""
    import torch
    w, h, s = 3840, 2160, 2048
    cam, light, glob = default_inputs(w, h)
    whole = abi.Renderer(w, h, s)
    whole.upload_scene(full_scene)
    whole.render_frame(cam, light, glob, abi.PASS_ALL)
    ref = whole.read_color()
    ref_depth = whole.read_depth()
    assert (ref[..., :3].max(axis=2) > 0).mean() > 0.5  # a real frame, not the clear colour
    world = 8
    nb = ((w + 63) // 64) * ((h + 63) // 64)
    per = (nb + world - 1) // world
    for rank in range(world):
        rr = abi.Renderer(w, h, s, tile_rank=rank, tile_world=world)
        rr.upload_scene(full_scene)
        rr.render_frame(cam, light, glob, abi.PASS_ALL)
        packed = torch.zeros(per * 64 * 64, dtype=torch.int32, device="cuda")
        n = rr.pack_owned_bins(packed.data_ptr())
        rr.sync()
        assert n == len(range(rank, nb, world))
        d = rr.read_depth()
        rr.close()
        # the rank's own bins carry the same depth as the whole frame
        by, bx = np.divmod(np.arange(rank, nb, world), (w + 63) // 64)
        for k in (0, len(by) // 2, len(by) - 1):
            ys, xs = slice(by[k] * 64, min(by[k] * 64 + 64, h)), slice(bx[k] * 64, min(bx[k] * 64 + 64, w))
            assert np.array_equal(d[ys, xs], ref_depth[ys, xs])
        whole.scatter_bins(packed.data_ptr(), rank, world)  # overwrite the whole frame bin by bin
        whole.sync()
    assert np.array_equal(whole.read_color(), ref)
    whole.close()


def test_gather_bins_on_the_device_with_a_stand_in_transport(small_scene):
    """wr_gather_bins_via, the code wr_gather_bins runs over RCCL, with three bin-sharded contexts
    on this one device and a transport that copies device memory between them: the root ends up
    with the frame one context renders alone.  (RCCL itself needs one device per rank; the box
    has one.)"""
    import ctypes as C
    import torch
    w, h, s = 1000, 520, 256
    cam, light, glob = default_inputs(w, h)
    whole = abi.Renderer(w, h, s)
    whole.upload_scene(small_scene)
    whole.render_frame(cam, light, glob, abi.PASS_ALL)
    ref = whole.read_color()
    whole.close()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    world, root = 3, 1
    ranks = [abi.Renderer(w, h, s, tile_rank=r, tile_world=world) for r in range(world)]
    sent, log = {}, []

    def transport(rank):
        def send(user, buf, nbytes, peer, stream):
            sent[(rank, peer)] = (buf, nbytes)
            log.append(("send", rank, peer, nbytes))
            return 0

        def recv(user, buf, nbytes, peer, stream):
            src, n = sent[(peer, rank)]
            log.append(("recv", rank, peer, nbytes))
            if n != nbytes:
                return 1
            torch.cuda.synchronize()  # the sender's pack kernel ran on its own stream
            return hip.hipMemcpy(buf, src, nbytes, 3)  # device to device

        t = abi.Transport(None, abi.Transport.GROUP(lambda u: 0), abi.Transport.SEND(send), abi.Transport.RECV(recv),
                          abi.Transport.GROUP(lambda u: 0))
        return t

    keep = []
    for r in ranks:
        r.upload_scene(small_scene)
        r.render_frame(cam, light, glob, abi.PASS_ALL)
    for k in [r for r in range(world) if r != root] + [root]:  # senders first: the stand-in has no queue of its own
        t = transport(k)
        keep.append(t)
        ranks[k].gather_bins_via(t, root)
        ranks[k].sync()
    assert np.array_equal(ranks[root].read_color(), ref)
    nb = ((w + 63) // 64) * ((h + 63) // 64)
    assert sorted(x[3] for x in log if x[0] == "send") == sorted(len(range(r, nb, world)) * 64 * 64 * 4 for r in range(world) if r != root)
    # a non-root rank still holds only its own bins
    other = ranks[0].read_color()
    assert not np.array_equal(other, ref)
    with pytest.raises(abi.WrError):
        ranks[0].gather_bins_via(keep[0], 5)  # root outside the world
    for r in ranks:
        r.close()
    one = abi.Renderer(64, 64, 64)
    with pytest.raises(abi.WrError):
        one.gather_bins_via(keep[0], 0)  # not a bin-sharded context
    one.close()


def test_largest_target(tiny_scene):
    """7168 x 7168 is the largest target wr_create accepts (16-bit pixel coordinates in the
    set-up records, guard band of 8w); one more pixel is an argument error."""
    w = h = 7168
    cam, light, glob = default_inputs(w, h)
    r = abi.Renderer(w, h, 64)
    r.upload_scene(tiny_scene)
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    vis = r.read_visibility()
    col = r.read_color()
    r.close()
    assert vis.shape == (h, w) and (vis != 0xFFFFFFFF).mean() > 0.3
    # the same view at 1/8 size covers the same share of the target
    small = render_gpu(tiny_scene, *default_inputs(896, 896), abi.PASS_ALL, 896, 896, 64)
    assert abs((vis != 0xFFFFFFFF).mean() - (small["vis"] != 0xFFFFFFFF).mean()) < 0.01
    assert col.shape == (h, w, 4) and (col[..., :3].max(axis=2) > 0).mean() > 0.3
    with pytest.raises(abi.WrError):
        abi.Renderer(7169, 64, 64)


# ------------------------------------------------------------------ device arithmetic units
def _hostcheck():
    import ctypes as C
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck", "libwr_hostcheck.so")
    H = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    H.hc_dev_sin.argtypes = [fp, C.c_int, fp]
    H.hc_dev_hash.argtypes = [fp, C.c_int, fp, fp]
    H.hc_dev_fract_mismatches.restype = C.c_longlong
    return H, fp


def _hash_arguments():
    """Arguments the noise hash sees: dot products of integer lattice points (noise.wgsl:4), a
    dense sweep, values around multiples of pi, the range limit, zeros, NaN and infinities.  No
    non-zero value below 2^-20 in magnitude: a lattice dot product cannot be one."""
    rng = np.random.default_rng(11)
    f32 = np.float32
    u = np.floor(rng.uniform(-17000, 17000, (400000, 3))).astype(f32)
    u[:50000] = np.floor(rng.uniform(-40, 40, (50000, 3))).astype(f32)
    dots = (u[:, 0] * f32(12.345) + u[:, 1] * f32(67.89)) + u[:, 2] * f32(412.12)
    k = np.arange(1, 200001, dtype=np.float64) * np.pi
    near_pi = np.concatenate([k.astype(f32), np.nextafter(k.astype(f32), f32(np.inf)), -k.astype(f32)])
    special = np.array([0.0, -0.0, 2.0 ** -20, -(2.0 ** -20), 8388607.5, 8388608.0, -8388608.0, 8388609.0, 1e9, -1e9,
                        1073741760.0, 1073741824.0, -1073741824.0, 1.2e9, -3e38, np.inf, -np.inf, np.nan], dtype=f32)
    sweep = np.concatenate([rng.uniform(-8.4e6, 8.4e6, 300000), rng.uniform(-1.1e9, 1.1e9, 100000)]).astype(f32)
    small = rng.uniform(-20, 20, 100000).astype(f32)
    return np.ascontiguousarray(np.concatenate([dots, near_pi, special, sweep, small]).astype(f32))


def test_device_sine_and_hash_equal_the_oracle_bit_for_bit():
    import ctypes as C
    from oracle_lib import lib as oracle_lib_handle
    H, fp = _hostcheck()
    L = oracle_lib_handle()
    x = _hash_arguments()
    n = len(x)
    s_dev = np.empty(n, np.float32)
    hg = np.empty(n, np.float32)
    hu = np.empty(n, np.float32)
    assert H.hc_dev_sin(x.ctypes.data_as(fp), n, s_dev.ctypes.data_as(fp)) == 0
    assert H.hc_dev_hash(x.ctypes.data_as(fp), n, hg.ctypes.data_as(fp), hu.ctypes.data_as(fp)) == 0
    L.wro_sin_many.argtypes = [fp, C.c_int, fp]
    L.wro_hash_many.argtypes = [fp, C.c_int, fp]
    s_or = np.empty(n, np.float32)
    h_or = np.empty(n, np.float32)
    L.wro_sin_many(x.ctypes.data_as(fp), n, s_or.ctypes.data_as(fp))
    L.wro_hash_many(x.ctypes.data_as(fp), n, h_or.ctypes.data_as(fp))
    assert np.array_equal(s_dev.view(np.uint32), s_or.view(np.uint32))
    assert np.array_equal(hg.view(np.uint32), h_or.view(np.uint32))
    # the unguarded form is only ever called below the range limit, where it must agree
    inr = np.abs(x) < np.float32(1073741824.0)
    assert np.array_equal(hu.view(np.uint32)[inr], h_or.view(np.uint32)[inr])
    # accuracy of the shared sine against double precision over the range the noise reaches
    ok = (np.abs(x) < np.float32(8388608.0)) & np.isfinite(x)
    assert np.abs(s_dev[ok].astype(np.float64) - np.sin(x[ok].astype(np.float64))).max() < 1.8e-7
    assert np.all((h_or[np.isfinite(h_or)] >= -1.0) & (h_or[np.isfinite(h_or)] <= 1.0))


def test_device_fraction_instruction_equals_x_minus_floor():
    """The hash takes its fractional part with the hardware's fraction instruction; over all
    2^32 f32 values it equals x - floor(x) wherever that is below 1 (the hash never sees the
    other case: checked exhaustively on the CPU for every argument of at least 2^-20 in
    magnitude, the smallest non-zero lattice dot product)."""
    H, _ = _hostcheck()
    assert H.hc_dev_fract_mismatches() == 0
