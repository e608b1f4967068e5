"""The ordering promises of WrConfig.stream (include/wr_abi.h) with a stream the caller owns,
and the RCCL shims of the library (wr_comm_*, the grouped send / recv of wr_gather_bins) on a
real communicator.  The reference orders everything through one wgpu queue
(src/core/state.rs:674,761,803: three submits per frame on `self.queue`); the caller's HIP
stream stands for that queue."""
import ctypes as C

import numpy as np
import pytest

import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import abi, scene

pytestmark = pytest.mark.gpu


def _inputs(w, h, k):
    rc = scene.REFERENCE_CAMERA
    pos, pitch, yaw = scene.flythrough_pose(7 * k + 3, 64)
    t = 1.0 + 0.3 * k
    return abi.camera_uniform(pos, pitch, yaw, rc["fovy"], w, h), abi.light_at_time(t), abi.global_uniforms(t, 1), abi.PASS_ALL


class _DeviceView:
    """Device memory the library owns, as a tensor (no copy): the colour target behind
    wr_color_device_ptr."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "|u1", "data": (ptr, False), "version": 2}


def _busy(torch, n=6):
    """A few milliseconds of work on the current stream."""
    a = torch.ones((2048, 2048), device="cuda")
    for _ in range(n):
        a = (a @ a) * 1e-4
    return a


@pytest.mark.parametrize("in_flight", [1, 3])
def test_frames_ordered_against_a_stream_of_the_caller(small_scene, in_flight):
    """Work queued on the caller's stream before a frame runs before it, work queued after a frame
    (one frame in flight) sees the finished frame, with no host wait anywhere in between; the
    frames equal those of a context on a library-owned stream that waits after every call."""
    import torch
    w, h, s = 400, 225, 256
    ref = abi.Renderer(w, h, s)
    ref.upload_scene(small_scene)
    st = torch.cuda.Stream()
    r = abi.Renderer(w, h, s, stream=st.cuda_stream, frames_in_flight=in_flight)
    r.upload_scene(small_scene)
    r.render_frame(*_inputs(w, h, 0))
    r.sync()
    want = []
    for k in range(1, 7):
        ref.render_frame(*_inputs(w, h, k))
        want.append(ref.read_color())
    snaps, statics = [], []
    with torch.cuda.stream(st):
        for k in range(1, 7):
            keep = _busy(torch)  # the frame is queued behind this
            r.render_frame(*_inputs(w, h, k))
            if in_flight == 1:
                # the frame is complete, as seen from the stream, before this copy
                view = torch.as_tensor(_DeviceView(r.color_device_ptr(), (h, w, 4)), device="cuda")
                snaps.append(view.clone())
            # a static frame into the caller's memory: the fill queued before it must not land on
            # top of it, the copy queued after it must see it
            out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
            keep = keep + _busy(torch, 2)
            out.fill_(55)
            r.render_frame_static(*_inputs(w, h, k), out.data_ptr())
            statics.append(out.clone())
            del keep
    if in_flight > 1:
        r.sync()
        assert np.array_equal(r.read_color(), want[-1])
    st.synchronize()
    for k, snap in enumerate(snaps):
        assert np.array_equal(snap.cpu().numpy(), want[k]), k
    for k, out in enumerate(statics):
        assert np.array_equal(out.cpu().numpy(), want[k]), k
    assert np.array_equal(r.read_color(), want[-1])
    ref.close()
    r.close()


def test_rccl_shims_on_a_communicator_of_one_rank(small_scene):
    """wr_comm_unique_id -> wr_comm_create (world 1) -> the grouped send / recv wr_gather_bins is
    made of (wr_comm_transport), rank 0 to itself, on the packed bins of a rendered frame ->
    wr_comm_destroy.  One GPU has no peer, so the gather itself cannot run here; every RCCL entry
    point it binds does."""
    import torch  # first: the process then holds one HIP runtime, the one RCCL was built against
    w, h, s = 400, 225, 128
    r = abi.Renderer(w, h, s)
    r.upload_scene(small_scene)
    r.render_frame(*_inputs(w, h, 2))
    owned, slot = abi.gather_layout(w, h, 1, 0)
    assert owned == slot == ((w + 63) // 64) * ((h + 63) // 64)
    nbytes = owned * 64 * 64 * 4
    src = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.full((nbytes,), 7, dtype=torch.uint8, device="cuda")
    assert r.pack_owned_bins(src.data_ptr()) == owned
    r.sync()
    torch.cuda.synchronize()
    assert int(src.count_nonzero()) > nbytes // 8

    path = abi.comm_library_path()
    assert "rccl" in path
    print("RCCL bound:", path)
    uid = abi.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = abi.comm_create(uid, 0, 1, 0)
    assert comm
    tp = abi.comm_transport(comm)
    st = torch.cuda.Stream()
    stream = C.c_void_p(st.cuda_stream)
    assert tp.group_start(tp.user) == 0
    assert tp.send(tp.user, src.data_ptr(), nbytes, 0, stream) == 0
    assert tp.recv(tp.user, dst.data_ptr(), nbytes, 0, stream) == 0
    assert tp.group_end(tp.user) == 0
    st.synchronize()
    assert torch.equal(src, dst)
    # and the bins that travelled are the frame: scattered into a second context's target
    r2 = abi.Renderer(w, h, s)
    r2.upload_scene(small_scene)
    r2.render_frame(*_inputs(w, h, 5))
    r2.scatter_bins(dst.data_ptr(), 0, 1)
    assert np.array_equal(r2.read_color(), r.read_color())
    abi.comm_destroy(comm)
    r.close()
    r2.close()
