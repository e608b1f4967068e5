#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one full frame of BASELINE.json configs[2]: the Sponza-class scene at 1920x1080
with the six shadow passes, PBS shading with PCF, the light marker and the fog pass, every
stage recomputed each frame as State::render does (state.rs:630-809).  With N > 1 every rank
(one per GPU) renders its own frames: frames shard across GPUs with no data-path collective
("scaling": "weak"); `value` is frames of all ranks divided by the slowest rank's time.

N > 1 runs either under torch.distributed.run (RANK / WORLD_SIZE in the environment) or
straight from `python bench.py --gpus N`: without WORLD_SIZE the script starts
`python -m torch.distributed.run --nproc-per-node N` on itself as a child process, before
anything touches a device, relays its output and exits with its status.  The ranks talk over
gloo (barriers, the maximum of the ranks' times, the 128 bytes of the communicator id), so
the only RCCL that is live in the process is the one libwr_hip.so loads for its bin gather.

After the timed region of the default workload the same run measures the two workloads
BASELINE.json's configs[3] and configs[4] name and prints them as `scale_workloads`:
`posebatch` renders the 512-pose flythrough, split contiguously over the ranks, through
wr_render_batch (total work fixed: strong scaling; shadow maps of the fixed light shared by
sixteen poses), `posebatch_animated` the same poses with t = 2 + k/60 (every pass per frame),
`tiles4k` one 3840x2160 frame whose screen bins are interleaved over the ranks and gathered
on rank 0 by wr_gather_bins (grouped RCCL send / recv queued behind the frame, no host
synchronisation).  --workload posebatch / tiles4k makes one of them the line itself.

res/models/Sponza.glb is not part of the reference checkout, so the scene is the procedural
stand-in of wgpu_renderer_amd.scene (about 262k triangles, 25 materials, 1024^2 textures)
unless --sponza PATH names a real file.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Every frame in flight runs on its own pair of HIP streams; with HIP's default of four hardware
# queues per process several of those streams share a queue and run one after the other.  Read
# by the HIP runtime when it initialises, so it is set before anything touches the device.
if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"

import numpy as np  # noqa: E402

# Algorithmic f32 operations of one fog_noise evaluation inside ray_march (fog.wgsl:64-85 +
# 48-62, noise.wgsl), a fused multiply-add counted as 2.  Sine (Cephes scheme, one of its two
# polynomials per call): octant 4 (multiply, truncate, round up to even, convert) + reduction 6
# + z 1 + polynomial 8 (sine 7, cosine 9) = 19; rand = dot 5 + sine 19 + hash 5 = 29 (+3 for
# the corner offset on 7 of 8 corners); value_noise = floor/fract 6 + smoothstep 27 + hashes
# 29 + 7*32 + 7 mixes * 4 = 314; fbm = 8 * (314 + 6) + 2 = 2562; fog_noise = 2 * 2562 + 62 =
# 5186; march bookkeeping 12.
FLOP_PER_FOG_NOISE_EVAL = 5198
PEAK_FP32_TFLOPS = 157.3   # MI355X f32 vector peak = dense f32 matrix peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
# VALU issue peak: one wave64 instruction per 2 cycles and SIMD (MI355X_MICROARCH.md, v_fma_f32
# row), 1024 SIMDs, 2.4 GHz.  (scripts/dev_ifetch_bench.hip measures 1.04 per ns and SIMD.)
VALU_ISSUE_PEAK_PER_NS = 1024 * 2.4 / 2.0
PROFILE_JSON = os.path.join(ROOT, "profiles", "r03_counters.json")


def source_hash():
    """Hash of the sources libwr_hip.so is built from: ties a counter profile to a build."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "wgpu-renderer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def load_profile(schedule):
    """Per-kernel counters of profiles/r03_counters.json for the fog schedule in use ("1": one
    frame in flight, "4": frames overlapping), or (None, reason) if the file is missing or was
    taken on other sources."""
    if not os.path.exists(PROFILE_JSON):
        return None, "no counter profile committed"
    prof = json.load(open(PROFILE_JSON))
    if prof.get("source_hash") != source_hash():
        return None, f"counter profile {prof.get('source_hash')} is of other kernel sources than this build ({source_hash()})"
    return prof["schedules"].get(schedule, {}), None


def build_scene(args):
    import wgpu_renderer_amd  # noqa: F401
    from wgpu_renderer_amd import abi, scene

    if args.sponza:
        if not os.path.exists(args.sponza):
            raise SystemExit(f"--sponza {args.sponza}: no such file (res/models/Sponza.glb is not part of the reference checkout)")
        from wgpu_renderer_amd import gltf_loader
        model, textures = gltf_loader.load_glb(args.sponza)
        # the same file through the library's own importer (the one a host without Python uses)
        g = abi.Glb(path=args.sponza)
        native = (g.mesh_count, g.material_count, g.texture_count)
        g.close()
        if native != (len(model.meshes), len(model.materials), len(textures)):
            raise SystemExit(f"importers disagree on {args.sponza}: native {native}, python {(len(model.meshes), len(model.materials), len(textures))}")
        cube = scene.cube_model()
        sc = scene.Scene(textures, model, cube, scene.cube_model(),
                         abi.instance_raw(scene.REFERENCE_GEOMETRY_INSTANCE["position"]),
                         abi.instance_raw(scene.REFERENCE_FOG_INSTANCE["position"], (1, 0, 0, 0), scene.REFERENCE_FOG_INSTANCE["scale"]))
        print(f"loaded {args.sponza}: V={model.vertex_count} I={3 * model.triangle_count} P={len(model.meshes)} M={len(model.materials)} "
              f"textures={len(textures)}", file=sys.stderr, flush=True)
        return sc, os.path.basename(args.sponza)
    return scene.sponza_standin(detail=args.detail, tex_size=args.tex_size), "procedural Sponza stand-in (seed 0x5EED)"


def algorithmic_bytes(sc, W, H, S, covered, fog_frags):
    """Compulsory HBM traffic of one frame: shadow 6x(V*12 + I*4) read + 6*S^2*4 written;
    raster V*12 + I*4 read + W*H*8 written; shade covered*(8 + 3*56 + 3*4 + 16*4) + W*H*4
    written; fog fragments*(4 + 4 + 4 + 64)."""
    V = sc.geometry.vertex_count
    I = sc.geometry.triangle_count * 3
    shadow = 6 * (V * 12 + I * 4) + 6 * S * S * 4
    raster = V * 12 + I * 4 + W * H * 8
    shade = covered * (8 + 3 * 56 + 3 * 4 + 16 * 4) + W * H * 4
    fog = fog_frags * (4 + 4 + 4 + 64)
    return shadow + raster + shade + fog


def cpu_baseline(sc, cam, light, glob, passes, W, H, S, threads, frames=3):
    """Times the CPU restatement (oracle/) on full frames of the same workload (median)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle

    o = Oracle(W, H, S)
    o.upload_scene(sc)
    times = []
    for _ in range(frames):
        t0 = time.perf_counter()
        o.render_frame(cam, light, glob, passes, threads=threads)
        times.append(time.perf_counter() - t0)
    cnt = o.counters()
    o.close()
    return float(np.median(times)), cnt


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["frame", "posebatch", "tiles4k"], default="frame")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--shadow-size", type=int, default=2048)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--time", type=float, default=2.0, help="animation time t (state.rs:609-617)")
    ap.add_argument("--sponza", type=str, default=os.environ.get("WR_SPONZA_GLB", ""))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--poses", type=int, default=512)
    ap.add_argument("--skip-march-alone", action="store_true",
                    help="leave out the extra one-frame-in-flight pass after the timed region (for profiling runs)")
    ap.add_argument("--no-scale-workloads", action="store_true",
                    help="leave out the posebatch and tiles4k measurements after the timed region (for profiling runs)")
    ap.add_argument("--animate", action="store_true",
                    help="posebatch: t = time + k/60 for pose k (light, fog drift and shadow maps change every frame)")
    ap.add_argument("--reuse-shadows", action="store_true",
                    help="posebatch with a fixed light: render the shadow maps once and reuse them (wr_render_batch's shared-shadow mode)")
    ap.add_argument("--frames-in-flight", type=int, default=4, choices=[1, 2, 3, 4],
                    help="sets of frame buffers the library alternates between (WrConfig.frames_in_flight)")
    ap.add_argument("--print-launch", action="store_true",
                    help="print, as JSON, the command --gpus N would start its ranks with, and exit")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: every rank renders on device 0 (the numbers mean nothing; "
                         "tiles4k is left out, RCCL refuses two ranks on one device)")
    return ap.parse_args(argv)


def launch_command(args, argv):
    """The child process `python bench.py --gpus N` starts when it is not already one of N ranks:
    torch.distributed.run with one rank per GPU on this node, rendezvous on 127.0.0.1."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    child_args = [a for a in argv if a != "--print-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + child_args


def launch_ranks(args, argv):
    """Starts the N ranks as a fresh child process (never a re-exec: nothing in this process has
    touched a device, and nothing will), relays their output and returns their exit status."""
    import subprocess
    cmd = launch_command(args, argv)
    if args.print_launch:
        print(json.dumps({"launch": cmd, "ranks": args.gpus}), flush=True)
        return 0
    import torch
    have = torch.cuda.device_count()  # counts devices without initialising one
    if have < args.gpus and not (args.share_device and have >= 1):
        print(f"bench.py: --gpus {args.gpus} asked for, {have} HIP device(s) visible: not starting", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in child.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


class Ranks:
    """The ranks of one run and what they exchange, all of it over gloo on the host: a barrier,
    the maximum of a per-rank time, an object from rank 0."""

    def __init__(self, share_device=False):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = 0 if share_device else int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            self.dist = dist

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_time(self, dt):
        if self.world == 1:
            return dt
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def from_root(self, obj):
        if self.world == 1:
            return obj
        box = [obj if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def timed(ranks, sync, body):
    """barrier + device idle, body(), device idle + barrier: the slowest rank's seconds."""
    import torch
    torch.cuda.synchronize()
    sync()
    ranks.barrier()
    t0 = time.perf_counter()
    body()
    torch.cuda.synchronize()
    sync()
    dt = time.perf_counter() - t0
    ranks.barrier()
    return ranks.max_time(dt)


def guarded(ranks, workload):
    """A scale workload that fails must not take the frame line, measured before it, down with
    it: the failure becomes the workload's entry.  The ranks agree on the outcome (a rank that
    failed alone would leave the others waiting in the next barrier)."""
    err, out = None, None
    try:
        out = workload()
    except Exception as e:  # noqa: BLE001 (reported in the line, whatever it was)
        err = f"{type(e).__name__}: {e}"
    if ranks.max_time(1.0 if err else 0.0) > 0.0:
        return {"error": err or "failed on another rank"}
    return out


def run_posebatch(ranks, sc, args, animate):
    """configs[4]: the 512-pose flythrough split contiguously over the ranks (total work fixed).
    Fixed t: wr_render_batch in groups of sixteen poses that share the group's shadow maps (the
    groups rotate through the sets of frame buffers and overlap on the device).  Animated:
    t = time + k/60 for pose k, every pass of every frame."""
    from wgpu_renderer_amd import abi, scene, sharding
    W, H, S = 1920, 1080, args.shadow_size
    rc = scene.REFERENCE_CAMERA
    mine = list(sharding.pose_shard(args.poses, ranks.rank, ranks.world))
    cams = []
    for k in mine:
        pos, pitch, yaw = scene.flythrough_pose(k, args.poses)
        cams.append(abi.camera_uniform(pos, pitch, yaw, rc["fovy"], W, H))
    r = abi.Renderer(W, H, S, device=ranks.local_rank, frames_in_flight=4)
    r.upload_scene(sc)
    light, glob = abi.light_at_time(args.time), abi.global_uniforms(args.time, 1)

    def body():
        if animate:
            for j, k in enumerate(mine):
                t = args.time + k / 60.0
                r.render_frame(cams[j], abi.light_at_time(t), abi.global_uniforms(t, 1), abi.PASS_ALL)
        else:
            for j in range(0, len(cams), 16):
                r.render_batch(cams[j:j + 16], light, glob, abi.PASS_ALL, shadows_every_frame=False)

    for c in cams[:4]:
        r.render_frame(c, light, glob, abi.PASS_ALL)
    dt = timed(ranks, r.sync, body)
    r.close()
    return {"value": args.poses / dt, "unit": "frames/s", "poses": args.poses, "poses_per_rank": len(mine), "seconds": dt, "scaling": "strong",
            "note": ("t = time + k/60 per pose: every pass of every frame" if animate else
                     "t fixed: wr_render_batch, shadow maps shared by sixteen poses")}


def run_tiles4k(ranks, sc, args, steps):
    """configs[3]: one 3840x2160 frame, its 64x64 bins dealt round-robin over the ranks, gathered
    on rank 0 by wr_gather_bins behind every frame (total work fixed)."""
    from wgpu_renderer_amd import abi, scene
    W, H, S = 3840, 2160, args.shadow_size
    rc = scene.REFERENCE_CAMERA
    cam = abi.camera_uniform(rc["position"], rc["pitch"], rc["yaw"], rc["fovy"], W, H)
    light, glob = abi.light_at_time(args.time), abi.global_uniforms(args.time, 1)
    world = ranks.world
    r = abi.Renderer(W, H, S, device=ranks.local_rank, tile_rank=ranks.rank if world > 1 else 0, tile_world=world, frames_in_flight=1)
    r.upload_scene(sc)
    comm = None
    if world > 1:
        # the library's own communicator (wr_comm_*): rank 0 makes the id, gloo carries its 128 bytes
        uid = None
        if ranks.rank == 0:
            try:
                uid = abi.comm_unique_id()
            except abi.WrError as e:  # every rank must leave here together
                uid = str(e)
        uid = ranks.from_root(uid)
        if isinstance(uid, str):
            r.close()
            raise RuntimeError(uid)
        comm = abi.comm_create(uid, ranks.rank, world, ranks.local_rank)

    def body():
        for _ in range(steps):
            r.render_frame(cam, light, glob, abi.PASS_ALL)
            if comm is not None:
                r.gather_bins(comm, 0)  # queued behind the frame on the context's stream: no host wait

    for _ in range(3):
        r.render_frame(cam, light, glob, abi.PASS_ALL)
        if comm is not None:
            r.gather_bins(comm, 0)
    dt = timed(ranks, r.sync, body)
    out = {"value": steps / dt, "unit": "frames/s", "frames": steps, "seconds": dt, "scaling": "strong",
           "gathered_bytes_per_frame": (W * H * 4 * (world - 1)) // world if world > 1 else 0,
           "rccl_library": abi.comm_library_path() if comm is not None else None,
           "note": "3840x2160, bins interleaved over ranks, grouped RCCL send/recv to rank 0 behind every frame; one frame in flight"}
    if comm is not None:
        abi.comm_destroy(comm)
    r.close()
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.print_launch or (args.gpus > 1 and "WORLD_SIZE" not in os.environ):
        # not yet one of N ranks: become their parent, before anything here touches a device
        raise SystemExit(launch_ranks(args, argv))

    import torch

    ranks = Ranks(args.share_device)
    world, rank, local_rank = ranks.world, ranks.rank, ranks.local_rank
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the render path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no HIP device {local_rank} ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)

    import wgpu_renderer_amd  # noqa: F401
    from wgpu_renderer_amd import abi, scene

    sc, scene_name = build_scene(args)
    if args.workload != "frame":
        # one of the scale workloads as the line itself
        res = (run_tiles4k(ranks, sc, args, args.steps) if args.workload == "tiles4k"
               else run_posebatch(ranks, sc, args, args.animate))
        if rank == 0:
            print(json.dumps({"metric": f"frames/sec {args.workload}", "value": res["value"], "unit": "frames/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / res["value"], "higher_is_better": True,
                              "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": res["note"], "scene": scene_name}, "detail": res}), flush=True)
        ranks.close()
        return

    W, H, S = args.width, args.height, args.shadow_size
    passes = abi.PASS_ALL
    light = abi.light_at_time(args.time)
    glob = abi.global_uniforms(args.time, 1)
    rc = scene.REFERENCE_CAMERA
    cam = abi.camera_uniform(rc["position"], rc["pitch"], rc["yaw"], rc["fovy"], W, H)

    in_flight = args.frames_in_flight
    r = abi.Renderer(W, H, S, device=local_rank, frames_in_flight=in_flight)
    r.upload_scene(sc)

    def step(i):
        r.render_frame(cam, light, glob, passes)

    # one counted frame (outside the timed region) for the algorithmic work of this frame
    r.set_counting(True)
    step(0)
    st = r.stats()
    fog_frags, noise_evals, shaded = int(st.fog_fragments), int(st.fog_noise_evals), int(st.shaded_fragments)
    evals_view, evals_light = int(st.fog_noise_evals_view), int(st.fog_noise_evals_light)
    replay_px, replay_slow = int(st.replay_pixels), int(st.replay_slow_pixels)
    bin_items = (int(st.geometry_bin_items), int(st.shadow_bin_items))
    r.set_counting(False)

    for i in range(args.warmup):
        step(i)
    # In the timed region only the dominant kernel is bracketed by HIP events (on its own stream);
    # the dozen events of the per-stage times cost a lone frame 3 % and are taken afterwards.
    r.set_profiling(2)

    def timed_frames():
        for i in range(args.steps):
            step(i)

    dt = timed(ranks, r.sync, timed_frames)
    st = r.stats()
    r.set_profiling(1)
    for i in range(min(args.steps, 20)):
        step(i)
    st_stages = r.stats()
    r.set_profiling(0)
    r.close()  # (its streams would share hardware queues with the next renderer's)

    # One frame at a time (the reference's own cadence, state.rs:678-806), outside the timed
    # region: frame rate and the dominant kernel of that schedule.
    one_frame = None
    if rank == 0 and in_flight > 1 and not args.skip_march_alone:
        r1 = abi.Renderer(W, H, S, device=local_rank, frames_in_flight=1)
        r1.upload_scene(sc)
        r1.set_counting(True)
        r1.render_frame(cam, light, glob, passes)
        s1 = r1.stats()
        ev1 = int(s1.fog_noise_evals_view)
        r1.set_counting(False)
        for i in range(20):
            r1.render_frame(cam, light, glob, passes)
        r1.sync()
        # back to back: frames queued one behind the other on the one set of buffers (the next
        # frame's vertex and triangle set-up runs next to the previous frame's fog tail)
        t1 = time.perf_counter()
        for i in range(200):
            r1.render_frame(cam, light, glob, passes)
        r1.sync()
        dt1 = (time.perf_counter() - t1) / 200
        # alone: the host waits for every frame before it submits the next
        t1 = time.perf_counter()
        for i in range(50):
            r1.render_frame(cam, light, glob, passes)
            r1.sync()
        lone = (time.perf_counter() - t1) / 50
        r1.set_profiling(2)
        for i in range(20):
            r1.render_frame(cam, light, glob, passes)
        s1 = r1.stats()
        r1.set_profiling(1)
        for i in range(20):
            r1.render_frame(cam, light, glob, passes)
        s1l = r1.stats()
        one_frame = {"frames_per_s": 1.0 / dt1, "lone_frame_ms": lone * 1e3, "view_ms": float(s1.ms_fog_march),
                     "light_ms": float(s1l.ms_fog_light_march), "evals_view": ev1}
        r1.close()
    ranks.barrier()

    # BASELINE.json's configs[4] and configs[3] in the same run (every rank takes part)
    scale_workloads = None
    if not args.no_scale_workloads and not args.sponza and args.detail == 1.0:
        scale_workloads = {
            "posebatch": guarded(ranks, lambda: run_posebatch(ranks, sc, args, animate=False)),
            "posebatch_animated": guarded(ranks, lambda: run_posebatch(ranks, sc, args, animate=True)),
            "tiles4k": guarded(ranks, lambda: run_tiles4k(ranks, sc, args, steps=40)) if not (args.share_device and world > 1) else None,
        }

    total_frames = args.steps * world
    fps = total_frames / dt
    ms_per_step = dt / args.steps * 1e3

    default_frame = ((W, H, S) == (1920, 1080, 2048) and args.detail == 1.0 and args.tex_size == 1024 and not args.sponza)
    if rank == 0:
        # The dominant kernel, bracketed by its own HIP events on its stream: with frames
        # overlapping, k_fog_march (both marches of an item); with one frame in flight,
        # k_fog_noise_view (the noise of the view-march steps).
        latency_schedule = in_flight == 1
        kernel = "k_fog_noise_view<false>" if latency_schedule else "k_fog_march<false>"
        kernel_evals = evals_view if latency_schedule else noise_evals
        launch_s = st.ms_fog_march * 1e-3
        launch_flops = FLOP_PER_FOG_NOISE_EVAL * kernel_evals
        # The frame's algorithmic fog arithmetic: every fog_noise evaluation the two marches of
        # fog.wgsl execute for this frame, whichever kernels carry them.
        frame_flops = FLOP_PER_FOG_NOISE_EVAL * noise_evals
        frame_bytes = algorithmic_bytes(sc, W, H, S, shaded, fog_frags)
        prof, why_not = load_profile("1" if latency_schedule else "4") if default_frame else (None, "counters are recorded for the default frame only")
        executed = frame_valu = traffic = frac_profile = profile_launch_ms = None
        if prof:
            pk = prof.get(kernel, {})
            if pk.get("valu_wave_instructions") and launch_s > 0:
                executed = {
                    "valu_wave_instructions_per_launch": pk["valu_wave_instructions"],
                    "issue_peak_wave_instructions_per_ns": VALU_ISSUE_PEAK_PER_NS,
                    "issue_frac_of_launch": pk["valu_wave_instructions"] / (launch_s * 1e9 * VALU_ISSUE_PEAK_PER_NS),
                    "lanes_active": (pk["thread_cycles_valu"] / (pk["valu_wave_instructions"] * 64.0)) if pk.get("thread_cycles_valu") else None,
                }
            if pk.get("fetch_size_kb") is not None and pk.get("write_size_kb") is not None:
                traffic = int((2 * pk["fetch_size_kb"] + pk["write_size_kb"]) * 1024)  # FETCH_SIZE doubled: gfx950 note of the guide
            if pk.get("trace_avg_ns"):
                profile_launch_ms = pk["trace_avg_ns"] * 1e-6
                frac_profile = launch_flops / (pk["trace_avg_ns"] * 1e-9) / 1e12 / PEAK_FP32_TFLOPS
            per_frame = sum(v["valu_wave_instructions"] or 0 for k, v in prof.items() if "<true>" not in k and "k_fill" not in k)
            frame_valu = {
                "wave_instructions_per_frame": per_frame,
                "issue_frac": per_frame / (ms_per_step * 1e6 * VALU_ISSUE_PEAK_PER_NS),
                "note": "all kernels of one frame (counter pass, kernels serialised) against the frame time of the timed region",
            }
        # Launches of consecutive frames overlap each other, so a launch's duration is not device
        # time: launch time / frame time = how many launches of the dominant kernel are running on
        # average.  It cannot exceed the frames in flight (one launch per frame).
        overlap = st.ms_fog_march / ms_per_step if ms_per_step > 0 else None
        if overlap is not None and int(st.profiled_frames) > 0:
            assert 0.0 < overlap <= in_flight + 0.25, f"dominant kernel {st.ms_fog_march:.3f} ms per launch cannot come from {in_flight} frame(s) in flight at {ms_per_step:.3f} ms per frame"
        frame_tflops = frame_flops / (ms_per_step * 1e-3) / 1e12
        roofline = {
            "kernel": kernel + (" (fog.wgsl fragment stage: noise of the view-march steps, 2 x 8 octaves of value noise each)" if latency_schedule
                                else " (fog.wgsl fragment stage: two ray marches through 2 x 8 octaves of value noise)"),
            "bound": "valu",
            "bound_note": "FP32 vector ALU: no stage of the path is a contraction, so no MFMA; the f32 vector peak priced against "
                          "equals the dense f32 MFMA peak; on straight-line v_fmac_f32 code this device issues 0.82-0.91 of the nominal "
                          "rate executed.issue_frac_of_launch is priced against (scripts/dev_ifetch_bench.hip)",
            "achieved": frame_tflops,
            "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": frame_tflops / PEAK_FP32_TFLOPS,
            "frac_note": "on the clock of the timed region: algorithmic flops of all fog_noise evaluations of one frame (5198 each) / ms_per_step / peak",
            "traffic": traffic,
            "flop_per_frame": frame_flops,
            "fog_noise_evals_per_frame": noise_evals,
            "launch_ms": st.ms_fog_march,
            "launch_note": "HIP-event time of one launch of the dominant kernel on its stream; launches of consecutive frames overlap "
                           "(overlap_factor of them run at a time), so this is not device time and is not what frac divides by",
            "overlap_factor": overlap,
            "launches_averaged": int(st.profiled_frames),
            "flop_per_launch": launch_flops,
            "fog_noise_evals_per_launch": kernel_evals,
            "frac_launch": (launch_flops / launch_s / 1e12 / PEAK_FP32_TFLOPS) if launch_s > 0 else None,
            "frac_profile": frac_profile,
            "profile_launch_ms": profile_launch_ms,
            "frac_profile_note": "flop_per_launch / the kernel's average duration in the committed rocprofv3 kernel trace (profiles/r03_kernel_stats.txt "
                                 "or r03_one_frame_stats.txt) / peak: comparable with frac_launch, not with frac",
            "algorithmic_note": "flops price every fog_noise evaluation at its algorithmic 5198 flop, also the octaves the kernel reads from "
                                "per-frame hash tables or proves irrelevant; frame_valu.issue_frac is what the frame's kernels issue against the VALU issue peak",
            "executed": executed,
            "frame_valu": frame_valu,
            "profile": {"file": os.path.relpath(PROFILE_JSON, ROOT), "source_hash": source_hash(), "used": bool(prof), "reason": why_not},
            "one_frame_in_flight": None if not one_frame else {
                "frames_per_s": one_frame["frames_per_s"],
                "lone_frame_ms": one_frame["lone_frame_ms"],
                "note": "frames_per_s: 200 frames queued back to back on one set of frame buffers, where a frame's geometry set-up overlaps "
                        "the previous frame's fog tail; lone_frame_ms: submit, then wait, frame by frame (includes the host's submission)",
                "kernel": "k_fog_noise_view<false>",
                "launch_ms": one_frame["view_ms"],
                "light_march_launch_ms": one_frame["light_ms"],
                "frac": frame_flops * one_frame["frames_per_s"] / 1e12 / PEAK_FP32_TFLOPS,
                "frac_launch": FLOP_PER_FOG_NOISE_EVAL * one_frame["evals_view"] / (one_frame["view_ms"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS if one_frame["view_ms"] > 0 else None},
        }
        stage_ms = {"shadow": st_stages.ms_shadow, "geometry": st_stages.ms_geometry, "shade": st_stages.ms_shade, "fog": st_stages.ms_fog,
                    "note": "20 frames after the timed region, with an event pair around every stage"}
        hbm = {
            "algorithmic_bytes_per_frame": frame_bytes,
            "achieved_GBps": frame_bytes * fps / world / 1e9,
            "peak_GBps": PEAK_HBM_GBS,
            "frac": frame_bytes * fps / world / 1e9 / PEAK_HBM_GBS,
        }
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            threads = min(args.cpu_threads, os.cpu_count() or 1)
            cdt, ccnt = cpu_baseline(sc, cam, light, glob, passes, W, H, S, threads)
            cpu = {
                "value": 1.0 / cdt, "unit": "frames/s", "cores": threads, "kind": "port",
                "sample": f"median of 3 full frames of the same workload ({W}x{H}, shadows+fog) on the CPU restatement in oracle/ "
                          f"(reference wgpu-on-lavapipe path not buildable offline); {cdt:.2f} s per frame",
                "fog_noise_evals": int(ccnt["fog_noise_evals"]),
                "fog_noise_evals_unused": int(ccnt["fog_noise_evals_unused"]),
            }
        out = {
            "metric": "frames/sec Sponza 1080p shadows+fog",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"configs[2]: Sponza-class scene {W}x{H}, shadow map 6x{S}^2 + PCF, PBS, light marker, fog; "
                            "reference default camera (-500,150,0) yaw 0, t=2.0; every pass recomputed each frame",
                "scene": scene_name,
                "triangles": sc.geometry.triangle_count,
                "vertices": sc.geometry.vertex_count,
                "materials": len(sc.geometry.materials),
                "primitives": len(sc.geometry.meshes),
                "textures": len(sc.textures),
                "texture_size": args.tex_size if "stand-in" in scene_name else None,
                "parallelism": f"frame-parallel x{world}",
                "frames_in_flight": in_flight,
                "hip_hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),
                "control_plane": "gloo" if world > 1 else None,
                "share_device": bool(args.share_device and world > 1),
            },
            "stage_ms": stage_ms,
            "fog_fragments": fog_frags,
            "fog_noise_evals": noise_evals,
            "fog_noise_evals_evaluated": {"view": evals_view, "light": evals_light},
            "shaded_fragments": shaded,
            "ordered_blend_pixels": replay_px,
            "ordered_blend_slow_pixels": replay_slow,
            "bin_items": {"geometry": bin_items[0], "shadow": bin_items[1]},
            "roofline": roofline,
            "roofline_hbm_frame": hbm,
            "cpu_baseline": cpu,
            "scale_workloads": scale_workloads,
        }
        print(json.dumps(out), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
