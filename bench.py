#!/usr/bin/env python3
"""Throughput of the MocapV2 per-frame hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dist mild|zero]

Workload (BASELINE.json configs[1]): 6 cameras x 1920x1080 synthetic IR frames, 8 markers.  One "step" = one pass
of the whole hot path (undistort -> box blur -> threshold -> median -> contours -> centroids -> epipolar
correspondence -> DLT triangulation) over one batch of T = 512 time steps (3072 camera images, 6.4 GB) that is
resident in HBM before the timed region starts.  A "frame" = one time step of all 6 cameras.  With N ranks every
rank processes its own 3072 images (weak scaling; camera-major sharding + one all-gather, mocapv2_amd/pipeline.py).

Three batches are in flight on three HIP streams (--depth).  Prints ONE JSON line (rank 0) with the driver's fields
plus `roofline` (the filter stage = streaming scan + patches + filter kernels against the HBM read roofline, from HIP
events recorded on the launch streams) and `cpu_baseline` (the C oracle on host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CAM, WIDTH, HEIGHT, N_MARKERS, T_STEPS = 6, 1920, 1080, 8, 512
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def render_local(scene, image_list, seed_base=1000):
    """uint8 [n, H, W] frames for (camera, global time step) pairs; time step t uses seed seed_base + t."""
    out = np.empty((len(image_list), scene.height, scene.width), np.uint8)
    cache = {}
    for i, (c, t) in enumerate(image_list):
        if t not in cache:
            rng = np.random.default_rng(seed_base + t)
            cache = {t: scene.markers(rng, N_MARKERS)}
        rng = np.random.default_rng((seed_base + t) * 64 + c)
        out[i] = scene.render(rng, cache[t], c, radius_range=(16.0, 22.0), salt=0.001)
    return out


def cpu_baseline(scene, arrays, frames, n_steps, bayer=False):
    """The oracle (scalar C port of the reference's CPU path) on `n_steps` time steps of the benchmark batch
    (frames uint8 [T, C, H, W], reused cyclically): thread-per-camera blob extraction as the reference does
    (RealtimeTracking_FLIR.py:307-312), then correspondence + DLT."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor

    import oracle
    K, dist, R, t, F = arrays
    prm = oracle.default_params(undistort=True, filter_order=2)
    oracle.lib()
    T = frames.shape[0]
    pool = ThreadPoolExecutor(N_CAM)
    t0 = time.perf_counter()
    n_pts = 0
    for i in range(n_steps):
        s = i % T
        gray = (lambda im: oracle.bayer_gray(im, 3, 14)) if bayer else (lambda im: im)
        lists = list(pool.map(lambda c: oracle.find_dot(gray(frames[s, c]), K[c], dist[c], params=prm), range(N_CAM)))
        P = max(1, max(len(l) for l in lists))
        pts = np.zeros((N_CAM, P, 2))
        cnt = np.zeros(N_CAM, np.int32)
        for c, l in enumerate(lists):
            cnt[c] = len(l)
            if l:
                pts[c, :len(l)] = l
        res = oracle.correspond(pts, cnt, K, dist, R, t, F)
        n_pts += len(res["root"])
    dt = time.perf_counter() - t0
    pool.shutdown()
    return n_steps / dt, dt, n_pts, (lists, res)


def main():
    global T_STEPS, N_MARKERS, WIDTH, HEIGHT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dist", choices=["mild", "zero"], default="mild")
    ap.add_argument("--time-steps", type=int, default=T_STEPS,
                    help="time steps per batch and GPU (one step = one pass over time_steps x 6 resident images)")
    ap.add_argument("--depth", type=int, default=3,
                    help="batches in flight (software pipelining of consecutive batches on separate HIP streams; 1 = off)")
    ap.add_argument("--from-host", action="store_true",
                    help="the batch stays in pinned host memory and is uploaded every step (PCIe-inclusive rate; not the headline)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks share cuda:0 and exchange through gloo (checks the multi-rank code path on a one-GPU box; "
                         "not a measurement)")
    ap.add_argument("--markers", type=int, default=N_MARKERS,
                    help="markers per frame: 8 = BASELINE.json configs[1] (the headline), 32 = configs[2] (a secondary measurement)")
    ap.add_argument("--frame", default=f"{WIDTH}x{HEIGHT}",
                    help="frame size WxH: 1920x1080 = the headline workload; 3840x2160 = the frame size of BASELINE.json configs[4] "
                         "(a secondary measurement; lower --time-steps accordingly)")
    ap.add_argument("--bayer", action="store_true",
                    help="secondary measurement: the resident frames are raw Bayer GR sensor frames and every step starts with the "
                         "Bayer -> gray pre-pass (RealtimeTracking_FLIR.py:103-104); not the headline workload")
    ap.add_argument("--cpu-steps", type=int, default=256, help="time steps in the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="skip the second timed run with the other distortion variant")
    args = ap.parse_args()
    T_STEPS = args.time_steps
    N_MARKERS = args.markers
    WIDTH, HEIGHT = (int(v) for v in args.frame.lower().split("x"))
    max_points = 32 if N_MARKERS <= 16 else 2 * N_MARKERS  # centroid record capacity per image

    import torch
    import torch.distributed as dist
    from mocapv2_amd.pipeline import BatchTracker, scene_arrays
    from mocapv2_amd.synth import MILD_DIST, ZERO_DIST, Scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with python -m torch.distributed.run --nproc-per-node N")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(tracker, frames):
        """W untimed + K timed steps of the resident batch, barrier + synchronize on both sides, max over ranks.
        Then a short sequential pass (one batch at a time, a synchronize after each) for per-kernel durations that are
        not stretched by the neighbouring batches' kernels (in the timed region `--depth` batches share the chip)."""
        for _ in range(args.warmup):
            tracker.step(frames)
        tracker.synchronize()
        barrier()
        tracker.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = tracker.step(frames)
        tracker.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        tracker.profile(False)
        prof_timed = tracker.profile_read()
        n_seq = max(1, min(args.steps, 6))
        tracker.profile(True)
        for _ in range(n_seq):
            out = tracker.step(frames)
            tracker.synchronize()
        tracker.profile(False)
        prof = tracker.profile_read()
        prof["steps"] = n_seq
        prof["timed_region"] = prof_timed
        prof["tiles"], prof["tiles_skipped"] = tracker.ctx.tile_stats()  # of the last step
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return out, elapsed, prof

    def measure(dist_name):
        """Build the scene / tracker / resident batch for one distortion variant and time K steps of it."""
        scene = Scene(N_CAM, WIDTH, HEIGHT, dist=MILD_DIST if dist_name == "mild" else ZERO_DIST)
        arrays = scene_arrays(scene)
        tracker = BatchTracker(*arrays, WIDTH, HEIGHT, T_STEPS, world=world, rank=rank, device=local_rank, depth=args.depth,
                               max_points=max_points, bayer_pattern=3 if args.bayer else None)
        images = tracker.local_image_list()
        frames_host = render_local(scene, images)
        frames = torch.from_numpy(frames_host).pin_memory() if args.from_host else torch.from_numpy(frames_host).cuda()
        torch.cuda.synchronize()
        out, elapsed, prof = timed(tracker, frames)
        return scene, arrays, tracker, images, frames_host, out, elapsed, prof, frames

    scene, arrays, tracker, images, frames_host, out, elapsed, prof, frames = measure(args.dist)

    n_roots = out["n"].cpu().numpy()
    status_ok = bool((n_roots >= 0).all()) and bool((tracker.records[:, 0] >= 0).all().item())

    if rank == 0:
        frames_per_step = T_STEPS * world
        value = frames_per_step * args.steps / elapsed
        KERNELS = (("scan", "bright_cells_kernel"), ("patch", "undistort_patches_kernel"), ("filter", "filter_mask_kernel"))

        def roofline_of(prof, n_images, n_launch_groups, dist_name):
            """Roofline entries of the two HBM-bound kernels of the filter stage; the one with the longer launches leads.
            Durations are HIP-event averages recorded by the library on the launch stream: `avg_launch_ms` from the
            sequential pass (one batch at a time), `avg_launch_ms_in_timed_region` from the timed region itself, where
            `--depth` batches are in flight and a kernel shares the chip with its neighbours' kernels."""
            per_launch = n_images / n_launch_groups
            traffic = {}
            tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tpath):  # HBM bytes per launch from the PMC passes (profiles/README.md), same workload
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("dist") == dist_name and tj.get("images_per_launch") == per_launch and tj.get("markers", 8) == N_MARKERS \
                        and (WIDTH, HEIGHT) == (1920, 1080) and not args.bayer:
                    traffic = tj.get("hbm_bytes_per_launch", {})
            ent = {}
            for key, name in KERNELS:
                n = prof[key + "_launches"]
                if n == 0:
                    continue
                ms = prof[key + "_ms"] / n
                ach = WIDTH * HEIGHT * per_launch / (ms * 1e-3) / 1e9
                ent[name] = {"kernel": name, "avg_launch_ms": round(ms, 4), "traffic": traffic.get(name)}
                if key == "scan" or len([k for k, _ in KERNELS if prof[k + "_launches"]]) == 1:
                    # the kernel that reads every frame byte: priced alone against the same algorithmic traffic
                    ent[name].update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(ach / HBM_PEAK_GBS, 4), "images_per_launch": per_launch,
                                      "algorithmic_bytes_per_image": WIDTH * HEIGHT})
                tr = prof.get("timed_region")
                if tr and tr[key + "_launches"]:
                    ent[name]["avg_launch_ms_in_timed_region"] = round(tr[key + "_ms"] / tr[key + "_launches"], 4)
            # The filter stage = the streaming scan (reads every pixel once, marks the tiles that can hold set pixels) + the
            # filter kernel (re-reads and filters only those tiles).  Its algorithmic traffic is one read of the frames,
            # so the stage is priced as a whole: bytes / (sum of the two kernels' average launch durations).
            both_ms = sum(e["avg_launch_ms"] for e in ent.values())
            ach = WIDTH * HEIGHT * per_launch / (both_ms * 1e-3) / 1e9
            tr_known = [traffic.get(k) for k in ent]
            roof = {"bound": "hbm", "kernel": " + ".join(ent), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": sum(tr_known) if all(t is not None for t in tr_known) else None,
                    "avg_launch_ms": round(both_ms, 4), "images_per_launch": per_launch,
                    "algorithmic_bytes_per_image": WIDTH * HEIGHT,
                    "per_kernel": ent,
                    "note": "the stage's algorithmic traffic is one read of the frames; bright_cells_kernel is the kernel that "
                            "moves those bytes (priced alone in per_kernel), the other kernels touch the marked tiles only"}
            return roof

        def kernel_ms(prof):
            n = prof["steps"]
            return {"scan": round(prof["scan_ms"] / n, 4), "patches": round(prof["patch_ms"] / n, 4),
                    "filter": round(prof["filter_ms"] / n, 4),
                    "contours": round(prof["contour_ms"] / n, 4), "correspond": round(prof["corr_ms"] / n, 4)}

        roof = roofline_of(prof, len(images), 1 if world == 1 else len(tracker.segs), args.dist)
        line = {
            "metric": "frames/sec (6-cam 1080p)" if (WIDTH, HEIGHT) == (1920, 1080) else f"frames/sec (6-cam {WIDTH}x{HEIGHT})", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8+int64+f64",
            "data": "synthetic",
            "config": {"workload": "6-camera 1920x1080 synthetic IR frames, 8 markers (BASELINE.json configs[1])"
                       if (N_MARKERS, WIDTH, HEIGHT) == (8, 1920, 1080) else
                       f"6-camera {WIDTH}x{HEIGHT} synthetic IR frames, {N_MARKERS} markers"
                       + (" (BASELINE.json configs[2])" if N_MARKERS == 32 else ""),
                       "cameras": N_CAM, "width": WIDTH, "height": HEIGHT, "markers": N_MARKERS,
                       "time_steps_per_step_per_gpu": T_STEPS, "distortion": args.dist,
                       "frames_resident_in_hbm": not args.from_host, "batches_in_flight": args.depth,
                       "input": "raw Bayer GR frames, gray conversion inside every step" if args.bayer else "gray frames",
                       "parallelism": "single launch, time-major" if world == 1 else f"camera-major blocks x{world} + 1 all-gather"},
            "roofline": roof,
            "kernel_ms_per_step": kernel_ms(prof),
            "status_ok": status_ok,
            "points_per_frame": float(n_roots.mean()),
            "dark_tile_early_out": {"tiles_per_step": prof["tiles"], "tiles_resolved_without_filtering": prof["tiles_skipped"],
                                    "note": "exact: bright_cells_kernel reads the frames once and sums the excess over 63 per 8x8 cell; "
                                            "a (240 col x 68 row) filter tile whose cells prove that no mask bit can be set is "
                                            "answered with zeros (DESIGN.md 4.1); disabled run below"},
        }
        if world == 1 and args.secondary:
            ref_out = {k: v.clone() for k, v in out.items()}  # the tracker reuses its output buffers
            ref_rec = tracker.records.clone()
            os.environ["MOCAP_SKIP_DARK"] = "0"  # same batch, early-out off: every tile runs the full filter
            out_d, el_d, prof_d = timed(tracker, frames)
            del os.environ["MOCAP_SKIP_DARK"]
            roof_d = roofline_of(prof_d, len(images), 1, None)
            line["without_early_out"] = {"value": round(T_STEPS * args.steps / el_d, 2), "unit": "frames/s",
                                         "ms_per_step": round(1e3 * el_d / args.steps, 4),
                                         "filter_avg_launch_ms": roof_d["avg_launch_ms"], "roofline_frac": roof_d["frac"],
                                         "same_results": bool(torch.equal(tracker.records, ref_rec) and torch.equal(out_d["n"], ref_out["n"])
                                                              and torch.equal(out_d["xyz"], ref_out["xyz"]))}
        if world == 1 and args.secondary:
            other = "zero" if args.dist == "mild" else "mild"
            _, _, tr2, im2, _, out2, el2, prof2, fr2 = measure(other)
            roof2 = roofline_of(prof2, len(im2), 1, other)
            line["other_distortion_variant"] = {
                "distortion": other, "value": round(T_STEPS * args.steps / el2, 2), "unit": "frames/s",
                "ms_per_step": round(1e3 * el2 / args.steps, 4),
                "roofline": roof2,
                "kernel_ms_per_step": kernel_ms(prof2),
                "status_ok": bool((out2["n"].cpu().numpy() >= 0).all()),
                "dark_tile_early_out": {"tiles_per_step": prof2["tiles"], "tiles_resolved_without_filtering": prof2["tiles_skipped"]},
                "note": "same workload with zero lens distortion: cv.undistort is then the identity map and the "
                        "kernel streams the frame without the remap gather (SURVEY.md section 8d lists both variants)"}
            del tr2, fr2
        if world == 1 and args.cpu_steps > 0:
            fps, dt, n_pts, last = cpu_baseline(scene, arrays, frames_host.reshape(T_STEPS, N_CAM, HEIGHT, WIDTH), args.cpu_steps, args.bayer)
            line["cpu_baseline"] = {"value": round(fps, 3), "unit": "frames/s", "cores": N_CAM, "kind": "port",
                                    "sample": f"{args.cpu_steps} time steps x {N_CAM} cameras of the same workload, "
                                              f"{dt:.1f} s, thread-per-camera C oracle ({os.cpu_count()} host cpus)"}
            # parity spot check of the timed batch against the oracle (not timed): last CPU time step
            s = (args.cpu_steps - 1) % T_STEPS
            k = int(n_roots[s])
            gpu_xyz = out["xyz"][s, :k].cpu().numpy()
            ref = last[1]
            line["parity_spot_check"] = bool(k == len(ref["root"]) and (k == 0 or np.abs(gpu_xyz - ref["xyz"]).max() < 1e-7))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
