#!/usr/bin/env python3
"""Throughput of the MocapV2 per-frame hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cameras C] [--markers M] [--frame WxH] [--dist mild|zero]

Default workload (BASELINE.json configs[1]): 6 cameras x 1920x1080 synthetic IR frames, 8 markers.  One "step" = one
pass of the whole hot path (undistort -> box blur -> threshold -> median -> contours -> centroids -> epipolar
correspondence -> DLT triangulation) over one batch of T time steps (T x C camera images, 6.4 GB by default) that is
resident in HBM before the timed region starts.  A "frame" = one time step of all C cameras.  With N ranks every rank
processes its own T x C images (weak scaling; camera-major sharding + ONE all-gather of centroid records per batch,
mocapv2_amd/pipeline.py).  Other BASELINE.json configs:
    configs[2]   --markers 32
    configs[3]   --gpus 8 --cameras 8                       (one camera per GPU, all-gather of centroids over xGMI)
    configs[4]   --gpus 8 --cameras 16 --frame 3840x2160 --markers 64   (two 4K cameras per GPU; + BA residual evaluation)
`--gpus N` without a torchrun environment launches the N ranks itself (python -m torch.distributed.run, before this
process touches a GPU); `--rehearse-on-one-gpu` lets the N ranks share cuda:0 and exchange through gloo (a check of the
multi-rank code path on a one-GPU box, not a measurement).

Three batches are in flight on three HIP streams (--depth).  Prints ONE JSON line (rank 0) with the driver's fields plus
`roofline` (the filter stage against the HBM read roofline, from HIP events recorded on the launch streams),
`cpu_baseline` (the C oracle on host cores), `parity` (3-D RMSE / centroid mismatches of the timed batch against that
oracle) and, with one GPU, `extra` sections (32 markers, bright background, 4K) measured the same way.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
DEFAULT_BATCH_BYTES = 3072 * 1920 * 1080  # images x bytes of the default batch per GPU


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dist", choices=["mild", "zero"], default="mild")
    ap.add_argument("--cameras", type=int, default=6, help="cameras of the rig: 6 = configs[1]/[2], 8 = configs[3], 16 = configs[4]")
    ap.add_argument("--time-steps", type=int, default=0,
                    help="time steps per batch and GPU (default: as many as make the batch 6.4 GB, 512 for 6 x 1080p)")
    ap.add_argument("--depth", type=int, default=3,
                    help="batches in flight (software pipelining of consecutive batches on separate HIP streams; 1 = off)")
    ap.add_argument("--from-host", action="store_true",
                    help="the batch stays in pinned host memory and is uploaded every step (PCIe-inclusive rate; not the headline)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks share cuda:0 and exchange through gloo (checks the multi-rank code path on a one-GPU box; "
                         "not a measurement)")
    ap.add_argument("--collective", choices=["auto", "rccl", "torch"], default="auto",
                    help="exchange of centroid records: rccl = mocap_allgather_centroids (ncclAllGather behind the C-ABI), "
                         "torch = torch.distributed; auto = rccl on GPUs")
    ap.add_argument("--markers", type=int, default=8,
                    help="markers per frame: 8 = BASELINE.json configs[1] (the headline), 32 = configs[2], 64 = configs[4]")
    ap.add_argument("--frame", default="1920x1080", help="frame size WxH: 1920x1080 = the headline; 3840x2160 = configs[4]")
    ap.add_argument("--background", default="0-60",
                    help="uniform background noise range lo-hi of the synthetic frames (0-60 = the headline scene)")
    ap.add_argument("--bayer", action="store_true",
                    help="secondary measurement: the resident frames are raw Bayer GR sensor frames and every step starts with the "
                         "Bayer -> gray pre-pass (RealtimeTracking_FLIR.py:103-104); not the headline workload")
    ap.add_argument("--cpu-steps", type=int, default=256, help="time steps in the CPU baseline / parity sample (0 = skip)")
    ap.add_argument("--cpu-single-steps", type=int, default=48, help="time steps of the single-thread CPU baseline leg (0 = skip)")
    ap.add_argument("--batches", type=int, default=3,
                    help="distinct resident batches the timed region rotates through (every image slot sees a different frame from "
                         "one step to the next, so the mask's on-demand clearing and the scan's base probe are inside the measurement)")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="skip the early-out-off rerun and the other distortion variant")
    ap.add_argument("--no-extra", dest="extra", action="store_false", help="skip the extra workload sections")
    return ap.parse_args(argv)


def visible_gpu_count():
    """GPUs this process would see, counted WITHOUT touching HIP (no torch.cuda call, no library load): the parent of the
    rank processes must stay GPU-free because it starts another program afterwards.  Sources: the visibility lists of the
    environment, else the KFD topology in sysfs (a node with simd_count > 0 is a GPU).  None = cannot tell."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(nodes):
            with open(os.path.join(nodes, d, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        n += 1
        return n
    except OSError:
        return None


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as children of a launcher process.  This process never
    touches a GPU: the devices are counted from the environment / sysfs (visible_gpu_count), not through torch.cuda."""
    n_dev = visible_gpu_count()
    extra = []
    if n_dev is not None and n_dev < args.gpus and not args.rehearse_on_one_gpu:
        raise SystemExit(f"--gpus {args.gpus}: only {n_dev} GPU(s) visible (add --rehearse-on-one-gpu to run the ranks on one GPU "
                         "through gloo: a functional check, not a measurement)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + extra
    return subprocess.run(cmd, env=env).returncode


class Workload:
    """Scene + frame recipe of one measured configuration."""

    def __init__(self, cameras, width, height, markers, dist_name, background=(0, 60), bayer=False):
        self.cameras, self.width, self.height, self.markers = cameras, width, height, markers
        self.dist_name, self.background, self.bayer = dist_name, tuple(background), bayer
        self.max_points = 32 if markers <= 16 else 2 * markers  # centroid record capacity per image
        # 64 markers in the default 1 m cube give single roots millions of candidate groups (the reference's cartesian expansion,
        # lib/Helpers.py:239-245, would not finish either; the kernel reports MOCAP_CORR_E_GROUPS): configs[4] spreads them over a
        # 2.4 m cube seen from a 4 m ring, as tests/test_gpu_fullsize.py::test_config5_* does
        self.spread = markers >= 64
        self.max_groups = 1 << 22 if self.spread else 4096  # candidate groups per root (the oracle's own limit for the crowded rig)

    def default_time_steps(self):
        t = DEFAULT_BATCH_BYTES // (self.cameras * self.width * self.height)
        return max(8, (t // 8) * 8)

    def name(self, world):
        std = self.dist_name == "mild" and self.background == (0, 60) and not self.bayer
        c, w, h, m = self.cameras, self.width, self.height, self.markers
        if std and (c, w, h, m) == (6, 1920, 1080, 8):
            return "6-camera 1920x1080 synthetic IR frames, 8 markers (BASELINE.json configs[1])"
        if std and (c, w, h, m) == (6, 1920, 1080, 32):
            return "6-camera 1080p, 32 markers, epipolar correspondence + batched DLT (BASELINE.json configs[2])"
        if std and (c, w, h) == (8, 1920, 1080) and world == 8:
            return f"8-camera 1080p sharded 1 cam/GPU over 8 GPUs, all-gather of centroids, {m} markers (BASELINE.json configs[3])"
        if std and (c, w, h, m) == (16, 3840, 2160, 64) and world == 8:
            return "16-camera 4K synthetic, 64 markers, 2 cameras/GPU over 8 GPUs (BASELINE.json configs[4])"
        s = f"{c}-camera {w}x{h} synthetic IR frames, {m} markers"
        if self.background != (0, 60):
            s += f", background {self.background[0]}-{self.background[1]}"
        return s

    def scene(self):
        from mocapv2_amd.synth import MILD_DIST, ZERO_DIST, Scene
        return Scene(self.cameras, self.width, self.height, dist=MILD_DIST if self.dist_name == "mild" else ZERO_DIST,
                     radius=4.0 if self.spread else 3.0)

    def render(self, scene, image_list, seed_base=1000):
        """uint8 [n, H, W] frames for (camera, global time step) pairs; time step t uses seed seed_base + t."""
        out = np.empty((len(image_list), scene.height, scene.width), np.uint8)
        cache = {}
        for i, (c, t) in enumerate(image_list):
            if t not in cache:
                rng = np.random.default_rng(seed_base + t)
                cache = {t: scene.markers(rng, self.markers, extent=1.2 if self.spread else 0.5)}
            rng = np.random.default_rng((seed_base + t) * 64 + c)
            out[i] = scene.render(rng, cache[t], c, radius_range=(16.0, 22.0), noise_min=self.background[0],
                                  noise_max=self.background[1], salt=0.001)
        return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, arrays, frames, n_steps, threads=None):
    """The oracle (scalar C port of the reference's CPU path, gcc -O3) on `n_steps` time steps of the benchmark batch
    (frames uint8 [T, C, H, W], reused cyclically): blob extraction with one thread per camera as the reference does
    (RealtimeTracking_FLIR.py:307-312) -- or on `threads` threads (1 = the single-thread leg of BASELINE.md section 4) --
    then correspondence + DLT.  Returns the rate and every time step's results (image-point lists per camera,
    correspondence output) for the parity comparison."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle
    K, dist, R, t, F = arrays
    C = wl.cameras
    prm = oracle.default_params(undistort=True, filter_order=2)
    oracle.lib()
    T = frames.shape[0]
    pool = ThreadPoolExecutor(threads or C)
    from mocapv2_amd.engine import GRAY_SHIFT
    gray = (lambda im: oracle.bayer_gray(im, 3, GRAY_SHIFT)) if wl.bayer else (lambda im: im)
    results = []
    t0 = time.perf_counter()
    for i in range(n_steps):
        s = i % T
        lists = list(pool.map(lambda c: oracle.find_dot(gray(frames[s, c]), K[c], dist[c], params=prm), range(C)))
        P = max(1, max(len(l) for l in lists))
        pts = np.zeros((C, P, 2))
        cnt = np.zeros(C, np.int32)
        for c, l in enumerate(lists):
            cnt[c] = len(l)
            if l:
                pts[c, :len(l)] = l
        try:
            res = oracle.correspond(pts, cnt, K, dist, R, t, F)
        except RuntimeError:  # a root with more candidate groups than the oracle's limit: the reference would not finish either
            res = None
        results.append((s, lists, res))
    dt = time.perf_counter() - t0
    pool.shutdown()
    return n_steps / dt, dt, results


def parity_report(results, batches, n_cam, max_points, T):
    """The results the timed region left behind against the oracle: for every resident batch (the last pass of each, taken
    from the buffers of the lane that ran it) every time step of the CPU sample -- image points per camera (bit-exact bar)
    and 3-D points (1e-7 world units = 1e-4 mm bar).  results: (base time step, lists, oracle correspondence) of the CPU
    sample; batches: (time shift of the batch, records [T*C, rec] host, out host arrays); slot s of a batch holds base time
    step (s + shift) % T."""
    mism_images = mism_roots = n_points = gave_up = compared = 0
    sq = 0.0
    max_abs = 0.0
    pairs = [((base - shift) % T, lists, ref, records, out) for shift, records, out in batches for base, lists, ref in results if base < T]
    for s, lists, ref, records, out in pairs:
        compared += 1
        for c in range(n_cam):
            rec = records[s * n_cam + c]
            n = int(rec[0])
            got = rec[2:2 + 2 * max(0, min(n, max_points))].reshape(-1, 2).tolist()
            if n != len(lists[c]) or got != [list(p) for p in lists[c]]:
                mism_images += 1
        k = int(out["n"][s])
        if ref is None:  # the oracle gave up on this time step (candidate groups beyond its limit): the kernel must report the same
            gave_up += 1
            mism_roots += 0 if k == -2 else 1
            continue
        if k != len(ref["root"]) or (k and not np.array_equal(out["grp"][s, :k], ref["groups"])):
            mism_roots += 1
            continue
        if k:
            d = out["xyz"][s, :k] - ref["xyz"]
            sq += float((d * d).sum())
            max_abs = max(max_abs, float(np.abs(d).max()))
            n_points += k
    rmse = (sq / max(1, n_points)) ** 0.5
    return {"time_steps_compared": compared, "resident_batches_compared": len(batches), "points_compared": n_points,
            "centroid_mismatches": mism_images, "correspondence_mismatches": mism_roots,
            "time_steps_both_gave_up": gave_up,
            "rmse_3d_vs_oracle": rmse, "max_abs_3d": max_abs, "unit": "world units (m)",
            "rmse_3d_mm": rmse * 1e3, "tolerance_mm": 1e-4,
            "ok": bool(mism_images == 0 and mism_roots == 0 and max_abs < 1e-7),
            "oracle": "oracle/ C restatement (geometry half pinned to the reference, blob half parity unpinned: DESIGN.md 2)"}


def same_results(rec_a, out_a, rec_b, out_b):
    """Two passes over one batch left the same results: counts, the image points below each count, root counts and the 3-D
    points / groups / order below each root count (what lies beyond a count is whatever an earlier batch left there)."""
    import torch
    ca, cb = rec_a[:, 0], rec_b[:, 0]
    if not torch.equal(ca, cb) or not torch.equal(out_a["n"], out_b["n"]):
        return False
    col = torch.arange(rec_a.shape[1] - 2, device=rec_a.device)[None, :]
    live = col < 2 * ca.clamp(min=0)[:, None]
    if not torch.equal(rec_a[:, 2:][live], rec_b[:, 2:][live]):
        return False
    n = out_a["n"].clamp(min=0)
    roots = torch.arange(out_a["xyz"].shape[1], device=n.device)[None, :] < n[:, None]
    return all(torch.equal(out_a[k][roots], out_b[k][roots]) for k in ("xyz", "grp", "order"))


def batch_index(i, depth, n_batches):
    """Which resident batch step i takes: consecutive steps take different batches, and the order is chosen so that consecutive
    steps of one LANE (every `depth`-th step: a lane = one context with its own mask) differ too -- otherwise a lane would
    filter the same frames every time and its mask would never need clearing."""
    return (i % n_batches) if depth % n_batches else ((i + i // depth) % n_batches)


KERNELS = (("scan", "bright_cells_kernel"), ("settle", "mark_tiles_kernel + settle_tiles_kernel"), ("filter", "box_filter_kernel + filter_mask_kernel (wide tiles)"))


def roofline_of(prof, wl, n_images, n_launch_groups, traffic_key):
    """Roofline entries of the filter stage; the streaming scan (the kernel that reads every frame byte) leads.
    Durations are HIP-event averages recorded by the library on the launch stream: `avg_launch_ms` from the sequential
    pass (one batch at a time), `avg_launch_ms_in_timed_region` from the timed region itself, where `--depth` batches
    are in flight and a kernel shares the chip with its neighbours' kernels."""
    per_launch = n_images / n_launch_groups
    bytes_img = wl.width * wl.height
    traffic, source = {}, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not traffic_key:
        source = "none: PMC passes exist for the headline workload only (profiles/README.md)"
    elif not os.path.exists(tpath):
        source = "none: profiles/hbm_traffic.json is not present next to bench.py"
    else:  # HBM bytes per launch from the builder's PMC passes (profiles/README.md)
        with open(tpath) as f:
            tj = json.load(f)
        tj = tj.get("workloads", {}).get(traffic_key) or (tj if tj.get("workload_key") == traffic_key else None)
        if tj and tj.get("images_per_launch") == per_launch:
            traffic = tj.get("hbm_bytes_per_launch", {})
            source = "profiles/hbm_traffic.json (builder's rocprofv3 PMC passes on this workload; not re-measured in this run)"
        else:
            source = f"none: profiles/hbm_traffic.json holds no PMC passes for {traffic_key} at {per_launch} images per launch"
    ent = {}
    for key, name in KERNELS:
        n = prof[key + "_launches"]
        if n == 0:
            continue
        ms = prof[key + "_ms"] / n
        parts = [traffic.get(p.split(" (")[0]) for p in name.split(" + ")]  # HBM bytes of every kernel behind this timer
        ent[name] = {"kernel": name, "avg_launch_ms": round(ms, 4), "traffic": sum(parts) if parts and all(t is not None for t in parts) else None}
        if key == "scan" or len([k for k, _ in KERNELS if prof[k + "_launches"]]) == 1:
            ach = bytes_img * per_launch / (ms * 1e-3) / 1e9
            ent[name].update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(ach / HBM_PEAK_GBS, 4)})
        tr = prof.get("timed_region")
        if tr and tr[key + "_launches"]:
            ent[name]["avg_launch_ms_in_timed_region"] = round(tr[key + "_ms"] / tr[key + "_launches"], 4)
    stage_ms = sum(e["avg_launch_ms"] for e in ent.values())
    ach = bytes_img * per_launch / (stage_ms * 1e-3) / 1e9
    tr_known = [e["traffic"] for e in ent.values()]
    roof = {"bound": "hbm", "kernel": " + ".join(ent), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": sum(tr_known) if tr_known and all(t is not None for t in tr_known) else None,
            "traffic_source": source,
            "avg_launch_ms": round(stage_ms, 4), "images_per_launch": per_launch, "algorithmic_bytes_per_image": bytes_img,
            "per_kernel": ent,
            "note": "algorithmic traffic = one read of the frames (SURVEY.md 8d); the scan kernel moves those bytes (priced alone "
                    "in per_kernel), the other kernels touch the marked tiles only"}
    if prof["contour_launches"]:
        # frame -> centroid: the filter stage plus the contour kernel (north_star's 'blob-centroid kernel' as a whole)
        cms = prof["contour_ms"] / prof["contour_launches"]
        b2c = bytes_img * per_launch / ((stage_ms + cms) * 1e-3) / 1e9
        roof["blob_to_centroid"] = {"kernel": roof["kernel"] + " + contours_kernel<1> (candidates) + contour_follow_kernel (two passes) + contours_kernel<2> (tree, two passes)", "avg_launch_ms": round(stage_ms + cms, 4),
                                    "achieved": round(b2c, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b2c / HBM_PEAK_GBS, 4)}
    return roof


def kernel_ms(prof, timed_region=False):
    """mean kernel time per step (HIP events on the launch streams): of the sequential pass, or of the timed region itself, where
    `--depth` batches share the chip and every kernel is stretched by its neighbours"""
    if timed_region:
        prof = prof["timed_region"]
        n = max(1, prof["scan_launches"] or prof["filter_launches"])
    else:
        n = prof["steps"]
    return {"scan": round(prof["scan_ms"] / n, 4), "settle": round(prof["settle_ms"] / n, 4), "filter": round(prof["filter_ms"] / n, 4),
            "contours": round(prof["contour_ms"] / n, 4), "correspond": round(prof["corr_ms"] / n, 4)}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    from mocapv2_amd.pipeline import BatchTracker, check_status, scene_arrays

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    W, H = (int(v) for v in args.frame.lower().split("x"))
    bg = tuple(int(v) for v in args.background.split("-"))
    main_wl = Workload(args.cameras, W, H, args.markers, args.dist, bg, args.bayer)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def batch_of(tracker, n_batches):
        return batch_index(tracker._k, len(tracker.lanes), n_batches)

    def timed(tracker, batches, steps, warmup):
        """W untimed + K timed steps rotating through the resident batches, barrier + synchronize on both sides, max over
        ranks.  Then a short sequential pass (one batch at a time, a synchronize after each) for per-kernel durations that
        are not stretched by the neighbouring batches' kernels (in the timed region `--depth` batches share the chip).
        Returns the last output, the elapsed time, the profile, and what the timed region left in every lane's buffers:
        (batch index, records, outputs) of the last batch each lane ran."""
        nb = len(batches)
        lane_batch = {}

        def one():
            k = batch_of(tracker, nb)
            lane_batch[tracker._k % len(tracker.lanes)] = k
            return tracker.step(batches[k])
        for _ in range(warmup):
            one()
        tracker.synchronize()
        barrier()
        tracker.profile(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = one()
        tracker.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        tracker.profile(False)
        prof_timed = tracker.profile_read()
        left = [(k, tracker.lanes[l].records.clone(), {key: v.clone() for key, v in tracker.lanes[l].out.items()})
                for l, k in sorted(lane_batch.items()) if tracker.lanes[l].out is not None]
        n_seq = max(1, min(steps, 6))
        tracker.profile(True)
        for _ in range(n_seq):
            out = one()
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
        return out, elapsed, prof, left

    def run_each(tracker, batches):
        """every resident batch once more, one at a time (not timed): (batch index, records, outputs)"""
        res = []
        for k, b in enumerate(batches):
            out = tracker.step(b)
            tracker.synchronize()
            res.append((k, tracker.records.clone(), {key: v.clone() for key, v in out.items()}))
        return res

    def measure(wl, time_steps, steps, warmup, n_batches=None):
        """Build the scene / tracker / resident batches of one workload and time `steps` steps of it.  Batch k holds, in
        image slot (camera c, time step t), the frame of time step (t + k * shift) % T_total of the rendered sequence: the
        same markers seen at other moments, so every slot's content changes from batch to batch."""
        n_batches = max(1, n_batches or args.batches)
        scene = wl.scene()
        arrays = scene_arrays(scene)
        tracker = BatchTracker(*arrays, wl.width, wl.height, time_steps, world=world, rank=rank, device=local_rank, depth=args.depth,
                               max_points=wl.max_points, max_groups=wl.max_groups, bayer_pattern=3 if wl.bayer else None,
                               collective=args.collective)
        images = tracker.local_image_list()
        frames_host = wl.render(scene, images)
        t_total = time_steps * world
        shift = max(1, t_total // n_batches + 1) if n_batches > 1 else 0
        row = {key: i for i, key in enumerate(images)}
        batches = []
        for k in range(n_batches):
            want = [(c, (t + k * shift) % t_total) for c, t in images]
            idx = [row.get(key) for key in want]
            if k == 0:
                host = frames_host
            elif all(i is not None for i in idx):
                host = frames_host[np.asarray(idx)]
            else:  # a rank that holds only part of a camera's time steps: render the shifted moments
                host = wl.render(scene, want)
            batches.append(torch.from_numpy(host).pin_memory() if args.from_host else torch.from_numpy(host).cuda())
            del host
        torch.cuda.synchronize()
        out, elapsed, prof, left = timed(tracker, batches, steps, warmup)
        return {"scene": scene, "arrays": arrays, "tracker": tracker, "images": images, "frames_host": frames_host, "batches": batches,
                "out": out, "elapsed": elapsed, "prof": prof, "T": time_steps, "left": left, "shift": shift, "n_batches": n_batches}

    def status_of(m):
        """no capacity code in any time step / image of any batch the timed region left behind"""
        ok = True
        for _, rec, o in m["left"]:
            try:
                check_status(o["n"].cpu().numpy())
                ok = ok and bool((rec[:, 0].cpu().numpy() >= 0).all())
            except RuntimeError:
                ok = False
        return ok, m["out"]["n"].cpu().numpy()

    def section(wl, m, steps, traffic_key=None):
        ok, n_roots = status_of(m)
        el, prof, T = m["elapsed"], m["prof"], m["T"]
        return {"workload": wl.name(world), "value": round(T * world * steps / el, 2), "unit": "frames/s", "ms_per_step": round(1e3 * el / steps, 4),
                "time_steps_per_step": T, "steps": steps, "images_per_step": len(m["images"]),
                "roofline": roofline_of(prof, wl, len(m["images"]), 1 if world == 1 else len(m["tracker"].segs), traffic_key),
                "kernel_ms_per_step": kernel_ms(prof), "kernel_ms_per_step_in_timed_region": kernel_ms(prof, True), "status_ok": ok, "points_per_frame": float(np.maximum(n_roots, 0).mean()),
                "dark_tile_early_out": {"tiles_per_step": prof["tiles"], "tiles_resolved_without_filtering": prof["tiles_skipped"]}}

    def ba_residual_section(n_eval=50):
        """The residual evaluation part of BASELINE.json configs[4]: 16 cameras x 64 markers, one bundle-adjustment residual
        vector = triangulate every point from its 16 views + reproject into all of them + per-point MSE (reference
        lib/Helpers.py:161-167), through the drop-in lib.Helpers surface (host lists in, host array out: the H2D / D2H
        copies and the packing are inside the figure), with the oracle's C restatement timed beside it."""
        import oracle
        import mocapv2_amd.lib.Helpers as Hh
        from mocapv2_amd.synth import MILD_DIST, Scene
        from scipy.spatial.transform import Rotation
        C, M = 16, 64
        sc = Scene(C, 3840, 2160, dist=MILD_DIST, radius=4.0)
        rng = np.random.default_rng(5)
        cents = sc.centroids(sc.markers(rng, M, extent=1.2), rng, jitter=0.3)
        groups = np.stack([np.stack([cents[c][m] for c in range(C)]) for m in range(M)]).astype(float)  # [M, C, 2]
        K, dd = np.stack([sc.K] * C), np.stack([sc.dist] * C)
        R, tt = np.stack([p["R"] for p in sc.poses]), np.stack([p["t"] for p in sc.poses])
        params = []
        for c in range(1, C):
            Rrel = R[c] @ R[0].T
            params += list(Rotation.from_matrix(Rrel).as_rotvec()) + list(tt[c] - Rrel @ tt[0])
        params = np.array(params)
        Hh.camera_params = np.array([{"intrinsic_matrix": K[i].tolist(), "distortion_coef": dd[i].tolist()} for i in range(C)])
        glist = groups.tolist()

        def one():
            poses = Hh.params_to_camera_poses(params, C)
            obj = Hh.triangulate_points(glist, poses)
            return Hh.calculate_reprojection_errors(glist, obj, poses).astype(np.float32)
        got = one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_eval):
            got = one()
        dt = (time.perf_counter() - t0) / n_eval
        valid = np.ones(groups.shape[:2], np.uint8)
        t0 = time.perf_counter()
        for _ in range(n_eval):
            exp = oracle.ba_residuals(params, C, groups, valid, K, dd)
        dt_cpu = (time.perf_counter() - t0) / n_eval
        # the resident form (mocap_ba_residuals): image points uploaded once, one launch + one stream wait per evaluation
        from mocapv2_amd.engine import default_context
        ctx = default_context()
        ctx.set_cameras(K, dd, R, tt)
        prob = ctx.ba_problem(groups, valid)
        res = prob.residuals(params)
        n_res = 20 * n_eval
        t0 = time.perf_counter()
        for _ in range(n_res):
            res = prob.residuals(params)
        dt_res = (time.perf_counter() - t0) / n_res
        jac_sets = np.tile(params, (1 + len(params), 1))  # a forward-difference Jacobian: 1 + 6 (C - 1) = 91 vectors, one launch
        jac_sets[1:] += np.eye(len(params)) * 3.45e-4
        prob.residuals(jac_sets)
        t0 = time.perf_counter()
        for _ in range(n_eval):
            prob.residuals(jac_sets)
        dt_jac = (time.perf_counter() - t0) / n_eval
        return {"workload": "bundle-adjustment residual vector, 16 cameras x 64 markers (BASELINE.json configs[4], residual part)",
                "ms_per_evaluation": round(1e3 * dt_res, 4), "evaluations_per_s": round(1 / dt_res, 1), "points_per_s": round(M / dt_res, 1),
                "includes": "mocap_ba_residuals: image points resident, parameters in / residuals out through one pinned block, "
                            "rotvec -> R + triangulation + reprojection + float32 cast in one launch, one stream wait",
                "matches_oracle": bool(res.shape == exp.shape and np.allclose(res, exp, rtol=2e-5, atol=1e-6)),
                "jacobian_91_vectors_ms": round(1e3 * dt_jac, 4),
                "drop_in_lists_ms_per_evaluation": round(1e3 * dt, 4),
                "drop_in_includes": "lib.Helpers params_to_camera_poses + triangulate_points + calculate_reprojection_errors on host lists: "
                                    "list packing, H2D, two kernels, D2H per call (what round 2 reported)",
                "drop_in_matches_oracle": bool(got.shape == exp.shape and np.allclose(got, exp, rtol=2e-5, atol=1e-6)),
                "cpu_oracle_ms_per_evaluation": round(1e3 * dt_cpu, 4)}


    def oracle_time_step(wl, arrays, frames_c):
        """one time step through the oracle (checker only): image-point lists per camera + correspondence result (or None)"""
        import oracle
        K, dist_, R, t, F = arrays
        lists = [oracle.find_dot(frames_c[c], K[c], dist_[c]) for c in range(wl.cameras)]
        P = max(1, max(len(l) for l in lists))
        pts = np.zeros((wl.cameras, P, 2))
        cnt = np.zeros(wl.cameras, np.int32)
        for c, l in enumerate(lists):
            cnt[c] = len(l)
            if l:
                pts[c, :len(l)] = l
        try:
            return lists, oracle.correspond(pts, cnt, K, dist_, R, t, F, max_groups=wl.max_groups)
        except RuntimeError:
            return lists, None

    def latency_section(wl, m, n_rep=200):
        """ONE time step (T = 1, one batch at a time) through the whole path, frames resident: the latency a live tracker would see
        per frame set (reference: one pass of track_points + track, RealtimeTracking_FLIR.py:95-143,157-209)."""
        frames_tc = m["frames_host"].reshape(m["T"], wl.cameras, wl.height, wl.width)
        trk = BatchTracker(*m["arrays"], wl.width, wl.height, 1, depth=1, max_points=wl.max_points, max_groups=wl.max_groups)
        dev = torch.from_numpy(np.ascontiguousarray(frames_tc[0])).cuda()
        for _ in range(5):
            out1 = trk.step(dev)
        trk.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_rep):
            out1 = trk.step(dev)
            trk.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n_rep
        lists, ref = oracle_time_step(wl, m["arrays"], frames_tc[0])
        rec = trk.records.cpu().numpy()
        n = int(out1["n"].cpu()[0])
        ok = all(int(rec[c, 0]) == len(lists[c]) and rec[c, 2:2 + 2 * len(lists[c])].reshape(-1, 2).tolist() == [list(p) for p in lists[c]]
                 for c in range(wl.cameras))
        ok = ok and ref is not None and n == len(ref["root"]) and (n == 0 or float(np.abs(out1["xyz"][0, :n].cpu().numpy() - ref["xyz"]).max()) < 1e-7)
        return {"workload": f"one time step of {wl.cameras} cameras, {wl.width}x{wl.height}, frames resident, launch to synchronised result",
                "latency_one_time_step_ms": round(ms, 4), "repetitions": n_rep, "points": n, "matches_oracle": bool(ok)}

    def replay_section(wl, m, runs=4):
        """N3: the headless replay tracker (mocapv2_amd/replay.py = track_points + track, RealtimeTracking_FLIR.py:95-143,157-209)
        over device-resident frames: the resident batches as one recording, batches of the headline size, three in flight, INCLUDING
        the read-back of every batch, the obj_count + 1 selection (lib/Helpers.py:274-279) and one msgpack message per time step
        (RealtimeTracking_FLIR.py:183-188)."""
        import oracle
        from mocapv2_amd.replay import OBJ_COUNT, ReplayTracker, tracker_message
        T = m["T"]
        shape = (T, wl.cameras, wl.height, wl.width)
        recording = [b.reshape(shape) for b in m["batches"]] * runs
        rp = ReplayTracker(*m["arrays"], wl.width, wl.height, batch=T, max_points=wl.max_points, max_groups=wl.max_groups, depth=3)
        first = list(rp.run(recording[0]))  # warm-up + the results that are checked
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_steps = 0
        for res in rp.run_batches(recording):
            n_steps += res["n_steps"]
        dt_bulk = time.perf_counter() - t0
        t0 = time.perf_counter()
        n_gen = 0
        for _res in rp.run(recording):
            n_gen += 1
        dt_gen = time.perf_counter() - t0
        # the first time steps against the oracle: object points, image points, message bytes
        frames_tc = m["frames_host"].reshape(shape)
        ok, point = True, [0] * 8
        for s in range(2):
            _, ref = oracle_time_step(wl, m["arrays"], frames_tc[s])
            obj = oracle.select_objects(ref, OBJ_COUNT)[0] if ref is not None and len(ref["root"]) else np.array([])
            if len(obj):
                point = [0, 0, 0, 0] + list(obj[0])
            got = first[s]
            ok = ok and len(got["object_points"]) == len(obj) and (len(obj) == 0 or float(np.abs(got["object_points"] - obj).max()) < 1e-7)
            ok = ok and (ref is None or len(ref["root"]) == 0 or np.array_equal(got["image_points"], ref["groups"].astype(np.int64)))
            ok = ok and got["message"][:15] == tracker_message(point)[:15] and len(got["message"]) == len(tracker_message(point))
        return {"workload": f"ReplayTracker(batch={T}, depth=3) over {len(recording)} x {T} time steps of the headline frames, device-resident",
                "value": round(n_steps / dt_bulk, 1), "unit": "time steps/s", "form": "run_batches: results and messages per batch, built in bulk",
                "per_time_step_generator": {"value": round(n_gen / dt_gen, 1), "unit": "time steps/s",
                                            "form": "run: one Python dict per time step (the generator itself is the bound)"},
                "includes": "device-to-host copy of every batch's outputs, obj_count + 1 selection, one msgpack message per time step",
                "matches_oracle": bool(ok)}

    def from_host_section(wl, m, steps=6):
        """The PCIe-inclusive rate: the batch stays in pinned host memory and every step uploads it (never the headline `value`)."""
        host = torch.from_numpy(m["frames_host"]).pin_memory()
        trk = m["tracker"]
        for _ in range(2):
            trk.step(host)
        trk.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out_h = trk.step(host)
        trk.synchronize()
        dt = time.perf_counter() - t0
        ref_out = trk.step(m["batches"][0])
        trk.synchronize()
        same = bool(torch.equal(out_h["n"], ref_out["n"]))
        nbytes = host.numel()
        del host
        return {"workload": "the headline batch uploaded from pinned host memory every step (PCIe-inclusive)", "value": round(m["T"] * steps / dt, 1),
                "unit": "frames/s", "ms_per_step": round(1e3 * dt / steps, 3), "h2d_GBps": round(nbytes * steps / dt / 1e9, 2),
                "same_results_as_resident": same}

    T_STEPS = args.time_steps or main_wl.default_time_steps()
    m = measure(main_wl, T_STEPS, args.steps, args.warmup)
    tracker, out, elapsed, prof = m["tracker"], m["out"], m["elapsed"], m["prof"]
    status_ok, n_roots = status_of(m)
    if main_wl.spread:
        status_ok = bool((tracker.records[:, 0].cpu().numpy() >= 0).all())  # crowded rig: given-up time steps are counted, not an error

    if rank == 0:
        value = T_STEPS * world * args.steps / elapsed
        std = (main_wl.cameras, main_wl.width, main_wl.height, main_wl.markers, main_wl.background, main_wl.bayer) == \
            (6, 1920, 1080, 8, (0, 60), False)
        roof = roofline_of(prof, main_wl, len(m["images"]), 1 if world == 1 else len(tracker.segs),
                           f"6x1920x1080-m8-{args.dist}" if std and world == 1 else None)
        res = f"{main_wl.width}x{main_wl.height}"
        line = {
            "metric": f"frames/sec ({main_wl.cameras}-cam {'1080p' if res == '1920x1080' else res})", "value": round(value, 2),
            "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8+int64+f64",
            "data": "synthetic",
            "config": {"workload": main_wl.name(world),
                       "cameras": main_wl.cameras, "width": main_wl.width, "height": main_wl.height, "markers": main_wl.markers,
                       "time_steps_per_step_per_gpu": T_STEPS, "distortion": args.dist, "background": list(main_wl.background),
                       "marker_cube_m": 2.4 if main_wl.spread else 1.0, "camera_ring_radius_m": 4.0 if main_wl.spread else 3.0,
                       "rig_note": ("64 markers are spread over a 2.4 m cube seen from a 4 m ring: in the default 1 m cube the reference's "
                                    "cartesian expansion (lib/Helpers.py:239-245) exceeds 4 M groups per root in half the time steps, which "
                                    "oracle and kernel both report as MOCAP_CORR_E_GROUPS (time_steps_without_result counts those left)")
                       if main_wl.spread else None,
                       "frames_resident_in_hbm": not args.from_host, "batches_in_flight": args.depth,
                       "resident_batches": m["n_batches"],
                       "batch_rotation": f"slot (camera, t) of batch k holds the frame of time step (t + {m['shift']} k) mod {T_STEPS * world}: "
                                         "every image slot changes from one step to the next",
                       "status_ok": status_ok, "parity_ok": None,
                       "input": "raw Bayer GR frames, gray conversion inside every step" if args.bayer else "gray frames",
                       "parallelism": "single launch, time-major" if world == 1 else f"camera-major blocks x{world} + 1 all-gather per batch",
                       "collective": None if world == 1 else
                       ("mocap_allgather_centroids (ncclAllGather behind the C-ABI, RCCL)" if tracker.collective == "rccl"
                        else ("torch.distributed over gloo (one-GPU rehearsal)" if args.rehearse_on_one_gpu else "torch.distributed all_gather_into_tensor (RCCL)"))},
            "roofline": roof,
            "kernel_ms_per_step": kernel_ms(prof),
            "kernel_ms_per_step_in_timed_region": kernel_ms(prof, True),
            "status_ok": status_ok,
            "time_steps_without_result": int((n_roots < 0).sum()),  # capacity codes (MOCAP_CORR_E_*): e.g. the reference's cartesian
                                                                    # expansion beyond max_groups in a crowded rig (configs[4])
            "points_per_frame": float(np.maximum(n_roots, 0).mean()),
            "dark_tile_early_out": {"tiles_per_step": prof["tiles"], "tiles_resolved_without_filtering": prof["tiles_skipped"],
                                    "note": "exact: the scan kernel reads the frames once and sums every 8x8 cell's excess over a base "
                                            "derived from the threshold; a filter tile whose cells prove that no mask bit can be set is "
                                            "answered with zeros (DESIGN.md 4.1); disabled run in without_early_out"},
        }
        if args.rehearse_on_one_gpu:
            line["rehearsal"] = "N ranks on one GPU over gloo: a functional check of the multi-rank path, NOT a measurement"
        if world == 1 and args.extra and std and not args.from_host:
            # figures that only existed in DESIGN.md, now in the driver-recorded line
            pipe = {}
            n_ss = 300
            _, el_s, prof_s, _ = timed(tracker, m["batches"], n_ss, args.warmup)
            pipe["steady_state"] = {"workload": main_wl.name(world), "steps": n_ss, "value": round(T_STEPS * n_ss / el_s, 2), "unit": "frames/s",
                                    "ms_per_step": round(1e3 * el_s / n_ss, 4),
                                    "kernel_ms_per_step_in_timed_region": kernel_ms(prof_s, True),
                                    "note": "the headline's timed region with 300 steps instead of --steps: filling and draining the lanes no longer counts"}
            pipe["latency"] = latency_section(main_wl, m)
            pipe["replay_tracker"] = replay_section(main_wl, m)
            pipe["from_host"] = from_host_section(main_wl, m)
            line["pipeline"] = pipe
        left = m["left"]  # what the timed region left in the lanes' buffers: (batch, records, outputs)
        if world == 1 and args.secondary:
            # the same batches with the early-out off: every tile runs the full filter (dense kernel); the results must not change
            for lane in tracker.lanes:
                lane.ctx.set_tuning("skip_dark", 0)
            n_d = max(3, args.steps // 3)
            out_d, el_d, prof_d, left_d = timed(tracker, m["batches"], n_d, 2)
            each_d = run_each(tracker, m["batches"])
            for lane in tracker.lanes:
                lane.ctx.set_tuning("skip_dark", 1)
            each_s = run_each(tracker, m["batches"])
            roof_d = roofline_of(prof_d, main_wl, len(m["images"]), 1, None)
            same = all(ka == kb and same_results(ra, oa, rb, ob) for (ka, ra, oa), (kb, rb, ob) in zip(each_d, each_s))  # every batch, both roads
            del each_d, each_s
            line["without_early_out"] = {"value": round(T_STEPS * n_d / el_d, 2), "unit": "frames/s",
                                         "ms_per_step": round(1e3 * el_d / n_d, 4),
                                         "filter_avg_launch_ms": roof_d["avg_launch_ms"], "roofline_frac": roof_d["frac"],
                                         "same_results": bool(same)}
            del left_d
        if world == 1 and args.cpu_steps > 0:
            frames_tc = m["frames_host"].reshape(T_STEPS, main_wl.cameras, main_wl.height, main_wl.width)
            fps, dt, results = cpu_baseline(main_wl, m["arrays"], frames_tc, args.cpu_steps)
            line["cpu_baseline"] = {"value": round(fps, 3), "unit": "frames/s", "cores": main_wl.cameras, "kind": "port",
                                    "sample": f"{args.cpu_steps} time steps x {main_wl.cameras} cameras of the same workload, "
                                              f"{dt:.1f} s, thread-per-camera C oracle (gcc -O3), {os.cpu_count()} host cpus: {cpu_model()}"}
            if args.cpu_single_steps > 0:
                fps1, dt1, _ = cpu_baseline(main_wl, m["arrays"], frames_tc, args.cpu_single_steps, threads=1)
                line["cpu_baseline"]["single_thread"] = {"value": round(fps1, 3), "unit": "frames/s", "cores": 1,
                                                         "sample": f"{args.cpu_single_steps} time steps x {main_wl.cameras} cameras, {dt1:.1f} s, one thread"}
            # parity of what the timed region left behind against the oracle (not timed): every time step of the CPU sample, in
            # every resident batch
            # (what the lanes hold after the timed region -- the last batch each of them ran -- and, so that every resident batch is
            # covered whatever the step count, one more pass over each batch)
            batches_host = [(k * m["shift"], rec.cpu().numpy(), {key: v.cpu().numpy() for key, v in o.items()})
                            for k, rec, o in left + run_each(tracker, m["batches"])]
            line["parity"] = parity_report(results, batches_host, main_wl.cameras, main_wl.max_points, T_STEPS)
            line["config"]["parity_ok"] = line["parity"]["ok"]
            line["parity_spot_check"] = line["parity"]["ok"]
            del batches_host
        # release the main workload before the next ones are built
        del left
        m.clear()
        del tracker, out
        torch.cuda.empty_cache()
        if world == 1 and args.secondary:
            other = "zero" if args.dist == "mild" else "mild"
            wl2 = Workload(main_wl.cameras, main_wl.width, main_wl.height, main_wl.markers, other, main_wl.background, main_wl.bayer)
            m2 = measure(wl2, T_STEPS, args.steps, args.warmup)
            sec = section(wl2, m2, args.steps)
            sec["distortion"] = other
            sec["note"] = ("same workload with zero lens distortion: cv.undistort is then the identity map and the kernels read "
                           "the frame without the remap gather (SURVEY.md section 8d lists both variants)")
            line["other_distortion_variant"] = sec
            m2.clear()
            torch.cuda.empty_cache()
        if world == 1 and args.extra and std:
            extra = {}
            for key, wl in (("markers32_configs2", Workload(6, 1920, 1080, 32, args.dist)),
                            ("bright_background", Workload(6, 1920, 1080, 8, args.dist, (90, 110))),
                            ("frame_4k", Workload(6, 3840, 2160, 8, args.dist)),
                            # N1: the same frames taken as raw Bayer GR sensor frames, the gray conversion fused with the scan inside every step
                            ("bayer_input", Workload(6, 1920, 1080, 8, args.dist, bayer=True))):
                Tx = wl.default_time_steps()
                mx = measure(wl, Tx, args.steps, args.warmup)
                extra[key] = section(wl, mx, args.steps, f"6x1920x1080-m32-{args.dist}" if key == "markers32_configs2" else None)
                mx.clear()
                torch.cuda.empty_cache()
            extra["ba_residual_eval_configs4"] = ba_residual_section()
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
