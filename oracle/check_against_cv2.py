#!/usr/bin/env python3
"""Pins the blob half of the oracle (oracle/blob_oracle.c, "parity unpinned": DESIGN.md section 2) to a real OpenCV
wherever one can be imported:

    python oracle/check_against_cv2.py [--write]

For a set of seeded frames and three lens models it runs every orc_* restatement of an OpenCV function next to the
cv2 call the reference makes (lib/ImageOperations.py:19-20,29-30,38,41-65; RealtimeTracking_FLIR.py:103-104) and
compares bit for bit:
    cv.undistort           <-> oracle.undistort
    cv.threshold           <-> oracle.threshold
    cv.medianBlur(., 5)    <-> oracle.median5
    cv.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE): number, order and points of the contours, the parent links
                           <-> oracle.find_contours
    cv.contourArea / cv.arcLength / cv.moments per contour <-> the oracle's area / perimeter / a00, a10, a01 / cx, cy
    cv.cvtColor(BayerXX2BGR) + cv.cvtColor(BGR2GRAY)       <-> oracle.bayer_gray with shift 14 and with shift 15
and reports which gray_shift the installed OpenCV uses (mocapv2_amd.engine.GRAY_SHIFT should equal it).  With --write
the inputs and the cv2 outputs go to tests/golden/blob_cv2_*.npz, which tests/test_oracle_blob.py then holds the C
oracle to on machines without cv2 (fixtures are data: arrays only).

In the build container there is no cv2 (no wheel, no network): the script says so and exits 0.  Test infrastructure:
nothing under mocapv2_amd/ imports this.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def frames(seed, H, W, n_discs=6):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 61, (H, W), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(n_discs):
        u, v, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(6, 24)
        d = np.sqrt((xx - u) ** 2 + (yy - v) ** 2)
        img = np.maximum(img, (np.clip((r + 0.75 - d) / 1.5, 0, 1) * 255).astype(np.uint8))
        if rng.random() < 0.4:  # a ring: hole borders, nesting
            img[d < 0.45 * r] = rng.integers(0, 61)
    img[rng.integers(0, H, H * W // 1000), rng.integers(0, W, H * W // 1000)] = 255
    return img


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="write tests/golden/blob_cv2_*.npz")
    args = ap.parse_args()
    try:
        import cv2 as cv
    except Exception as e:  # noqa: BLE001
        print(f"cv2 is not importable here ({type(e).__name__}: {e}); nothing checked, blob oracle stays 'parity unpinned'")
        return 0
    import oracle
    from mocapv2_amd.synth import MILD_DIST, intrinsics
    print("OpenCV", cv.__version__)
    bad = []

    def check(name, ok, detail=""):
        print(("ok   " if ok else "FAIL ") + name + (" " + detail if detail and not ok else ""))
        if not ok:
            bad.append(name)

    shift_votes = {14: 0, 15: 0}
    for case, (W, H, scale) in enumerate([(640, 360, 1.0), (251, 77, 4.0), (500, 300, -3.0), (960, 540, 0.0)]):
        K = intrinsics(W, H)
        dist = np.array(MILD_DIST) * scale
        img = frames(100 + case, H, W)
        und = cv.undistort(img, K, dist)
        check(f"undistort {W}x{H} x{scale}", np.array_equal(und, oracle.undistort(img, K, dist)))
        _, thr = cv.threshold(und, 255 * 0.85, 255, cv.THRESH_BINARY)
        check(f"threshold {W}x{H}", np.array_equal(thr, oracle.threshold(und)))
        med = cv.medianBlur(und, 5)
        check(f"medianBlur {W}x{H}", np.array_equal(med, oracle.median5(und)))
        # the filter chain of image_filter_gpu with the numba blur restated (floor of the in-bounds mean)
        blur = oracle.box_blur(und)
        _, t2 = cv.threshold(blur, 255 * 0.85, 255, cv.THRESH_BINARY)
        mask = cv.medianBlur(t2, 5)
        check(f"filter chain {W}x{H}", np.array_equal(mask, oracle.image_filter(und, 0)))
        contours, hierarchy = cv.findContours(mask, cv.RETR_TREE, cv.CHAIN_APPROX_SIMPLE)
        mine = oracle.find_contours(mask, with_points=True)
        same = len(contours) == len(mine)
        check(f"findContours count {W}x{H}", same, f"{len(contours)} vs {len(mine)}")
        rows = []
        if same:
            pts_ok = all(np.array_equal(c.reshape(-1, 2), m["points"]) for c, m in zip(contours, mine))
            check(f"findContours order + points {W}x{H}", pts_ok)
            meas_ok = True
            for c, m in zip(contours, mine):
                area, per, mo = cv.contourArea(c), cv.arcLength(c, True), cv.moments(c)
                meas_ok &= area == m["area"] and per == m["perimeter"]
                if mo["m00"] != 0:
                    meas_ok &= int(mo["m10"] / mo["m00"]) == m["cx"] or not m["kept"]
                    meas_ok &= int(mo["m01"] / mo["m00"]) == m["cy"] or not m["kept"]
                rows.append([area, per, mo["m00"], mo["m10"], mo["m01"]])
            check(f"contourArea / arcLength / moments {W}x{H}", bool(meas_ok))
            par = hierarchy[0][:, 3] if hierarchy is not None else np.zeros(0, int)
            check(f"hierarchy parents {W}x{H}", np.array_equal(par, np.array([m["parent_order"] for m in mine], dtype=par.dtype)))
        bay = frames(200 + case, H, W)
        grays = {}
        for pat, code in ((0, cv.COLOR_BayerBG2BGR), (1, cv.COLOR_BayerGB2BGR), (2, cv.COLOR_BayerRG2BGR), (3, cv.COLOR_BayerGR2BGR)):
            g = cv.cvtColor(cv.cvtColor(bay, code), cv.COLOR_BGR2GRAY)
            grays[pat] = g
            m14, m15 = np.array_equal(g, oracle.bayer_gray(bay, pat, 14)), np.array_equal(g, oracle.bayer_gray(bay, pat, 15))
            shift_votes[14] += m14
            shift_votes[15] += m15
            check(f"Bayer pattern {pat} -> gray {W}x{H}", m14 or m15, "neither the 14-bit nor the 15-bit luma matches")
        if args.write:
            out = os.path.join(ROOT, "tests", "golden", f"blob_cv2_{case}.npz")
            np.savez_compressed(out, cv_version=cv.__version__, K=K, dist=dist, img=img, undistorted=und, threshold=thr, median=med,
                                mask=mask, n_contours=len(contours),
                                contour_points=np.concatenate([c.reshape(-1, 2) for c in contours]) if contours else np.zeros((0, 2), int),
                                contour_sizes=np.array([len(c) for c in contours]), measures=np.array(rows),
                                parents=hierarchy[0][:, 3] if hierarchy is not None else np.zeros(0, int),
                                bayer=bay, **{f"gray_pattern{p}": g for p, g in grays.items()},
                                provenance="cv2 outputs on seeded frames, written by oracle/check_against_cv2.py")
            print("wrote", out)
    shift = 15 if shift_votes[15] >= shift_votes[14] else 14
    from mocapv2_amd.engine import GRAY_SHIFT
    print(f"BGR2GRAY fixed point of this OpenCV: shift {shift} (votes {shift_votes}); mocapv2_amd.engine.GRAY_SHIFT = {GRAY_SHIFT}")
    if shift != GRAY_SHIFT:
        bad.append("GRAY_SHIFT default")
    print("ALL OK: the blob oracle is pinned to this OpenCV" if not bad else f"{len(bad)} mismatches: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
