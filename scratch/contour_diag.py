"""Re-runs the first failing case of test_contours_match_oracle_large_masks with both forms of the contour stage and prints the borders
whose measurements differ from the oracle's."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from gpu_util import pack_mask
from test_gpu_blob import structured_mask
from mocapv2_amd.engine import MocapContext
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(7000 + seed)
H, W = [(300, 420), (257, 513), (480, 300)][seed % 3]
masks = []
for i in range(12):
    m = structured_mask(rng, H, W, n=rng.integers(3, 14)) != 0
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(int(rng.integers(0, 4))):
        x0, y0, x1, y1 = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(0, W), rng.uniform(0, H)
        t = ((xx - x0) * (x1 - x0) + (yy - y0) * (y1 - y0)) / max(1e-9, (x1 - x0) ** 2 + (y1 - y0) ** 2)
        d = np.hypot(xx - (x0 + t * (x1 - x0)), yy - (y0 + t * (y1 - y0)))
        m |= (d <= rng.choice([0.5, 0.8, 1.2])) & (t >= 0) & (t <= 1)
    if i % 4 == 3:
        m ^= rng.random((H, W)) < 0.002
    masks.append((m * 255).astype(np.uint8))
for split in (1, 0):
    ctx = MocapContext(W, H)
    ctx.set_tuning("contours_split", split)
    ctx.set_blob_params(min_area=20.0, min_circ=0.05)
    xy, cnt, recs = ctx.contours_from_mask(pack_mask(np.stack(masks)), max_blobs=128, debug_cap=384)
    nbad = 0
    for i, m in enumerate(masks):
        table = oracle.find_contours(m, min_area=20.0, min_circ=0.05, with_points=True)
        for c in table:
            c["key"] = c["oy"] * (W + 1) + c["ox"] + (1 if c["is_hole"] else 0)
        by_key = {(r["key"], r["is_hole"]): r for r in recs[i]}
        for c in table:
            r = by_key.get((c["key"], c["is_hole"]))
            if r is None:
                print("split", split, "image", i, "missing border", c["key"]); nbad += 1; continue
            if r["perimeter"] != c["perimeter"] or r["npts"] != c["npts"] or r["a00"] != c["a00"] or r["steps"] != c["steps"]:
                nbad += 1
                pts = c["points"]
                d = np.diff(np.vstack([pts, pts[:1]]), axis=0)
                runs = [(int(abs(a) if a else abs(b)), bool(a and b)) for a, b in d]
                print("split", split, "image", i, "start", (c["ox"], c["oy"]), "hole", c["is_hole"], "steps", c["steps"], r["steps"], "npts", c["npts"], r["npts"],
                      "per", repr(c["perimeter"]), repr(r["perimeter"]), "diff", r["perimeter"] - c["perimeter"])
                print("   first runs (len, diag):", runs[:6], "... last:", runs[-6:])
    print("split", split, "bad borders", nbad)
