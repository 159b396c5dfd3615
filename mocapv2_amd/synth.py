"""Seeded synthetic scenes for tests and bench.

Ring of cameras looking at the origin, world->camera poses (x_c = R X + t, the convention of
reference lib/Helpers.py:60), fundamental matrices F(0->i) built from the poses with the formula of
reference CalculateCameraPoses.py:46-73 (x_i^T F x_0 = 0), markers rendered as anti-aliased discs on
a noisy dark background.  Pure NumPy; nothing here runs on the GPU.
"""
import numpy as np

MILD_DIST = (-0.10, 0.02, 1e-3, 1e-3, 0.0)
ZERO_DIST = (0.0, 0.0, 0.0, 0.0, 0.0)


def intrinsics(width, height):
    f = 1400.0 * width / 1920.0
    return np.array([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]])


def ring_cameras(n_cam, radius=3.0, height=1.5, phase=0.1):
    """World->camera poses for n_cam cameras on a ring, all looking at the origin."""
    poses = []
    for i in range(n_cam):
        ang = phase + 2.0 * np.pi * i / n_cam
        c = np.array([radius * np.cos(ang), radius * np.sin(ang), height])
        fwd = -c / np.linalg.norm(c)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])
        poses.append({"R": R, "t": -R @ c})
    return poses


def fundamental_from_poses(pose_a, pose_b, K_a, K_b):
    """F with x_b^T F x_a = 0 (pixels), from two world->camera poses."""
    Ra, ta = np.asarray(pose_a["R"], float), np.asarray(pose_a["t"], float).reshape(3)
    Rb, tb = np.asarray(pose_b["R"], float), np.asarray(pose_b["t"], float).reshape(3)
    R = Rb @ Ra.T
    t = tb - R @ ta
    tx = np.array([[0.0, -t[2], t[1]], [t[2], 0.0, -t[0]], [-t[1], t[0], 0.0]])
    return np.linalg.inv(K_b).T @ (tx @ R) @ np.linalg.inv(K_a)


def project(points, pose, K, dist):
    """Pinhole + Brown distortion projection of world points [N,3] -> pixels [N,2] (float64)."""
    P = np.asarray(points, float)
    pc = P @ np.asarray(pose["R"], float).T + np.asarray(pose["t"], float).reshape(3)
    x, y = pc[:, 0] / pc[:, 2], pc[:, 1] / pc[:, 2]
    k1, k2, p1, p2, k3 = dist
    r2 = x * x + y * y
    cd = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], axis=1)


class Scene:
    """n_cam ring cameras sharing one K / distortion, with F(0->i) for i = 1..n_cam-1."""

    def __init__(self, n_cam, width=1920, height=1080, dist=ZERO_DIST, radius=3.0):
        self.n_cam, self.width, self.height = n_cam, width, height
        self.K = intrinsics(width, height)
        self.dist = np.array(dist, float)
        self.poses = ring_cameras(n_cam, radius=radius)
        self.Fs = [fundamental_from_poses(self.poses[0], self.poses[i], self.K, self.K)
                   for i in range(1, n_cam)]
        self.camera_params = [{"intrinsic_matrix": self.K.tolist(), "distortion_coef": self.dist.tolist()}
                              for _ in range(n_cam)]

    def markers(self, rng, n_markers, extent=0.5):
        return rng.uniform(-extent, extent, size=(n_markers, 3))

    def centroids(self, markers, rng=None, jitter=0.0):
        """Integer pixel centroids per camera (ideal detections), list of [M,2] int arrays."""
        out = []
        for pose in self.poses:
            px = project(markers, pose, self.K, ZERO_DIST)
            if rng is not None and jitter:
                px = px + rng.normal(0, jitter, px.shape)
            out.append(np.floor(px).astype(np.int64))
        return out

    def render(self, rng, markers, cam, radius_range=(16.0, 22.0), noise_max=60, salt=0.0, distorted=True, noise_min=0):
        """uint8[H,W] frame of camera `cam`: discs (core 255, 1.5 px linear edge) over uniform noise in
        [noise_min, noise_max]."""
        H, W = self.height, self.width
        img = rng.integers(noise_min, noise_max + 1, size=(H, W), dtype=np.uint8)
        if salt > 0:
            n = int(salt * H * W)
            img[rng.integers(0, H, n), rng.integers(0, W, n)] = 255
        px = project(markers, self.poses[cam], self.K, self.dist if distorted else ZERO_DIST)
        radii = rng.uniform(radius_range[0], radius_range[1], size=len(markers))
        for (u, v), r in zip(px, radii):
            x0, x1 = int(np.floor(u - r - 2)), int(np.ceil(u + r + 2)) + 1
            y0, y1 = int(np.floor(v - r - 2)), int(np.ceil(v + r + 2)) + 1
            x0c, x1c, y0c, y1c = max(x0, 0), min(x1, W), max(y0, 0), min(y1, H)
            if x0c >= x1c or y0c >= y1c:
                continue
            yy, xx = np.mgrid[y0c:y1c, x0c:x1c]
            d = np.sqrt((xx - u) ** 2 + (yy - v) ** 2)
            val = np.clip((r + 0.75 - d) / 1.5, 0.0, 1.0) * 255.0
            patch = img[y0c:y1c, x0c:x1c]
            np.maximum(patch, val.astype(np.uint8), out=patch)
        return img

    def render_batch(self, seed, n_steps, n_markers, cams=None, **kw):
        """uint8[T, C, H, W] frames; time step t uses rng seed+t."""
        cams = list(range(self.n_cam)) if cams is None else list(cams)
        out = np.empty((n_steps, len(cams), self.height, self.width), np.uint8)
        for t in range(n_steps):
            rng = np.random.default_rng(seed + t)
            mk = self.markers(rng, n_markers)
            for j, c in enumerate(cams):
                out[t, j] = self.render(rng, mk, c, **kw)
        return out
