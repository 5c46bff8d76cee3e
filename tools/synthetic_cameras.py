"""Synthetic camera rig for benchmarks, tests and the fixture generator."""
import numpy as np


def synthetic_occ2img(rng, B, T, N, image_h, image_w):
    """N pinhole cameras at 360/N degree yaw spacing, T ego poses -> [B, T*N, 4, 4]."""
    mats = np.zeros((B, T * N, 4, 4), np.float64)
    f = 0.8 * image_w
    K = np.array([[f, 0, image_w / 2, 0], [0, f, image_h / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    for b in range(B):
        for t in range(T):
            ego = np.eye(4)
            ego[:3, 3] = rng.uniform(-1.5, 1.5, 3) * np.array([1, 1, 0.1])
            for n in range(N):
                yaw = 2 * np.pi * n / N + rng.uniform(-0.05, 0.05)
                c, s = np.cos(yaw), np.sin(yaw)
                # camera looks along +x_ego rotated by yaw; cam axes: x right, y down, z forward
                R = np.array([[s, -c, 0], [0, 0, -1], [c, s, 0]])
                E = np.eye(4)
                E[:3, :3] = R
                E[:3, 3] = -R @ np.array([0.0, 0.0, 1.5])
                mats[b, t * N + n] = K @ E @ ego
    return mats.astype(np.float32)
