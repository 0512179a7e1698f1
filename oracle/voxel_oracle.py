"""CPU ORACLE (test infrastructure only) — NumPy restatement of the voxeliser specification in
timed-design_amd/timed_hip/voxeliser.py (items 3-6), float32 with the same operation order as the kernel.

PARITY UNPINNED: the reference delegates voxelisation to aposteriori==2.4.0 (ui.py:73-86, README.md:83-97), which is
absent from /root/reference and from this image, and the reference holds no test or fixture at this boundary.  This
file pins the GPU kernel to the written specification, not to aposteriori.

Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np


def voxelise(atoms_xyz, atom_channel, atom_sigma, frames_rt, voxels_per_side=21, frame_edge_length=21.0, n_channels=5,
             gaussian=True):
    f32 = np.float32
    xyz = np.asarray(atoms_xyz, dtype=f32)
    chn = np.asarray(atom_channel)
    sig = np.asarray(atom_sigma, dtype=f32)
    frt = np.asarray(frames_rt, dtype=f32)
    V = int(voxels_per_side)
    a = f32(frame_edge_length) / f32(V)
    centre = V // 2
    out = np.zeros((frt.shape[0], V, V, V, n_channels), dtype=f32 if gaussian else np.uint8)
    for r in range(frt.shape[0]):
        R, ca = frt[r, :9].reshape(3, 3), frt[r, 9:]
        d = xyz - ca[None, :]                                            # float32 subtract
        loc = np.stack([(R[i, 0] * d[:, 0] + R[i, 1] * d[:, 1]) + R[i, 2] * d[:, 2] for i in range(3)], axis=1).astype(f32)
        idx = np.floor(loc / a + f32(0.5)).astype(np.int64) + centre
        inside = np.all((idx >= 0) & (idx < V), axis=1)
        for k in np.nonzero(inside)[0]:                                   # atom order
            c = int(chn[k])
            if not 0 <= c < n_channels:
                continue
            if not gaussian:
                out[r, idx[k, 0], idx[k, 1], idx[k, 2], c] = 1
                continue
            inv2s2 = f32(1.0) / (f32(2.0) * sig[k] * sig[k])
            w = np.zeros((3, 3, 3), dtype=f32)
            for dz in range(3):
                for dy in range(3):
                    for dx in range(3):
                        v = idx[k] + np.array([dz - 1, dy - 1, dx - 1])
                        cen = (v - centre).astype(f32) * a
                        e = cen - loc[k]
                        r2 = (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]
                        w[dz, dy, dx] = np.exp(-(r2 * inv2s2), dtype=f32)
            total = f32(0.0)
            for val in w.ravel():                                         # fixed order: dz, dy, dx
                total = f32(total + val)
            for dz in range(3):
                for dy in range(3):
                    for dx in range(3):
                        v = idx[k] + np.array([dz - 1, dy - 1, dx - 1])
                        if np.all((v >= 0) & (v < V)):
                            out[r, v[0], v[1], v[2], c] = f32(out[r, v[0], v[1], v[2], c] + f32(w[dz, dy, dx] / total))
    return out
