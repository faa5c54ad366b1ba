"""A COLMAP *binary* model written field by field from COLMAP's documented on-disk format (the layout that COLMAP's own
`scripts/python/read_write_model.py` documents), independently of sucre_amd.sfm's reader and of its text writer:

  cameras.bin   uint64 n; per camera: int32 camera_id, int32 model_id, uint64 width, uint64 height, float64 params[k]
                (model_id 1 = PINHOLE: fx, fy, cx, cy)
  images.bin    uint64 n; per image: int32 image_id, float64 qvec[4] (w, x, y, z), float64 tvec[3], int32 camera_id,
                name bytes + '\\0', uint64 n_points2D, per point2D: float64 x, float64 y, int64 point3D_id
  points3D.bin  uint64 n; per point: uint64 id, float64 xyz[3], uint8 rgb[3], float64 error, uint64 track_length,
                per track element: int32 image_id, int32 point2D_idx

Run anywhere (numpy + scipy):  python tests/golden/gen_colmap_bin_fixture.py
Writes tests/golden/colmap_bin_model/{cameras,images,points3D}.bin and expected.npz: what a reader must produce --
per image the WORLD-from-camera pose (the reference inverts pycolmap's cam_from_world, sfm.py:219-222), computed here
in float64 with scipy's Rotation from the very (qvec, tvec) written; two PINHOLE cameras stored in descending id order;
image ids not contiguous and not sorted; names with a sub-directory; 2-D points on most images (their records must be
skipped correctly); a quaternion with negative w.
"""
import struct
from pathlib import Path

import numpy as np
from scipy.spatial.transform import Rotation

HERE = Path(__file__).resolve().parent
OUT = HERE / 'colmap_bin_model'


def main():
    rng = np.random.default_rng(20261003)
    OUT.mkdir(exist_ok=True)
    cameras = [  # (camera_id, model_id, width, height, params) -- stored in this (descending id) order
        (7, 1, 1920, 1080, [1497.6, 1501.25, 960.0, 540.5]),
        (2, 1, 640, 480, [499.2, 498.75, 320.0, 239.5]),
    ]
    with open(OUT / 'cameras.bin', 'wb') as f:
        f.write(struct.pack('<Q', len(cameras)))
        for cam_id, model_id, w, h, params in cameras:
            f.write(struct.pack('<iiQQ', cam_id, model_id, w, h))
            f.write(struct.pack(f'<{len(params)}d', *params))
    ids = [12, 3, 40, 5, 9, 31]
    names = ['dive1/frame_0012.png', 'frame 0003.jpg', 'dive2/frame_0040.png', 'frame_0005.JPG', 'a.png', 'dive1/z.png']
    cam_of = [7, 2, 7, 2, 7, 7]
    rec = dict(image_id=[], camera_id=[], qvec=[], tvec=[], R_wfc=[], t_wfc=[], n2d=[])
    with open(OUT / 'images.bin', 'wb') as f:
        f.write(struct.pack('<Q', len(ids)))
        for i, (image_id, name, cam_id) in enumerate(zip(ids, names, cam_of)):
            rot = Rotation.from_rotvec(rng.normal(0, 0.6, 3))
            x, y, z, w = rot.as_quat()           # scipy: scalar last
            if i == 2 and w > 0:                 # q and -q are the same rotation; COLMAP files hold either sign
                x, y, z, w = -x, -y, -z, -w
            q = np.array([w, x, y, z])
            t = rng.normal(0, 2.0, 3)
            f.write(struct.pack('<i', image_id))
            f.write(struct.pack('<4d', *q))
            f.write(struct.pack('<3d', *t))
            f.write(struct.pack('<i', cam_id))
            f.write(name.encode() + b'\x00')
            n2d = [0, 3, 17, 1, 0, 250][i]
            f.write(struct.pack('<Q', n2d))
            for _ in range(n2d):
                f.write(struct.pack('<ddq', rng.uniform(0, 640), rng.uniform(0, 480), int(rng.integers(-1, 50))))
            R_cfw = rot.as_matrix()              # cam_from_world rotation
            rec['image_id'].append(image_id); rec['camera_id'].append(cam_id); rec['qvec'].append(q); rec['tvec'].append(t)
            rec['R_wfc'].append(R_cfw.T); rec['t_wfc'].append(-R_cfw.T @ t); rec['n2d'].append(n2d)
    with open(OUT / 'points3D.bin', 'wb') as f:
        f.write(struct.pack('<Q', 4))
        for pid in (1, 2, 10, 11):
            f.write(struct.pack('<Q3d3Bd', pid, *rng.normal(0, 1, 3), 10, 20, 30, 0.5))
            track = [(12, 0), (3, 1)] if pid < 10 else [(40, 5), (5, 0), (31, 100)]
            f.write(struct.pack('<Q', len(track)))
            for image_id, idx in track:
                f.write(struct.pack('<ii', image_id, idx))
    np.savez(OUT / 'expected.npz', names=np.array(names), camera_ids=np.array([c[0] for c in cameras]),
             camera_wh=np.array([[c[2], c[3]] for c in cameras]), camera_params=np.array([c[4] for c in cameras]),
             **{k: np.array(v) for k, v in rec.items()})
    for p in sorted(OUT.iterdir()):
        print(p.name, p.stat().st_size, 'bytes')


if __name__ == '__main__':
    main()
