"""Generate ``demo_scene.npz``: the reference's demo assets (/root/reference/assets/demo: pcd.ply, img_{0..3}.png, mask_{0..3}.png,
R_list / t_list / intr_list) as one small data fixture for the end-to-end fitting test -- what ``demo.py:124-159`` feeds
``GSTrainer.update_state_no_env`` (/root/reference/src/real_world/gs/trainer.py:76-98).  Images and masks are box-filtered down by 4
(1280x720 -> 320x180, uint8) and the intrinsics scaled with them, to keep the fixture small; the point cloud is stored whole.
Data only -- no reference source text.

    python tests/golden/gen_demo_fixture.py
"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/assets/demo"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "demo_scene.npz")
F = 4


def main():
    raw = open(os.path.join(SRC, "pcd.ply"), "rb").read()
    head = raw[:raw.index(b"end_header\n")].decode()
    n = int([l for l in head.splitlines() if l.startswith("element vertex")][0].split()[-1])
    body = raw[raw.index(b"end_header\n") + len(b"end_header\n"):]
    rec = np.frombuffer(body, dtype=np.dtype([("xyz", "<f8", 3), ("rgb", "u1", 3)]), count=n)
    imgs, masks = [], []
    for v in range(4):
        im = np.asarray(Image.open(os.path.join(SRC, f"img_{v}.png")).convert("RGB"), np.float32)
        mk = np.asarray(Image.open(os.path.join(SRC, f"mask_{v}.png")), np.float32)
        if mk.ndim == 3:
            mk = mk[..., 0]
        H, W = im.shape[0] // F * F, im.shape[1] // F * F
        imgs.append(im[:H, :W].reshape(H // F, F, W // F, F, 3).mean((1, 3)).round().astype(np.uint8))
        masks.append(mk[:H, :W].reshape(H // F, F, W // F, F).mean((1, 3)).round().astype(np.uint8))
    intr = np.load(os.path.join(SRC, "intr_list.npy")).astype(np.float64).copy()
    intr[:, :2, :] /= F                                   # fx, fy, cx, cy of the down-sampled images
    np.savez_compressed(OUT, xyz=rec["xyz"].astype(np.float32), rgb=rec["rgb"].copy(), imgs=np.stack(imgs), masks=np.stack(masks),
                        R_list=np.load(os.path.join(SRC, "R_list.npy")), t_list=np.load(os.path.join(SRC, "t_list.npy")), intr_list=intr,
                        downsample=np.array([F]))
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", n, "points; images", imgs[0].shape, "mask max", int(np.stack(masks).max()))


if __name__ == "__main__":
    main()
