"""Generate ``demo_splat.npz``: the ONE artefact under /root/reference that the real (upstream CUDA) rasterizer produced --
``assets/demo/gs_orig.splat``, the Gaussians ``GSTrainer.train`` fitted to the four masked demo images
(/root/reference/src/real_world/gs_sim_real_gradio.py:170-191, written by ``save_to_splat``,
/root/reference/src/real_world/gs/convert.py:23-51: 32 bytes per Gaussian = position f32 x 3 (mean-subtracted, rotated by the inverse of
rot_x_90), scale f32 x 3, rgba u8 x 4, quaternion u8 x 4) -- together with what it was fitted to: the four camera images times their
masks at the full 1280x720 (/root/reference/src/real_world/gs/trainer.py:74-80), the masks, and the cameras.  Data only.

    python tests/golden/gen_demo_splat_fixture.py

Used by tools/splat_sanity.py: the HIP path renders these Gaussians from the four cameras; the masked PSNR against
the images is a SANITY check of the pixel-centre / dilation / axis conventions (u8 colours and quaternions cap it far below 1e-4)."""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/assets/demo"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "demo_splat.npz")


def main():
    raw = np.fromfile(os.path.join(SRC, "gs_orig.splat"), dtype=np.uint8).reshape(-1, 32)
    pos = raw[:, 0:12].copy().view(np.float32)
    scales = raw[:, 12:24].copy().view(np.float32)
    rgba, quat = raw[:, 24:28].copy(), raw[:, 28:32].copy()
    imgs, masks = [], []
    for v in range(4):
        im = np.asarray(Image.open(os.path.join(SRC, f"img_{v}.png")).convert("RGB"), np.uint8)
        mk = np.asarray(Image.open(os.path.join(SRC, f"mask_{v}.png")))
        if mk.ndim == 3:
            mk = mk[..., 0]
        mk = (mk > 0).astype(np.uint8)
        imgs.append(im * mk[..., None])
        masks.append(mk)
    np.savez_compressed(OUT, pos=pos, scales=scales, rgba=rgba, quat=quat, imgs_masked=np.stack(imgs), masks=np.packbits(np.stack(masks), axis=-1),
                        R_list=np.load(os.path.join(SRC, "R_list.npy")), t_list=np.load(os.path.join(SRC, "t_list.npy")),
                        intr_list=np.load(os.path.join(SRC, "intr_list.npy")))
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", pos.shape[0], "Gaussians; images", imgs[0].shape, "foreground pixels per view", [int(m.sum()) for m in masks])


if __name__ == "__main__":
    main()
