"""GPU box: the target-moments kernel (image_loss_fwd MODE 2: one map in, two blurred maps out -- memory instructions dominate) with
16-byte global accesses (aligned input) against 4-byte ones (the same data one float off alignment): is the per-instruction pace of the
memory pipe what bounds the image-loss kernels?"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from diff_gaussian_rasterization import _hip
from gsdyn import losses as L
dev = torch.device("cuda:0")
lib = _hip.load_library()
win = _hip._window(L._window_1d())
Cc, H, W, NT = 4, 800, 800, 6      # 6 targets of 4 channels per timed batch (the library takes up to 4 channels per call)
bufs = [torch.rand(Cc * H * W + 4, device=dev) for _ in range(NT)]
outs = [torch.empty(2 * Cc * H * W + 4, device=dev) for _ in range(NT)]
res = {}
for name, off in (("aligned (16-byte accesses)", 0), ("one float off (4-byte accesses)", 1), ("aligned (16-byte accesses) ", 0)):
    imgs = [b_[off:off + Cc * H * W].view(Cc, H, W) for b_ in bufs]
    ms = [o_[off:off + 2 * Cc * H * W].view(2, Cc, H, W) for o_ in outs]
    st = _hip._stream(dev)
    def batch():
        for img, m in zip(imgs, ms):
            rc = lib.gsr_target_moments(win, Cc, H, W, _hip._ptr(img), _hip._ptr(m), st)
            assert rc == 0, lib.gsr_last_error()
    for _ in range(3):
        batch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        batch()
    b.record(); torch.cuda.synchronize()
    res[name] = (a.elapsed_time(b) / 20 * 1e3, ms[0].clone())
    print(f"{name}: {res[name][0]:.1f} us per batch (6 launches, 24 planes of 800x800 in all)")
ks = list(res)
print("maps equal:", torch.equal(res[ks[0]][1], res[ks[1]][1]))
