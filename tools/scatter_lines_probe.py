"""How many memory-side requests does the field's gradient scatter NEED?  (the question behind "sparse records
for levels 8-15".)

A float-atomic request is one (instruction, 64-byte line) pair; the backward sends the gradient of a sample's eight corners at a
level as x-pairs, i.e. one request per distinct line among them.  Whatever intermediate layout a scheme uses (cell-major
records, hashed records, sorted runs), the gradient TABLE has to be touched once per distinct line a batch's samples reach --
that count, per level, is the floor of any scheme that merges perfectly across the whole batch, and the direct count
(distinct lines per sample, summed) is what a scheme without merging sends.  Their ratio is the most a better merge can save.

Samples: the training probe's geometry (random pixels of 100 orbit cameras at 800 x 800, near 0.05, scene contraction, S
uniform-in-disparity-ish samples per ray from the piecewise sampler -- the final samples of a fitted model cluster on surfaces,
which only RAISES sharing at the coarse levels that are cell-major already).

    python tools/scatter_lines_probe.py [rays] [samples_per_ray]      (torch on cuda:0; no library call)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cropnerf_amd import config as PC, synthetic  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    g = torch.Generator().manual_seed(0)
    c2w, intr = synthetic.orbit_cameras(100)
    cam = torch.randint(0, 100, (R,), generator=g)
    py = torch.randint(0, 800, (R,), generator=g).float() + 0.5
    px = torch.randint(0, 800, (R,), generator=g).float() + 0.5
    K = intr[cam]
    d_cam = torch.stack([(px - K[:, 2]) / K[:, 0], -(py - K[:, 3]) / K[:, 1], -torch.ones(R)], -1)
    rot, org = c2w[cam][:, :, :3], c2w[cam][:, :, 3]
    d = torch.nn.functional.normalize(torch.einsum("rij,rj->ri", rot, d_cam), dim=-1)
    # piecewise-linear-in-disparity spacing between near 0.05 and far 1000 (the proposal sampler's first level), jittered
    u = (torch.arange(S)[None, :] + torch.rand(R, S, generator=g)) / S
    near, far = 0.05, 1000.0
    s_near, s_far = (lambda x: torch.where(x < 1, x / 2, 1 - 1 / (2 * x)))(torch.tensor([near, far]))
    sp = s_near + u * (s_far - s_near)
    t = torch.where(sp < 0.5, 2 * sp, 1 / (2 - 2 * sp))
    pos = (org[:, None, :] + d[:, None, :] * t[..., None]).reshape(-1, 3).to(dev)
    # L-inf scene contraction, then (p + 2) / 4  (fruit_nerf.py:94, fruit_field.py:172-174)
    mag = pos.abs().max(dim=-1, keepdim=True).values
    pos = torch.where(mag < 1, pos, (2 - 1 / mag) * (pos / mag))
    pos = ((pos + 2) / 4).clamp(0, 1)
    spec = PC.GridSpec()
    T = 1 << spec.log2_hashmap_size
    m1, m2 = 2654435761, 805459861
    n = pos.shape[0]
    print(f"{R} rays x {S} samples = {n} samples; table 2^{spec.log2_hashmap_size} entries per level, 8 bytes each (8 entries per line)")
    print("level  res   direct lines/sample   distinct lines in the batch   floor / direct   cells / sample")
    tot_direct = tot_floor = 0
    for lvl, scale in enumerate(spec.scalings()):
        x = pos * scale
        f = torch.floor(x).to(torch.int64)
        lines = []
        for c in range(8):
            cx, cy, cz = f[:, 0] + (c & 1), f[:, 1] + ((c >> 1) & 1), f[:, 2] + (c >> 2)
            e = (cx ^ ((cy * m1) & 0xFFFFFFFF) ^ ((cz * m2) & 0xFFFFFFFF)) & (T - 1)
            lines.append(e >> 3)
        L = torch.stack(lines, -1)  # [n, 8]
        Ls, _ = torch.sort(L, dim=-1)
        direct = int((Ls[:, 1:] != Ls[:, :-1]).sum()) + n
        floor = int(torch.unique(L).numel())
        cells = int(torch.unique(f[:, 0] + (f[:, 1] << 21) + (f[:, 2] << 42)).numel())
        tot_direct += direct
        tot_floor += floor
        print(f"{lvl:5d} {int(scale):5d} {direct / n:20.2f} {floor:29d} {floor / direct:16.3f} {cells / n:15.3f}")
    print(f"all levels: direct {tot_direct / n:.1f} lines per sample, perfect batch-wide merge {tot_floor / n:.1f} "
          f"({tot_floor / tot_direct:.3f} of direct)")


if __name__ == "__main__":
    main()
