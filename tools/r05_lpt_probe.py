"""GPU box, timing only: would longest-list-first dispatch shorten the ONE-view backward composite?  python tools/r05_lpt_probe.py
The dense backward kernel takes work item `items[sg_tile_of_block(blockIdx.x)]`; here the item array the forward left is re-ordered on the
device between forward and backward so that ascending blockIdx meets the tiles by descending list length (no kernel change), and the
records kernel is timed with HIP events against the forward's own order."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from sings_amd import _lib
from sings_amd.engine import RasterEngine, _ptr
from sings_amd.rasterizer import GaussianRasterizationSettings
from sings_amd.scene import synthetic_scene

dev = torch.device("cuda:0")
N, W, H, deg = 200000, 1920, 1080, 3
s = synthetic_scene(N, W, H, deg, 3)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]), scale_modifier=1.0,
                                   viewmatrix=t(s["viewmatrix"]), projmatrix=t(s["projmatrix"]), sh_degree=deg, campos=t(s["campos"]),
                                   prefiltered=False, debug=False)
means3D, shs, opac, scales, rots = t(s["means3D"]), t(s["shs"]), t(s["opacities"]), t(s["scales"]), t(s["rotations"])
dL = t(s["dL_dimage"])
e = RasterEngine(N, W, H, shs.shape[1], dev, capacity_pairs=900000)
e.set_camera(rs, short_lists=True)
L = e.L
gx, gy = (W + 15) // 16, (H + 15) // 16
T = gx * gy
RUN = 16


def tile_of_block(b):
    xcd, slot = b & 7, b >> 3
    within = (slot + 7 * (slot >> 5)) & (RUN - 1)
    return ((slot // RUN) * 8 + xcd) * RUN + within


e.forward(means3D, shs, opac, scales, rots); torch.cuda.synchronize()
hdr = e.binning[L.bin_header:L.bin_header + 32].view(torch.int32)
nitems = int(hdr[5])
ranges = e.binning[L.bin_ranges:L.bin_ranges + 8 * T].view(torch.int32).view(T, 2)
lens = (ranges[:, 1] - ranges[:, 0]).to(torch.int64)
print("items", nitems, "tiles", T, "list length mean / max", float(lens.float().mean()), int(lens.max()))
nb = ((nitems + 127) // 128) * 128
blocks = np.arange(nb)
slot_of_block = torch.from_numpy(np.array([tile_of_block(int(b)) for b in blocks], dtype=np.int64)).to(dev)
valid = slot_of_block < nitems


def permute(mode):
    items = e.binning[L.bin_items:L.bin_items + 4 * nitems].view(torch.int32)
    it = items.clone()
    w = lens[(it & 0xfffff).long()]
    order = torch.argsort(w, descending=(mode == "longest_first"), stable=True)
    srt = it[order]
    # block b (dispatch order) works on slot tile_of_block(b): give the k-th valid block the k-th item of the sorted list
    tgt = slot_of_block[valid][:nitems]
    out = torch.empty_like(it)
    out[tgt] = srt
    items.copy_(out)


def time_bwd(mode, iters=30):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        e.forward(means3D, shs, opac, scales, rots)
        if mode != "forward_order":
            permute(mode)
        a.record()
        _lib.check(e.lib.sg_rasterize_backward_records(C.byref(e._s), e.P, _ptr(e.geom), _ptr(e.binning), e.cap, _ptr(e.img), _ptr(e.bwd_ws),
                                                       _ptr(dL), e._stream()), "records")
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return 1e3 * ts[len(ts) // 2]


for mode in ("forward_order", "longest_first", "shortest_first", "forward_order", "longest_first"):
    print(f"{mode:16s} backward composite (records) {time_bwd(mode):7.1f} us", flush=True)
