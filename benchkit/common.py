"""Small helpers shared by the legs of bench.py."""
import ctypes as C
import json
import sys


def _release():
    """Between legs of one process: drop what the finished leg allocated (its locals are gone) and reset the library's global modes."""
    import gc
    import torch
    from sings_amd import rasterizer as _rz
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    _rz.set_overflow_check("sync")
    _rz.reset_overflow_state()


def _tile_list_stats(eng, W, H):
    """mean / max length of the per-tile depth-sorted lists of the engine's last forward (reported with
    every number); read from the tile ranges in the binning workspace."""
    import torch
    T = ((W + 15) // 16) * ((H + 15) // 16)
    rg = eng.binning[eng.L.bin_ranges:eng.L.bin_ranges + 8 * T].view(torch.int32).view(T, 2)
    n = (rg[:, 1] - rg[:, 0]).clamp_(min=0)
    return float(n.float().mean().item()), int(n.max().item())


def _emit(obj):
    """Print the result as the LAST line of stdout: text that native libraries (RCCL) left in the C stdio buffer is
    flushed first, otherwise it would come out at process exit, after the JSON line."""
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(obj), flush=True)
