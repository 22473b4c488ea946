"""A fixed-seed slice of the randomised HIP-vs-oracle sweeps (tests/tools/fuzz_parity.py, fuzz_skinned.py) as collected tests:
round 3 ran the sweeps by hand and committed their logs -- the driver never saw them.  The crafted tile lists sit exactly on the
internal boundaries of the path (128 = rank sort, 256 = depth segment / forward batch, 1024 = in-kernel sort, 4096, 12 288 = keys
the bucket sort keeps resident in LDS, beyond: group slots + sg_group_sort_kernel); the random cases draw image size, Gaussian
count, SH degree, opacity and splat scale; every case checks bit-exact binning, RGB off borderline pixels and EVERY gradient element --
and then renders once more with the direct-binning promise of its regime (SG_FLAG_SHORT_LISTS / SG_FLAG_LONG_ROWS): the same lists,
keys and image bit for bit where the promise holds, a refusal where it does not (crafted 20 000 > a 16 384-key row)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fp():
    from tests.tools import fuzz_parity
    return fuzz_parity


@pytest.mark.parametrize("n", [127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 8193,
                               12287, 12288, 12289, 20000])
def test_crafted_tile_list_lengths(n):
    fp = _fp()
    assert n in fp.CRAFTED_LENGTHS
    R, mx = fp.check(fp.crafted(n, n), f"crafted {n}")
    assert mx >= n and R >= n


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_clustered_depths_walk_every_level_of_the_bucket_sort(seed):
    fp = _fp()
    z = fp.clustered_depths(seed)
    R, mx = fp.check(fp.crafted(z.size, 100 + seed, depths=z), f"clustered depths {seed}")
    assert mx >= z.size


def test_random_scenes_fixed_seed_slice():
    """10 random small frames + 2 frames of more than 4096 tiles (the other pair of composite kernels), FUZZ_SEED 1234."""
    fp = _fp()
    rs = np.random.RandomState(1234)
    for c in range(10):
        s, tag = fp.random_case(rs)
        fp.check(s, f"case {c}: {tag}")
    for c in range(2):
        s, tag = fp.random_case(rs, large=True)
        fp.check(s, f"large case {c}: {tag}")


def test_skinned_random_slice():
    """6 cases of tests/tools/fuzz_skinned.py (FUZZ_SEED 4321): joint counts 24 / 30 / 52, isotropic and 6-D rotations."""
    from tests import test_gpu_skinned as T
    rs = np.random.RandomState(4321)
    for c in range(6):
        J = int(rs.choice([24, 30, 52])); iso = bool(rs.rand() < 0.3); seed = int(rs.randint(1 << 20))
        T.test_fused_forward(J, iso, seed)
        T.test_fused_backward(J, iso, seed + 1)
