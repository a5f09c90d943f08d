"""Checks against mathematical definitions that do not depend on either restatement of the reference: the two-pass L1
transform (imgproc.h:137-146, 176-181) is the exact city-block distance to the nearest rasterised pixel, so stage 1 of an
L1 build must equal a brute-force minimum over the seed pixels -- for the oracle (CPU) and for the HIP path (GPU)."""
import numpy as np
import pytest

from oracle import oracle as O

FLT_MAX = np.float32(3.4028234663852886e38)


def _brute_force_l1(vol):
    """vol: (depth, W, H) stage-1 volume; seeds are its zeros.  Returns the exact city-block distance per slice."""
    out = np.empty_like(vol)
    m, W, H = vol.shape
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")
    for k in range(m):
        sx, sy = np.nonzero(vol[k] == 0)
        if len(sx) == 0:
            out[k] = FLT_MAX
            continue
        d = np.full((W, H), np.iinfo(np.int64).max, dtype=np.int64)
        for a in range(0, len(sx), 256):  # blocks of seeds keep the broadcast small
            dd = np.abs(xs[:, :, None] - sx[None, None, a:a + 256]) + np.abs(ys[:, :, None] - sy[None, None, a:a + 256])
            d = np.minimum(d, dd.min(axis=2))
        out[k] = d.astype(np.float32)
    return out


def _scene(S, n, seed):
    from openfdcm_amd import synthetic
    return synthetic.scene(S, n, seed)


@pytest.mark.parametrize("S,n,depth,seed", [(48, 9, 5, 2), (97, 20, 7, 3), (128, 30, 12, 4)])
def test_oracle_l1_transform_is_the_city_block_distance(S, n, depth, seed):
    orc = O.build(_scene(S, n, seed), depth=depth, coeff=5.0, padding=1.0, distance=O.L1, nthreads=4, stop_after=1)
    vol = orc.volume()
    assert (vol == 0).any()
    assert np.array_equal(vol, _brute_force_l1(vol))


@pytest.mark.gpu
@pytest.mark.parametrize("S,n,depth,seed", [(97, 20, 7, 3), (256, 60, 16, 5)])
def test_device_l1_transform_is_the_city_block_distance(S, n, depth, seed):
    from openfdcm_amd.engine import DeviceFeatureMap
    dev = DeviceFeatureMap.build(_scene(S, n, seed), depth=depth, coeff=5.0, padding=1.0, distance=O.L1, stop_after=1)
    vol = dev.volume()
    assert (vol == 0).any()
    assert np.array_equal(vol, _brute_force_l1(vol))
