"""Engine-level rules of the C ABI that need a device: configuration limits, the process-wide LDS
attribute, which intermediates an entry keeps."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import stereo_synthetic as syn                      # noqa: E402
from oracle_lib import OracleConfig                 # noqa: E402


@pytest.fixture(scope="module")
def cd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cuda_depth
    return cuda_depth


def test_radii_beyond_the_lds_tile_are_refused_with_a_message(cd):
    """large_mbm_radius = 20 passes the range check (<= 32) but its tile needs ~72 KB of LDS:
    smx_create must return SMX_ERR_UNSUPPORTED and say why (and not read the freed engine)."""
    from cuda_depth import _native as N
    cfg = cd.StereoMatchingConfiguration(height=96, width=160, min_disparity=0, max_disparity=31, large_mbm_radius=20)
    with pytest.raises(RuntimeError, match=r"radii too large for the LDS tile.*large_mbm_radius 20.*\(status -5\)"):
        cd.StereoMatching(cfg)
    c = cfg._as_struct(0, 1, 0)
    h = C.c_void_p()
    assert N.LIB.smx_create(C.byref(c), C.byref(h)) == -5 and not h.value
    # the largest radius the header documents for ncc radius 1 still works
    cd.StereoMatching(cd.StereoMatchingConfiguration(height=96, width=160, min_disparity=0, max_disparity=31,
                                                     large_mbm_radius=18))


def test_a_later_small_engine_does_not_shrink_an_earlier_engines_lds(cd, oracle_omp):
    """MaxDynamicSharedMemorySize is per function and process-wide: engine A (Dd = 128, 80 KB exact-order
    tiles) must still run after engine B (Dd = 16) was created."""
    H, W, K = 96, 400, 2
    a = cd.StereoMatching(cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0,
                                                         max_disparity=255), max_batch=8)
    b = cd.StereoMatching(cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0,
                                                         max_disparity=31), max_batch=8)
    L = np.stack([syn.random_rgb_pair(H, W, 256, K, 70 + i)[0] for i in range(8)])
    R = np.stack([syn.random_rgb_pair(H, W, 256, K, 70 + i)[1] for i in range(8)])
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out_b = b.compute_disparity_map_batch(tl, tr).cpu().numpy()
    out_a = a.compute_disparity_map_batch(tl, tr).cpu().numpy()          # 8 pairs: the unsplit 80 KB kernel
    out_a1 = a.compute_disparity_map(tl[0], tr[0]).cpu().numpy()         # 1 pair: the disparity-split kernel
    oa = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=255)
    ob = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=31)
    for i in (0, 7):
        assert np.array_equal(out_a[i], oracle_omp.run(oa, L[i], R[i])), f"engine A pair {i}"
        assert np.array_equal(out_b[i], oracle_omp.run(ob, L[i], R[i])), f"engine B pair {i}"
    assert np.array_equal(out_a1, out_a[0])


def test_gray_f32_entry_keeps_no_gray_planes(cd):
    from cuda_depth import _native as N
    H, W = 64, 96
    sm = cd.StereoMatching(cd.StereoMatchingConfiguration(height=H, width=W, min_disparity=0, max_disparity=15))
    l, r, _ = syn.make_pair(H, W, 16, 2, 0)
    with pytest.raises(RuntimeError, match="no call has been made"):
        sm.intermediate(N.STAGE_GRAY_LEFT)
    sm.compute_disparity_map_gray(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda())
    with pytest.raises(RuntimeError, match="caller's own buffers"):
        sm.intermediate(N.STAGE_GRAY_LEFT)
    sm.compute_disparity_map_gray(torch.from_numpy(l.astype(np.uint8)).cuda(), torch.from_numpy(r.astype(np.uint8)).cuda())
    assert np.array_equal(sm.intermediate(N.STAGE_GRAY_LEFT).cpu().numpy(), l)       # the u8 entry owns its planes
    sm.compute_disparity_map(torch.from_numpy(syn.gray_to_rgb(l)).cuda(), torch.from_numpy(syn.gray_to_rgb(r)).cuda())
    assert sm.intermediate(N.STAGE_GRAY_RIGHT).shape == (H, W)
