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


# ----------------------------------------------------------------------------- stream lanes
def _lane_inputs(n, H, W, Dd, K, seed0=300):
    L = np.stack([syn.make_pair(H, W, Dd, K, seed0 + i)[0] for i in range(n)])
    R = np.stack([syn.make_pair(H, W, Dd, K, seed0 + i)[1] for i in range(n)])
    return L, R


def _submit(sm, l, r, out=None):
    """One call on the engine's own streams, joined into the current stream."""
    o = sm.compute_disparity_map_batch(l, r, out=out, engine_streams=True)
    sm.join()
    return o


def test_stream_lanes_same_bits_as_one_stream(cd, oracle_omp, monkeypatch):
    """SMX_STREAM_ENGINE: a call of >= overlap_min_pairs pairs (default: twice the smallest batch in the throughput shape; 32 here) is enqueued as two halves
    on the engine's two streams (include/stereo_mi355x.h): same bits as the call on a caller's stream and
    as the oracle, for even and odd n, below and above the threshold, for every entry, and for the
    intermediates of pairs of the second half."""
    from cuda_depth import _native as N
    H, W, K, Dd = 64, 200, 2, 16
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    L, R = _lane_inputs(41, H, W, Dd * K, K)
    L[5] += 0.25                                        # one pair off the grid: its half takes the exact-order kernel too
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    sm = cd.StereoMatching(cfg, max_batch=41, overlap_min_pairs=32)
    assert sm.overlap_lanes(41) == 2 and sm.overlap_lanes(32) == 2 and sm.overlap_lanes(31) == 1
    monkeypatch.setenv("SMX_OVERLAP_MIN_PAIRS", "16")                 # the environment moves the default only
    assert cd.StereoMatching(cfg, max_batch=41).overlap_lanes(16) == 2
    assert cd.StereoMatching(cfg, max_batch=41, overlap_min_pairs=-1).overlap_lanes(41) == 1
    monkeypatch.delenv("SMX_OVERLAP_MIN_PAIRS")
    # default: twice the smallest batch whose aggregation launch takes the throughput shape (13 / 8 workgroups per CU):
    # this 32 x 100 pooled image is 2 tall workgroups per pair -> 208 pairs per half on a 256-CU device
    cus = cd.StereoMatching(cfg).route_info()["compute_units"]
    n_tall = -(-13 * cus // (8 * 2))
    dflt = cd.StereoMatching(cfg, max_batch=2 * n_tall)
    assert dflt.overlap_lanes(2 * n_tall) == 2 and dflt.overlap_lanes(2 * n_tall - 1) == 1
    del dflt
    c2 = cd.StereoMatchingConfiguration(height=375, width=1242, downscale_factor=2, min_disparity=0, max_disparity=127)
    e2 = cd.StereoMatching(c2, max_batch=32)
    assert e2.overlap_lanes(32) == 2 and (cus != 256 or (e2.overlap_lanes(26) == 2 and e2.overlap_lanes(25) == 1))
    del e2
    torch.cuda.synchronize()                            # inputs complete, as the mode requires
    side = torch.cuda.Stream()
    for n in (41, 32, 31, 2, 1):
        want = sm.compute_disparity_map_batch(tl[:n], tr[:n]).clone()
        want_last = sm.intermediate(N.STAGE_REFINED, n - 1).clone()
        want_costs = sm.intermediate(N.STAGE_MBM_COSTS, n - 1).clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):                   # join into a caller-chosen stream
            got = _submit(sm, tl[:n], tr[:n]).clone()
            last = sm.intermediate(N.STAGE_REFINED, n - 1).clone()
            costs = sm.intermediate(N.STAGE_MBM_COSTS, n - 1).clone()
        side.synchronize()
        assert torch.equal(got, want), f"n={n}"
        assert torch.equal(last, want_last) and torch.equal(costs, want_costs), f"n={n}"
    full = _submit(sm, tl, tr).cpu().numpy()
    for i in (0, 5, 20, 21, 40):
        assert np.array_equal(full[i], oracle_omp.run(ocfg, L[i], R[i])), f"pair {i}"
    # u8 and RGB entries take the same route (different bytes per pair in the input offsets)
    l8, r8 = torch.from_numpy(L.astype(np.uint8)).cuda(), torch.from_numpy(R.astype(np.uint8)).cuda()
    rgb_l = torch.from_numpy(np.stack([syn.gray_to_rgb(x) for x in L[:33]])).cuda()
    rgb_r = torch.from_numpy(np.stack([syn.gray_to_rgb(x) for x in R[:33]])).cuda()
    rgb8_l, rgb8_r = rgb_l.to(torch.uint8), rgb_r.to(torch.uint8)
    torch.cuda.synchronize()
    for a, b in ((l8[:33], r8[:33]), (rgb_l, rgb_r), (rgb8_l, rgb8_r)):
        want = sm.compute_disparity_map_batch(a, b).clone()
        torch.cuda.synchronize()
        assert torch.equal(_submit(sm, a, b), want)
        want_gray = sm.intermediate(N.STAGE_GRAY_RIGHT, 32).clone()        # engine-owned gray of a second-half pair
        sm.compute_disparity_map_batch(a, b)
        assert torch.equal(sm.intermediate(N.STAGE_GRAY_RIGHT, 32), want_gray)


def test_stream_lanes_profile_and_back_to_back_calls(cd):
    """The event profile of a split call counts both halves' launches; calls queued back to back on the
    engine's streams stay ordered among themselves (the same output buffer is rewritten by every call)."""
    H, W, K, Dd = 64, 200, 2, 16
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    sm = cd.StereoMatching(cfg, max_batch=40, overlap_min_pairs=32)
    La, Ra = _lane_inputs(40, H, W, Dd * K, K, 500)
    Lb, Rb = _lane_inputs(40, H, W, Dd * K, K, 700)
    ta, tb = (torch.from_numpy(La).cuda(), torch.from_numpy(Ra).cuda()), (torch.from_numpy(Lb).cuda(), torch.from_numpy(Rb).cuda())
    out = torch.empty((40, H, W), device="cuda")
    torch.cuda.synchronize()
    sm.profile_begin(5)
    for k in range(5):                                  # a b a b a: the last writer wins
        t = ta if k % 2 == 0 else tb
        sm.compute_disparity_map_batch(t[0], t[1], out=out, engine_streams=True)
    sm.join()
    final = out.clone()
    prof = sm.profile_end()
    torch.cuda.synchronize()
    assert prof["prologue"][1] == 10 and prof["fill"][1] == 10      # 5 calls x 2 halves
    ref = cd.StereoMatching(cfg, max_batch=40)
    assert torch.equal(final, ref.compute_disparity_map_batch(ta[0], ta[1]))
    sm.profile_begin(2)
    sm.compute_disparity_map_batch(tb[0], tb[1], out=out)           # caller's stream: one launch per kernel
    prof = sm.profile_end()
    assert prof["prologue"][1] == 1 and torch.equal(out, ref.compute_disparity_map_batch(tb[0], tb[1]))


def test_batch_call_inside_a_captured_graph(cd):
    """A batch call on a caller's stream is plain stream work: it can be captured into a HIP graph."""
    H, W, K, Dd = 64, 200, 2, 16
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    sm = cd.StereoMatching(cfg, max_batch=36, overlap_min_pairs=32)
    L, R = _lane_inputs(36, H, W, Dd * K, K, 900)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((36, H, W), device="cuda")
    want = sm.compute_disparity_map_batch(tl, tr).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sm.compute_disparity_map_batch(tl, tr, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_engine_streams_submit_and_join(cd):
    """SMX_STREAM_ENGINE: calls go to the engine's own streams and pipeline; smx_join (or the next call on a
    caller's stream, or smx_get_intermediate) orders a stream behind them.  Same bits as the ordinary call,
    for split calls and unsplit ones."""
    from cuda_depth import _native as N
    H, W, K, Dd = 64, 200, 2, 16
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    L, R = _lane_inputs(40, H, W, Dd * K, K, 1100)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    torch.cuda.synchronize()                                  # inputs complete, as the mode requires
    for B in (40, 8):
        sm = cd.StereoMatching(cfg, max_batch=B, overlap_min_pairs=32)
        want = sm.compute_disparity_map_batch(tl[:B], tr[:B]).clone()
        want_rev = sm.compute_disparity_map_batch(tr[:B], tl[:B]).clone()
        torch.cuda.synchronize()
        outs = [torch.zeros((B, H, W), device="cuda") for _ in range(6)]
        torch.cuda.synchronize()
        for k, o in enumerate(outs):                          # six calls back to back, alternating inputs
            a, b = (tl, tr) if k % 2 == 0 else (tr, tl)
            sm.compute_disparity_map_batch(a[:B], b[:B], out=o, engine_streams=True)
        sm.join()
        got = [o.clone() for o in outs]                       # on the current stream, behind the join
        torch.cuda.synchronize()
        for k, g in enumerate(got):
            assert torch.equal(g, want if k % 2 == 0 else want_rev), f"B={B} call {k}"
        # a call on the caller's stream after engine-stream calls joins by itself
        sm.compute_disparity_map_batch(tl[:B], tr[:B], out=outs[0], engine_streams=True)
        again = sm.compute_disparity_map_batch(tr[:B], tl[:B]).clone()
        torch.cuda.synchronize()
        assert torch.equal(again, want_rev) and torch.equal(outs[0], want)
        # ... and so does the intermediate read-back
        sm.compute_disparity_map_batch(tl[:B], tr[:B], out=outs[1], engine_streams=True)
        ref = sm.intermediate(N.STAGE_REFINED, B - 1).clone()
        torch.cuda.synchronize()
        sm.compute_disparity_map_batch(tl[:B], tr[:B])
        assert torch.equal(ref, sm.intermediate(N.STAGE_REFINED, B - 1))


def test_engine_stream_calls_of_changing_size_and_mixed_with_caller_stream_calls(cd):
    """The lanes run unordered against each other while they work on disjoint pairs; a call whose split differs from
    what the other lane has in flight, and an engine-stream call after a call on a caller's stream, must wait.
    No host synchronisation between the calls; every call has its own inputs and output."""
    H, W, K, Dd = 96, 320, 2, 24
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    B = 48
    sm = cd.StereoMatching(cfg, max_batch=B, overlap_min_pairs=16)
    ref = cd.StereoMatching(cfg, max_batch=B)
    sets = []
    for k in range(3):
        L, R = _lane_inputs(B, H, W, Dd * K, K, 2000 + 100 * k)
        sets.append((torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()))
    want = {}
    sizes = [48, 20, 8, 40, 48, 12, 30, 16, 48, 4, 44]
    for i, n in enumerate(sizes):
        l, r = sets[i % 3]
        want[i] = ref.compute_disparity_map_batch(l[:n], r[:n]).clone()
    torch.cuda.synchronize()
    for rep in range(3):
        outs = [torch.zeros((n, H, W), device="cuda") for n in sizes]
        torch.cuda.synchronize()
        for i, n in enumerate(sizes):
            l, r = sets[i % 3]
            # every third call goes to the caller's stream: the lanes must then wait for it, and it for them
            sm.compute_disparity_map_batch(l[:n], r[:n], out=outs[i], engine_streams=(i % 3 != 2))
        sm.join()
        torch.cuda.synchronize()
        for i in range(len(sizes)):
            assert torch.equal(outs[i], want[i]), f"rep {rep} call {i} (n={sizes[i]})"


def test_first_engine_stream_call_waits_for_an_unfinished_call_on_a_caller_stream(cd):
    """A FRESH engine: an asynchronous call on the caller's stream, then -- without any host synchronisation -- the
    first SMX_STREAM_ENGINE call with other inputs.  The lanes did not exist during the first call, so nothing
    recorded its tail; the engine must still order the lanes behind it (include/stereo_mi355x.h: 'behind its last
    call on a caller's stream').  Both outputs are compared with a reference engine."""
    H, W, K, Dd = 192, 640, 2, 32
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    B = 48
    ref = cd.StereoMatching(cfg, max_batch=B)
    La, Ra = _lane_inputs(B, H, W, Dd * K, K, 5100)
    Lb, Rb = _lane_inputs(B, H, W, Dd * K, K, 5200)
    ta, tb = (torch.from_numpy(La).cuda(), torch.from_numpy(Ra).cuda()), (torch.from_numpy(Lb).cuda(), torch.from_numpy(Rb).cuda())
    want_a = ref.compute_disparity_map_batch(*ta).clone()
    want_b = ref.compute_disparity_map_batch(*tb).clone()
    torch.cuda.synchronize()
    for rep in range(3):
        sm = cd.StereoMatching(cfg, max_batch=B, overlap_min_pairs=16)       # fresh: no lanes yet
        out_a, out_b = torch.zeros((B, H, W), device="cuda"), torch.zeros((B, H, W), device="cuda")
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(4):                                              # a queue of work on the caller's stream
                sm.compute_disparity_map_batch(*ta, out=out_a)
        sm.compute_disparity_map_batch(*tb, out=out_b, engine_streams=True)  # must not overtake it
        sm.join()
        torch.cuda.synchronize()
        assert torch.equal(out_a, want_a), f"rep {rep}: the caller-stream call was disturbed"
        assert torch.equal(out_b, want_b), f"rep {rep}: the engine-stream call"


def test_stream_capture_with_unjoined_engine_stream_work_is_refused(cd):
    """A capture cannot wait for work outside of it: with SMX_STREAM_ENGINE calls not joined yet, a call on a capturing
    stream returns SMX_ERR_UNSUPPORTED with a message; after a join outside the capture it is captured normally, also
    on an engine that owns lanes."""
    H, W, K, Dd = 64, 200, 2, 16
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    sm = cd.StereoMatching(cfg, max_batch=36, overlap_min_pairs=32)
    L, R = _lane_inputs(36, H, W, Dd * K, K, 910)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((36, H, W), device="cuda")
    want = sm.compute_disparity_map_batch(tl, tr).clone()
    torch.cuda.synchronize()
    sm.compute_disparity_map_batch(tl, tr, out=out, engine_streams=True)     # lanes now exist, work is pending
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    refused = False
    with torch.cuda.stream(s):
        g.capture_begin()
        try:
            sm.compute_disparity_map_batch(tl, tr, out=out)
        except RuntimeError as e:
            refused = "capture" in str(e)
        g.capture_end()
    assert refused
    sm.join()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        sm.compute_disparity_map_batch(tl, tr, out=out)
    out.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    # an engine-stream call after the captured call: ordered by the caller (the sync above), same bits
    sm.compute_disparity_map_batch(tr, tl, out=out, engine_streams=True)
    sm.join()
    torch.cuda.synchronize()
    assert torch.equal(out, sm.compute_disparity_map_batch(tr, tl))


def test_small_engine_stream_calls_alternate_between_the_lanes(cd):
    """Unsplit engine-stream calls that need at most half of the engine's pair slots alternate between the two lanes and
    the two halves of the buffers, so consecutive single frames (the reference runner's call pattern,
    depth_estimation_pipeline_runner.py:51-52) run side by side.  Seven calls of 1 - 3 pairs with different inputs, gray
    and RGB (the disparity-split exact-order kernel has one slice region per lane), no host synchronisation; every output
    and the intermediates of the last call (which sit in the second half of the buffers) against a reference engine."""
    from cuda_depth import _native as N
    H, W, K, Dd = 96, 320, 2, 24
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    sm, ref = cd.StereoMatching(cfg, max_batch=6), cd.StereoMatching(cfg, max_batch=6)
    L, R = _lane_inputs(12, H, W, Dd * K, K, 9100)
    g = (torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
    rgb = [syn.random_rgb_pair(H, W, Dd * K, K, 9200 + i) for i in range(4)]
    c = (torch.from_numpy(np.stack([p[0] for p in rgb])).cuda(), torch.from_numpy(np.stack([p[1] for p in rgb])).cuda())
    calls = [(g, 0, 1), (g, 1, 3), (c, 0, 2), (g, 4, 2), (c, 2, 1), (c, 3, 1), (g, 6, 3)]
    want = [ref.compute_disparity_map_batch(t[0][a:a + n], t[1][a:a + n]).clone() for t, a, n in calls]
    want_ref = ref.intermediate(N.STAGE_REFINED, 2).clone()
    want_costs = ref.intermediate(N.STAGE_MBM_COSTS, 1).clone()
    torch.cuda.synchronize()
    for rep in range(3):
        outs = [torch.zeros((n, H, W), device="cuda") for _, _, n in calls]
        torch.cuda.synchronize()
        for k, (t, a, n) in enumerate(calls):
            sm.compute_disparity_map_batch(t[0][a:a + n], t[1][a:a + n], out=outs[k], engine_streams=True)
        got_ref = sm.intermediate(N.STAGE_REFINED, 2).clone()          # joins by itself
        got_costs = sm.intermediate(N.STAGE_MBM_COSTS, 1).clone()
        sm.join()
        torch.cuda.synchronize()
        for k in range(len(calls)):
            assert torch.equal(outs[k], want[k]), f"rep {rep} call {k}"
        assert torch.equal(got_ref, want_ref) and torch.equal(got_costs, want_costs), f"rep {rep}: intermediates of the last call"


def test_engine_stream_calls_that_share_an_output_are_ordered(cd):
    """Two consecutive small engine-stream calls alternate between the lanes; when they write the SAME output memory the
    later call must win, as on one stream (the engine orders calls whose `out` ranges overlap).  A heavy call
    (3 RGB pairs, exact-order kernel) followed by a light one (1 gray pair) into the same tensor: without the ordering the
    light call finishes first and the heavy one overwrites it.  Also: overlapping but not identical ranges, a ring of
    distinct outputs (no ordering needed, results intact), and out=None -- the wrapper alternates halves of its batch buffer,
    so the previous result stays valid while the next call is in flight."""
    H, W, K, Dd = 96, 320, 2, 24
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    sm, ref = cd.StereoMatching(cfg, max_batch=6), cd.StereoMatching(cfg, max_batch=6)
    L, R = _lane_inputs(8, H, W, Dd * K, K, 9300)
    g = (torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
    rgb = [syn.random_rgb_pair(H, W, Dd * K, K, 9400 + i) for i in range(3)]
    c = (torch.from_numpy(np.stack([p[0] for p in rgb])).cuda(), torch.from_numpy(np.stack([p[1] for p in rgb])).cuda())
    want_heavy = ref.compute_disparity_map_batch(c[0], c[1]).clone()
    want_light = [ref.compute_disparity_map_batch(g[0][i:i + 1], g[1][i:i + 1]).clone() for i in range(8)]
    torch.cuda.synchronize()
    for rep in range(4):
        out = torch.zeros((3, H, W), device="cuda")
        torch.cuda.synchronize()
        sm.compute_disparity_map_batch(c[0], c[1], out=out, engine_streams=True)                       # lane A: long
        sm.compute_disparity_map_batch(g[0][rep:rep + 1], g[1][rep:rep + 1], out=out[1:2], engine_streams=True)   # lane B: short, overlaps pair 1
        sm.join()
        torch.cuda.synchronize()
        assert torch.equal(out[0], want_heavy[0]) and torch.equal(out[2], want_heavy[2]), f"rep {rep}"
        assert torch.equal(out[1], want_light[rep][0]), f"rep {rep}: the later call must win"
    # a ring of distinct outputs: nothing to order, every result intact
    ring = [torch.zeros((1, H, W), device="cuda") for _ in range(8)]
    torch.cuda.synchronize()
    for i in range(8):
        sm.compute_disparity_map_batch(g[0][i:i + 1], g[1][i:i + 1], out=ring[i], engine_streams=True)
    sm.join()
    torch.cuda.synchronize()
    for i in range(8):
        assert torch.equal(ring[i], want_light[i]), f"ring {i}"
    # out=None: consecutive calls get alternate halves of the wrapper's batch buffer
    a = sm.compute_disparity_map_batch(c[0], c[1], engine_streams=True)
    b = sm.compute_disparity_map_batch(g[0][:1], g[1][:1], engine_streams=True)
    assert a.data_ptr() != b.data_ptr()
    sm.join()
    torch.cuda.synchronize()
    assert torch.equal(a, want_heavy) and torch.equal(b, want_light[0])
    # ... and the very same default output for calls too large to alternate (later call wins)
    big = cd.StereoMatching(cfg, max_batch=3)
    a = big.compute_disparity_map_batch(c[0], c[1], engine_streams=True)
    b = big.compute_disparity_map_batch(g[0][:2], g[1][:2], engine_streams=True)
    assert a.data_ptr() == b.data_ptr()
    big.join()
    torch.cuda.synchronize()
    assert torch.equal(b[0], want_light[0][0]) and torch.equal(b[1], want_light[1][0]) and torch.equal(a[2], want_heavy[2])


def test_engines_of_one_device_share_the_lane_streams(cd):
    """The stream lanes are one pair of streams per device, shared by every engine on it (two hardware queues of their own,
    whatever else the process created).  Two engines of different shapes submit engine-stream calls alternately, without
    host synchronisation in between; each engine's join covers its own calls; destroying one engine leaves the other's
    lanes alive."""
    import gc
    H1, W1, H2, W2, K, Dd = 64, 200, 96, 320, 2, 16
    c1 = cd.StereoMatchingConfiguration(height=H1, width=W1, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    c2 = cd.StereoMatchingConfiguration(height=H2, width=W2, downscale_factor=K, min_disparity=0, max_disparity=Dd * K - 1)
    n = 36
    L1, R1 = _lane_inputs(n, H1, W1, Dd * K, K, 8100)
    L2, R2 = _lane_inputs(n, H2, W2, Dd * K, K, 8200)
    a = (torch.from_numpy(L1).cuda(), torch.from_numpy(R1).cuda())
    b = (torch.from_numpy(L2).cuda(), torch.from_numpy(R2).cuda())
    ref1, ref2 = cd.StereoMatching(c1, max_batch=n), cd.StereoMatching(c2, max_batch=n)
    want1, want1r = ref1.compute_disparity_map_batch(*a).clone(), ref1.compute_disparity_map_batch(a[1], a[0]).clone()
    want2, want2r = ref2.compute_disparity_map_batch(*b).clone(), ref2.compute_disparity_map_batch(b[1], b[0]).clone()
    torch.cuda.synchronize()
    e1, e2 = cd.StereoMatching(c1, max_batch=n, overlap_min_pairs=16), cd.StereoMatching(c2, max_batch=n, overlap_min_pairs=16)
    o1 = [torch.zeros((n, H1, W1), device="cuda") for _ in range(4)]
    o2 = [torch.zeros((n, H2, W2), device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    for k in range(4):
        e1.compute_disparity_map_batch(*(a if k % 2 == 0 else a[::-1]), out=o1[k], engine_streams=True)
        e2.compute_disparity_map_batch(*(b if k % 2 == 0 else b[::-1]), out=o2[k], engine_streams=True)
    e1.join()
    got1 = [o.clone() for o in o1]
    e2.join()
    got2 = [o.clone() for o in o2]
    torch.cuda.synchronize()
    for k in range(4):
        assert torch.equal(got1[k], want1 if k % 2 == 0 else want1r), f"engine 1 call {k}"
        assert torch.equal(got2[k], want2 if k % 2 == 0 else want2r), f"engine 2 call {k}"
    del e1
    gc.collect()                                                   # smx_destroy of engine 1: the pool keeps the streams for engine 2
    out = e2.compute_disparity_map_batch(*b, engine_streams=True)
    e2.join()
    torch.cuda.synchronize()
    assert torch.equal(out, want2)


def test_call_counter_wrap_clears_the_flags_and_keeps_the_bits(cd, oracle_omp, monkeypatch):
    """Per-pair device flags are stamped with a call counter instead of being cleared per call; when the counter wraps
    (once per 2^31 calls) the engine clears them behind a device synchronisation.  SMX_TEST_EPOCH_START puts a fresh engine
    six calls before the wrap: on-grid and off-grid pairs (whose flags differ), lanes and caller-stream calls across the
    wrap, all compared with the oracle."""
    H, W, K, D = 96, 320, 2, 32
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    n = 40
    L, R = _lane_inputs(n, H, W, D, K, 7000)
    Lo = L.copy()
    Lo[::3] += 0.25                                            # every third pair off the grid
    want_on = [oracle_omp.run(ocfg, L[i], R[i]) for i in (0, 1, 39)]
    want_off = [oracle_omp.run(ocfg, Lo[i], R[i]) for i in (0, 1, 39)]
    tl, tlo, tr = torch.from_numpy(L).cuda(), torch.from_numpy(Lo).cuda(), torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    monkeypatch.setenv("SMX_TEST_EPOCH_START", str(0x7fffffff - 6))
    sm = cd.StereoMatching(cfg, max_batch=n, overlap_min_pairs=16)
    monkeypatch.delenv("SMX_TEST_EPOCH_START")
    for k in range(14):                                         # the wrap happens at the 7th call
        off = k % 2 == 1
        out = sm.compute_disparity_map_batch(tlo if off else tl, tr, engine_streams=(k % 3 != 0))
        sm.join()
        got = out.cpu().numpy()
        for j, i in enumerate((0, 1, 39)):
            assert np.array_equal(got[i], (want_off if off else want_on)[j]), f"call {k} pair {i}"


# ----------------------------------------------------------------------------- filtered exact-order route
def test_filtered_exact_order_route_is_bit_exact(cd, oracle_omp):
    """RGB batches take the filtered route (k_match_filter.h): a cheap pass on the inputs rounded to the grid marks the
    disparities that can hold the maximum, only those are evaluated in the reference's order.  Same bits as the
    dense kernel and as the oracle -- on textured pairs (few candidates), pure noise and flat images (every
    disparity a candidate), gray levels at the ends of the range, and a pair whose gray leaves [0, 255] (f32 RGB:
    dense kernel through the device-side range flag)."""
    from cuda_depth import _native as N
    H, W, K, D = 150, 700, 2, 64
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    rng = np.random.default_rng(77)
    pairs = []
    for i in range(3):
        pairs.append(syn.random_rgb_pair(H, W, D, K, 40 + i))
    nl, nr = rng.integers(0, 256, (3, H, W)).astype(np.float32), rng.integers(0, 256, (3, H, W)).astype(np.float32)
    pairs.append((nl, nr))                                                   # noise: nothing stands out
    pairs.append((np.full((3, H, W), 37.0, np.float32), np.full((3, H, W), 37.0, np.float32)))     # flat: all equal
    pairs.append((np.where(nl > 128, 255.0, 0.0).astype(np.float32), np.where(nr > 128, 255.0, 0.0).astype(np.float32)))
    sl, sr = syn.make_slanted_pair(H, W, D, K, 3)[:2]
    w3 = np.array([0.9, 1.0, 0.8], np.float32)[:, None, None]
    pairs.append((np.rint(sl[None] * w3).astype(np.float32), np.rint(sr[None] * w3).astype(np.float32)))
    # periodic texture (period 16 columns): every disparity congruent to the true one modulo 8 pooled columns costs the
    # same up to a little noise -- near-ties and exact ties, the first maximum has to win
    tile = rng.integers(0, 256, (3, H, 16)).astype(np.float32)
    per_l = np.tile(tile, (1, 1, W // 16 + 1))[:, :, :W]
    per_r = np.roll(per_l, -10, axis=2).copy()
    spots = rng.random((3, H, W)) < 0.002
    per_r[spots] = np.clip(per_r[spots] + rng.integers(-3, 4, int(spots.sum())), 0, 255)
    pairs.append((per_l, per_r))
    pairs.append((per_l, np.roll(per_l, -10, axis=2).copy()))                # ... and exact ties only
    bad = syn.random_rgb_pair(H, W, D, K, 50)
    bad = (bad[0] * 1.5 - 20.0, bad[1] * 1.5 - 20.0)                         # gray leaves [0, 255]: dense kernel
    pairs.append(bad)
    uniq = len(pairs)
    n = 48                                                                  # enough workgroups for the tall-band shape (no small-call path)
    L = np.stack([pairs[i % uniq][0] for i in range(n)]).astype(np.float32)
    R = np.stack([pairs[i % uniq][1] for i in range(n)]).astype(np.float32)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    filt = cd.StereoMatching(cfg, max_batch=n, exact_filter=1)               # always filtered (the default route follows the content)
    dense = cd.StereoMatching(cfg, max_batch=n, exact_filter=-1)
    filt.profile_begin(1)
    of = filt.compute_disparity_map_batch(tl, tr).cpu().numpy()
    assert filt.profile_end()["match_fast"][1] == 1                          # the filter kernel ran
    dense.profile_begin(1)
    od = dense.compute_disparity_map_batch(tl, tr).cpu().numpy()
    assert dense.profile_end()["match_fast"][1] == 0
    assert np.array_equal(of, od)
    for i in range(n):
        for st in (N.STAGE_WTA, N.STAGE_MBM_COSTS, N.STAGE_REFINED):
            assert torch.equal(filt.intermediate(st, i), dense.intermediate(st, i)), f"pair {i} stage {st}"
    for i in range(uniq):
        assert np.array_equal(of[i], oracle_omp.run(ocfg, pairs[i][0], pairs[i][1])), f"pair {i}"
    # a second call on the same engine: the candidate bits of the first were cleared
    of2 = filt.compute_disparity_map_batch(tr, tl).cpu().numpy()
    assert np.array_equal(of2, dense.compute_disparity_map_batch(tr, tl).cpu().numpy())
    # u8 RGB entry (always in range) and the stream lanes
    l8, r8 = torch.from_numpy(np.clip(L[:7 * 3], 0, 255).astype(np.uint8)).cuda(), torch.from_numpy(np.clip(R[:7 * 3], 0, 255).astype(np.uint8)).cuda()
    want8 = dense.compute_disparity_map_batch(l8, r8).clone()
    torch.cuda.synchronize()
    assert torch.equal(filt.compute_disparity_map_batch(l8, r8), want8)
    lanes = cd.StereoMatching(cfg, max_batch=n, overlap_min_pairs=12, exact_filter=1)   # halves of 24: the second half takes the small-call (dense) path
    got = lanes.compute_disparity_map_batch(tl, tr, engine_streams=True)
    lanes.join()
    assert np.array_equal(got.cpu().numpy(), od)


def test_filtered_route_with_min_disparity(cd, oracle_omp):
    """min_disparity > 0 (the reference's default configuration has 75): the filtered route delivers the arg-max, the
    sparse capture kernel the three costs step 6 reads (oracle rule S6).  Same bits as the dense kernel and the oracle."""
    from cuda_depth import _native as N
    H, W, K = 150, 700, 2
    for dmin, dmax in ((20, 83), (74, 201)):
        cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
        ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
        pairs = [syn.random_rgb_pair(H, W, dmax + 1, K, 60 + i, dmin=dmin) for i in range(2)]
        rng = np.random.default_rng(5)
        pairs.append((rng.integers(0, 256, (3, H, W)).astype(np.float32), rng.integers(0, 256, (3, H, W)).astype(np.float32)))
        n = 48
        L = np.stack([pairs[i % 3][0] for i in range(n)]).astype(np.float32)
        R = np.stack([pairs[i % 3][1] for i in range(n)]).astype(np.float32)
        tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        filt = cd.StereoMatching(cfg, max_batch=n, exact_filter=1)
        dense = cd.StereoMatching(cfg, max_batch=n, exact_filter=-1)
        filt.profile_begin(1)
        of = filt.compute_disparity_map_batch(tl, tr).cpu().numpy()
        assert filt.profile_end()["match_fast"][1] == 1                      # the filter kernel ran
        od = dense.compute_disparity_map_batch(tl, tr).cpu().numpy()
        assert np.array_equal(of, od), f"dmin={dmin}"
        for i in (0, 1, 2, 47):
            for st in (N.STAGE_WTA, N.STAGE_MBM_COSTS, N.STAGE_REFINED):
                assert torch.equal(filt.intermediate(st, i), dense.intermediate(st, i)), f"dmin={dmin} pair {i} stage {st}"
        for i in range(3):
            assert np.array_equal(of[i], oracle_omp.run(ocfg, pairs[i][0], pairs[i][1])), f"dmin={dmin} pair {i}"


def test_content_aware_choice_between_filtered_and_dense_exact_order(cd, oracle_omp):
    """exact_filter = 0 (default): the sparse kernel reports the candidate density of every filtered launch without a
    synchronisation.  Noise (every disparity a candidate everywhere) sends the engine to the dense kernel, which it
    leaves again once a probe sees structured content -- and whichever route a call takes, the bits are the dense
    kernel's and the oracle's (multi_block_matching_cost_aggregation.cu:54-88, wta_disparity_selection.cu:22-30)."""
    H, W, K, D = 150, 700, 2, 64
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    n = 48
    rng = np.random.default_rng(9)
    noise = (rng.integers(0, 256, (3, H, W)).astype(np.float32), rng.integers(0, 256, (3, H, W)).astype(np.float32))
    tex = [syn.random_rgb_pair(H, W, D, K, 80 + i) for i in range(2)]
    Ln, Rn = (torch.from_numpy(np.stack([noise[k]] * n)).cuda() for k in (0, 1))
    Lt, Rt = (torch.from_numpy(np.stack([tex[i % 2][k] for i in range(n)])).cuda() for k in (0, 1))
    sm = cd.StereoMatching(cfg, max_batch=n)
    dense = cd.StereoMatching(cfg, max_batch=n, exact_filter=-1)
    want_n = dense.compute_disparity_map_batch(Ln, Rn).clone()
    want_t = dense.compute_disparity_map_batch(Lt, Rt).clone()
    assert np.array_equal(want_n[0].cpu().numpy(), oracle_omp.run(ocfg, noise[0], noise[1]))
    assert np.array_equal(want_t[1].cpu().numpy(), oracle_omp.run(ocfg, tex[1][0], tex[1][1]))
    info = sm.route_info()
    assert info["filter_available"] == 1 and info["route_dense"] == 0 and info["compute_units"] >= 1
    # structured content: stays filtered, low density
    for _ in range(3):
        assert torch.equal(sm.compute_disparity_map_batch(Lt, Rt), want_t)
        torch.cuda.synchronize()
    info = sm.route_info()
    assert info["route_dense"] == 0 and 0.0 < info["candidate_density"] < 0.40, info
    # noise: the first call is still filtered, its report flips the route
    assert torch.equal(sm.compute_disparity_map_batch(Ln, Rn), want_n)
    torch.cuda.synchronize()
    info = sm.route_info()
    assert info["route_dense"] == 1 and info["candidate_density"] > 0.9, info
    sm.profile_begin(1)
    assert torch.equal(sm.compute_disparity_map_batch(Ln, Rn), want_n)
    assert sm.profile_end()["match_fast"][1] == 0                    # no filter kernel: the dense route
    # content changes back: within probe_period calls a probe reports a low density and the filter returns
    calls = 0
    while sm.route_info()["route_dense"] == 1 and calls < 40:
        assert torch.equal(sm.compute_disparity_map_batch(Lt, Rt), want_t)
        torch.cuda.synchronize()
        calls += 1
    assert sm.route_info()["route_dense"] == 0 and calls <= 20, (calls, sm.route_info())
    # the stream lanes report per lane (96 pairs: halves of 48, large enough for the filtered route)
    lanes = cd.StereoMatching(cfg, max_batch=2 * n, overlap_min_pairs=16)
    for l, r, want in ((Ln, Rn, want_n), (Ln, Rn, want_n), (Lt, Rt, want_t)):
        l2, r2 = torch.cat([l, l]), torch.cat([r, r])
        torch.cuda.synchronize()                                     # engine-stream calls need complete inputs
        out = lanes.compute_disparity_map_batch(l2, r2, engine_streams=True)
        lanes.join()
        torch.cuda.synchronize()
        assert torch.equal(out[:n], want) and torch.equal(out[n:], want)
    assert lanes.route_info()["route_dense"] == 1


def test_content_aware_choice_between_the_sparse_and_the_dense_form_of_the_fast_kernel(cd, oracle_omp):
    """On-grid gray batches, min_disparity = 0: the fast kernel's sparse second pass reports (no synchronisation) which share
    of the disparity range it revisited per window.  Banded surfaces keep the sparse form; noise -- every disparity wins
    somewhere in every window -- sends the engine to the form that keeps the winner's two neighbours during pass 1
    (k_match_fast<..., DENSE>), which it leaves again once a probe sees smooth content.  Whichever form a call takes, the
    bits are the oracle's (multi_block_matching_cost_aggregation.cu:54-88, wta_disparity_selection.cu:22-30,
    secondary_matching.cu:56-58)."""
    H, W, K, D = 150, 700, 2, 63                     # odd range: the last march of pass 1 is a single disparity
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    n = 48
    noise = [syn.make_noise_pair(H, W, 30 + i) for i in range(3)]
    # smooth content: one fronto-parallel surface per pair (the banded synthetic pairs hold too many disparities per window at
    # this small size to stay clearly below the switching threshold)
    band = []
    for i in range(3):
        l = syn.make_pair(H, W, D, K, 40 + i)[0]
        band.append((l, np.roll(l, -K * (5 + 9 * i), axis=1).copy()))
    Ln, Rn = (torch.from_numpy(np.stack([noise[i % 3][k] for i in range(n)])).cuda() for k in (0, 1))
    Lb, Rb = (torch.from_numpy(np.stack([band[i % 3][k] for i in range(n)])).cuda() for k in (0, 1))
    want_n = [oracle_omp.run(ocfg, *noise[i]) for i in range(3)]
    want_b = [oracle_omp.run(ocfg, *band[i]) for i in range(3)]

    def check(out, want):
        o = out.cpu().numpy()
        for i in (0, 1, 2, n - 1):
            assert np.array_equal(o[i], want[i % 3]), i

    sm = cd.StereoMatching(cfg, max_batch=n)
    assert sm.route_info()["fast_dense"] == 0
    for _ in range(3):                               # smooth content: the sparse form stays
        check(sm.compute_disparity_map_batch(Lb, Rb), want_b)
        torch.cuda.synchronize()
    assert sm.route_info()["fast_dense"] == 0
    check(sm.compute_disparity_map_batch(Ln, Rn), want_n)       # still sparse; its report flips the choice
    torch.cuda.synchronize()
    assert sm.route_info()["fast_dense"] == 1
    for _ in range(3):
        check(sm.compute_disparity_map_batch(Ln, Rn), want_n)   # dense form
        torch.cuda.synchronize()
    assert sm.route_info()["fast_dense"] == 1
    calls = 0                                        # content changes back: a probe (every 16 calls at first) reports it
    while sm.route_info()["fast_dense"] == 1 and calls < 40:
        check(sm.compute_disparity_map_batch(Lb, Rb), want_b)
        torch.cuda.synchronize()
        calls += 1
    assert sm.route_info()["fast_dense"] == 0 and calls <= 20, (calls, sm.route_info())
    # the stream lanes: both halves report, both halves switch
    lanes = cd.StereoMatching(cfg, max_batch=2 * n, overlap_min_pairs=16)
    for l, r, want in ((Ln, Rn, want_n), (Ln, Rn, want_n), (Ln, Rn, want_n), (Lb, Rb, want_b)):
        l2, r2 = torch.cat([l, l]), torch.cat([r, r])
        torch.cuda.synchronize()
        out = lanes.compute_disparity_map_batch(l2, r2, engine_streams=True)
        lanes.join()
        torch.cuda.synchronize()
        check(out[:n], want)
        check(out[n:], want)
    assert lanes.route_info()["fast_dense"] == 1


@pytest.mark.parametrize("H,W,K,D", [(240, 320, 1, 32), (150, 400, 2, 48)])
def test_single_f32_gray_calls_follow_the_grid_hint(cd, oracle_omp, H, W, K, D):
    """AUTO, one f32 gray pair per call: on-grid input takes ONE aggregation launch that branches on the device flag
    (its off-grid branch is the small generic exact-order body); once a call has reported off-grid input
    (k_refine_auto -> pinned host word, no synchronisation) the next calls take the two gated launches with the
    disparity-split register-tiled kernel.  Whatever the plan, the bits are the oracle's
    (multi_block_matching_cost_aggregation.cu:54-88, wta_disparity_selection.cu:22-30)."""
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    l, r = syn.make_pair(H, W, D, K, 31)[:2]
    lo = (l + 0.3).astype(np.float32)                        # off the grid (and not integer-valued: float step 6)
    want_on, want_off = oracle_omp.run(ocfg, l, r), oracle_omp.run(ocfg, lo, r)
    tl, tlo, tr = torch.from_numpy(l).cuda(), torch.from_numpy(lo).cuda(), torch.from_numpy(r).cuda()
    sm = cd.StereoMatching(cfg)
    assert sm.route_info()["offgrid_hint"] == -1            # no report yet
    seq = ["on", "on", "off", "off", "off", "on", "on", "off", "on"]
    hints = []
    for k, kind in enumerate(seq):
        sm.profile_begin(1)
        out = sm.compute_disparity_map_gray(tlo if kind == "off" else tl, tr).cpu().numpy()      # (.cpu() synchronises: the report has arrived)
        prof = sm.profile_end()
        assert np.array_equal(out, want_off if kind == "off" else want_on), f"call {k} ({kind})"
        hints.append(sm.route_info()["offgrid_hint"])
        expect_two = k == 0 or seq[k - 1] == "off"          # plan of call k follows what call k-1 reported; nothing reported yet: two
        assert (prof["match_exact"][1] == 1) == expect_two, f"call {k}: launches {prof}"
    assert hints == [1 if kind == "off" else 0 for kind in seq]
    # the same without any synchronisation between the calls: the plan lags, the bits do not change
    outs = [torch.empty((H, W), device="cuda") for _ in seq]
    for k, kind in enumerate(seq):
        outs[k].copy_(sm.compute_disparity_map_gray(tlo if kind == "off" else tl, tr))
    torch.cuda.synchronize()
    for k, kind in enumerate(seq):
        assert np.array_equal(outs[k].cpu().numpy(), want_off if kind == "off" else want_on), f"async call {k} ({kind})"


@pytest.mark.parametrize("H,W,K,D", [(150, 400, 2, 48), (375, 1242, 2, 128), (240, 320, 1, 32), (188, 622, 2, 192)])
def test_one_launch_auto_kernel_with_mixed_pairs(cd, oracle_omp, H, W, K, D):
    """The one-launch AUTO kernel (k_match_auto.h) on calls of 2 - 4 pairs of which SOME are off the grid: per pair the
    workgroups of the launch run the fast body or become the disparity-split exact-order kernel, whose slices the last
    workgroup of a tile to arrive merges (tickets per pair slot and tile, reset by the merger).  Three such calls in a row
    on the caller's stream and two on the stream lanes (the tickets and slice records of the two pair-slot halves), every
    pair against the oracle (multi_block_matching_cost_aggregation.cu:54-88, wta_disparity_selection.cu:22-30)."""
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    on = [syn.make_pair(H, W, D, K, 60 + i)[:2] for i in range(4)]
    off = [((l + np.float32(0.3)).astype(np.float32), r) for l, r in on]          # off the grid, not integer-valued
    want_on = [oracle_omp.run(ocfg, l, r) for l, r in on]
    want_off = [oracle_omp.run(ocfg, l, r) for l, r in off]
    sm = cd.StereoMatching(cfg, max_batch=8)

    def batch(kinds):
        L = np.stack([(off if k else on)[i][0] for i, k in enumerate(kinds)])
        R = np.stack([(off if k else on)[i][1] for i, k in enumerate(kinds)])
        return torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), [(want_off if k else want_on)[i] for i, k in enumerate(kinds)]

    # an on-grid call of <= 4 pairs first: its report (pair 0 on the grid) moves the next calls to the one-launch kernel
    tl, tr, want = batch([0, 0, 0, 0])
    out = sm.compute_disparity_map_batch(tl, tr).cpu().numpy()
    assert sm.route_info()["offgrid_hint"] == 0
    for kinds in ([0, 1, 0, 1], [0, 1, 1, 0], [0, 0, 1], [0, 1]):
        tl, tr, want = batch(kinds)
        sm.profile_begin(1)
        out = sm.compute_disparity_map_batch(tl, tr).cpu().numpy()
        prof = sm.profile_end()
        assert prof["match_exact"][1] == 0 and prof["match_fast"][1] == 1, prof       # ONE aggregation launch
        for i in range(len(kinds)):
            assert np.array_equal(out[i], want[i]), f"{kinds}: pair {i}"
        assert sm.route_info()["offgrid_hint"] == 0                               # pair 0 was on the grid
    # the same on the stream lanes: consecutive small calls alternate between the lanes and the halves of the pair slots
    outs = []
    torch.cuda.synchronize()
    for kinds in ([0, 1, 0, 1], [0, 1, 1, 0], [0, 1, 0, 1]):
        tl, tr, want = batch(kinds)
        torch.cuda.synchronize()
        o = torch.zeros((len(kinds), H, W), device="cuda")
        sm.compute_disparity_map_batch(tl, tr, out=o, engine_streams=True)
        outs.append((o, want, kinds))
    sm.join()
    torch.cuda.synchronize()
    for o, want, kinds in outs:
        got = o.cpu().numpy()
        for i in range(len(kinds)):
            assert np.array_equal(got[i], want[i]), f"lanes {kinds}: pair {i}"


@pytest.mark.parametrize("H,W,K,D", [(94, 260, 2, 32), (75, 131, 1, 16), (123, 517, 4, 64), (375, 1242, 2, 128)])
def test_fused_refine_fill_launch_is_bit_exact(cd, oracle_omp, monkeypatch, H, W, K, D):
    """SMX_FUSED_REFINE_FILL=1 (opt-in, k_refine_fill.h): step 6 and the fills of a gray batch in one launch,
    one halo row and column recomputed per 64 x 16 tile -- the same bits as the two launches and as the oracle
    (secondary_matching.cu:24-71, upscale_disparity_vertical_fill.cu:17-51, horizontal_disparity_fill.cu:16-40),
    for sizes that are not multiples of the tile or of K, the u8 entry, and an AUTO batch with an off-grid pair."""
    from cuda_depth import _native as N
    n = 6
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    L = np.stack([syn.make_pair(H, W, D, K, 700 + i)[0] for i in range(n)])
    R = np.stack([syn.make_pair(H, W, D, K, 700 + i)[1] for i in range(n)])
    L[1], R[1] = syn.make_noise_pair(H, W, 3)                  # colour branches of both fills, per-pixel step-6 route
    L[2], R[2] = syn.make_slanted_pair(H, W, D, K, 5)[:2]
    Lf = L.copy()
    Lf[4] += 0.25                                              # off the grid: float step 6 for this pair (AUTO)
    if not cd.build_features()["experimental"]:
        pytest.skip("library built without SMX_EXPERIMENTAL (python stereo-depth_amd/build.py --experimental)")
    monkeypatch.delenv("SMX_FUSED_REFINE_FILL", raising=False)
    sm = cd.StereoMatching(cfg, max_batch=n)
    monkeypatch.setenv("SMX_FUSED_REFINE_FILL", "1")                 # read once, when the engine is created
    smf = cd.StereoMatching(cfg, max_batch=n)
    for tl, tr, src in ((torch.from_numpy(Lf).cuda(), torch.from_numpy(R).cuda(), Lf),
                        (torch.from_numpy(L.astype(np.uint8)).cuda(), torch.from_numpy(R.astype(np.uint8)).cuda(), L)):
        want = sm.compute_disparity_map_batch(tl, tr).clone()
        want_ref = [sm.intermediate(N.STAGE_REFINED, i).clone() for i in range(n)]
        smf.profile_begin(1)
        got = smf.compute_disparity_map_batch(tl, tr).clone()
        prof = smf.profile_end()
        assert prof["fill"][1] == 0 and prof["refine"][1] == 1          # the fused launch did run
        assert torch.equal(got, want)
        for i in range(n):
            assert torch.equal(smf.intermediate(N.STAGE_REFINED, i), want_ref[i]), f"refined, pair {i}"
        for i in (0, 1, 2, 4):
            assert np.array_equal(got[i].cpu().numpy(), oracle_omp.run(ocfg, src[i], R[i])), f"pair {i}"
    monkeypatch.delenv("SMX_FUSED_REFINE_FILL", raising=False)
