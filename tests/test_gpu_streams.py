"""The C-ABI calls only enqueue work on the caller's stream (include/subpixal_hip.h,
"Conventions"): they must be capturable into a HIP graph once the constant tables
exist (spx_prepare), and independent on independent streams.  GPU only."""
import pytest

pytestmark = pytest.mark.gpu


def _pairs(seed, count=2048, n=64):
    from subpixal_amd import synth
    return synth.gaussian_pairs(count, n, seed=seed)


def test_graph_capture_and_replay():
    import torch
    from subpixal_amd import _ffi, cc, device
    device.init()
    _ffi.check(_ffi.load().spx_prepare(10))        # tables built before the capture
    ref, img, _ = _pairs(11)
    ref2, img2, truth2 = _pairs(12)
    eager2 = cc.xcorr_refine_batch(ref2, img2, upsample=10)
    torch.cuda.synchronize()

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out, status = cc.xcorr_refine_batch(ref, img, upsample=10, return_status=True)
    # replay on new data written into the captured input buffers
    ref.copy_(ref2)
    img.copy_(img2)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager2)                # same kernel, same inputs: bitwise
    assert int(status.abs().max()) == 0
    assert float((out - truth2).abs().max()) < 2e-4
    # and once more with the first inputs
    ref1, img1, truth1 = _pairs(11)
    ref.copy_(ref1)
    img.copy_(img1)
    graph.replay()
    torch.cuda.synchronize()
    assert float((out - truth1).abs().max()) < 2e-4


def test_graph_capture_reference_mode():
    import torch
    import datagen
    from subpixal_amd import cc, device
    device.init()
    ref, im4, _ = datagen.dither_batch(3, 64, 48)
    r = torch.as_tensor(ref, device='cuda')
    m = torch.as_tensor(im4, device='cuda')
    eager = cc.find_displacement_batch(r, m, cc_type='NCC')
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = cc.find_displacement_batch(r, m, cc_type='NCC')
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


@pytest.mark.parametrize('n,up', [(32, 7), (80, 9), (96, 11), (128, 21), (150, 13)])
def test_capture_every_kernel_family_without_prior_eager_call(n, up):
    """spx_prepare(upsample) builds the tables of EVERY kernel family and raises
    every kernel's LDS limit, so the very first launch at a (shape, upsample) can already sit
    inside a stream capture -- 32 tile, 64 tile's fold path, period-192 path; pair mode and
    reference mode; float32 and float64 inputs.  Each upsample here is used nowhere else in the
    suite, so no earlier eager call can have built its tables."""
    import torch
    import datagen
    from subpixal_amd import cc, device
    device.prepare(up, shape=(n, n) if n > 128 else None)      # general path: size-dependent tables
    ref, img, truth = _pairs(100 + n, count=512 if n <= 128 else 64, n=n)
    r5, m4, _ = datagen.dither_batch(n, 8, n)
    r5 = torch.as_tensor(r5, device='cuda')
    m4 = torch.as_tensor(m4, device='cuda')
    ref64, img64 = ref[:64].double(), img[:64].double()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out, status = cc.xcorr_refine_batch(ref, img, upsample=up, return_status=True)
        out64 = cc.xcorr_refine_batch(ref64, img64, upsample=up)
        d5 = cc.find_displacement_batch(r5, m4, cc_type='NCC')
        d5_64 = cc.find_displacement_batch(r5.double(), m4.double(), cc_type='ZNCC')
    graph.replay()
    torch.cuda.synchronize()
    assert int(status.abs().max()) == 0
    if n > 32:                       # (the generator's sigma 4..6 spots are clipped by a 32 px tile)
        assert float((out - truth).abs().max()) < 1e-3
    assert float((out64 - out[:64]).abs().max()) < 1e-4
    eager = cc.xcorr_refine_batch(ref, img, upsample=up)
    assert torch.equal(out, eager)
    assert torch.equal(d5, cc.find_displacement_batch(r5, m4, cc_type='NCC'))
    assert torch.equal(d5_64, cc.find_displacement_batch(r5.double(), m4.double(), cc_type='ZNCC'))


def test_shutdown_frees_and_rebuilds_tables():
    import torch
    from subpixal_amd import cc, device
    ref, img, truth = _pairs(31, count=256)
    a = cc.xcorr_refine_batch(ref, img, upsample=10)
    torch.cuda.synchronize()
    device.shutdown()
    b = cc.xcorr_refine_batch(ref, img, upsample=10)       # tables rebuilt on first use
    assert torch.equal(a, b)


def test_two_streams_are_independent():
    import torch
    from subpixal_amd import cc, device
    device.init()
    ref_a, img_a, truth_a = _pairs(21, count=20000)
    ref_b, img_b, truth_b = _pairs(22, count=20000)
    base_a = cc.xcorr_refine_batch(ref_a, img_a, upsample=10)
    base_b = cc.xcorr_refine_batch(ref_b, img_b, upsample=2)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sa):
        out_a = cc.xcorr_refine_batch(ref_a, img_a, upsample=10)
    with torch.cuda.stream(sb):
        out_b = cc.xcorr_refine_batch(ref_b, img_b, upsample=2)
    sa.synchronize()
    sb.synchronize()
    assert torch.equal(out_a, base_a)
    assert torch.equal(out_b, base_b)
    assert float((out_a - truth_a).abs().max()) < 2e-4
