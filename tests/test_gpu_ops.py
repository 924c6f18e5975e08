"""Parity of every HIP kernel (through the C ABI) against the CPU oracle -- runs on MI355X only.
Tolerance: north_star asks 1e-3 relative fp32; kernels are checked at 2e-4 of the output scale vs the
fp64 oracle (observed ~1e-6)."""
import numpy as np
import pytest
import torch

from oracle import np_ops as N
from oracle import torch_ops as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    import dl4ds_amd.ops as o
    return o


def close(a, ref, tol=2e-4):
    ref = np.asarray(ref, np.float64)
    a = np.asarray(a, np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-6)
    err = np.abs(a - ref).max() / scale
    assert err < tol, f'max rel err {err:.3e}'


rng = np.random.default_rng(123)


def R(*s):
    return rng.standard_normal(s).astype(np.float32)


CONV_CASES = [
    # N, H, W, Cin, Cout, KS
    (2, 13, 19, 1, 8, 3), (1, 16, 16, 8, 8, 3), (2, 9, 33, 3, 16, 3), (1, 20, 17, 8, 1, 3),
    (1, 8, 16, 48, 192, 3), (1, 24, 24, 50, 40, 3), (2, 16, 32, 40, 48, 3), (1, 12, 12, 192, 48, 3),
    (2, 17, 16, 8, 48, 1), (1, 16, 16, 48, 8, 1), (1, 10, 10, 24, 200, 1), (1, 16, 16, 6, 12, 5),
    (1, 9, 9, 16, 32, 5), (1, 8, 8, 130, 128, 3), (1, 6, 6, 256, 100, 3),
    # Cout that no NT divides (conv_stream.hip: zero-padded filter copy, NT = 3 on 40 / 44 couts), 3x3 and 1x1
    (2, 16, 20, 32, 40, 3), (1, 16, 16, 40, 40, 3), (1, 24, 16, 48, 40, 3), (1, 16, 16, 16, 44, 3), (2, 17, 16, 32, 40, 1),
    # 7x7: ConvNext stem and closing ConvBlocks (sp_postups.py:121,205-210)
    (2, 13, 19, 2, 8, 7), (1, 16, 16, 8, 8, 7), (1, 20, 17, 8, 1, 7), (1, 12, 12, 24, 24, 7), (1, 9, 9, 64, 16, 7),
    # stencil path (conv_direct.hip): pad2(Cin) * pad2(Cout) <= 8, ragged tiles, several tiles per block
    (2, 21, 70, 1, 1, 3), (1, 9, 33, 3, 1, 3), (2, 8, 32, 1, 3, 3), (1, 17, 40, 2, 4, 3), (1, 5, 5, 4, 2, 3),
    (3, 40, 100, 7, 1, 3), (1, 11, 65, 2, 2, 3), (1, 16, 31, 1, 7, 3), (1, 3, 2, 5, 1, 3),
    # register-resident-filter MFMA path (conv_narrow.hip): Cin, Cout <= 16, Cin % 4 == 0
    (2, 40, 35, 8, 8, 3), (1, 33, 20, 16, 12, 3), (2, 70, 18, 4, 16, 3), (1, 64, 64, 12, 8, 3), (1, 5, 3, 8, 16, 3),
    (6, 32, 16, 16, 16, 3),
    # ... and its two-pixels-per-MFMA-column variant (Cin <= 8, Cout in {4, 8}): ragged 32x16 tiles
    (2, 21, 70, 8, 8, 3), (1, 9, 33, 4, 8, 3), (1, 17, 40, 8, 4, 3), (1, 5, 3, 4, 4, 3), (3, 33, 31, 8, 8, 3),
    # ... with channel counts that are not multiples of 4 on the input side (scalar staging loads)
    (2, 21, 70, 5, 8, 3), (1, 16, 20, 6, 8, 3), (1, 9, 17, 3, 4, 3), (2, 12, 33, 7, 8, 3),
    # streamed-filter MFMA path (conv_stream.hip): Cin >= 16, Cin % 4 == 0, Cout % 4 == 0, H*W >= 256
    (1, 16, 16, 48, 48, 3), (2, 20, 33, 48, 192, 3), (1, 16, 16, 192, 48, 3), (1, 17, 19, 24, 32, 3),
    (1, 18, 16, 32, 64, 3), (1, 16, 17, 64, 24, 3), (1, 16, 16, 96, 40, 3), (2, 32, 32, 40, 40, 3), (1, 16, 16, 20, 36, 3),
    # producer / consumer kernel for 9..16 input channels (conv_narrow16_ws), incl. channel counts that are not multiples of
    # four (16-byte loads at dword alignment, zero filter rows beyond Cin) and ragged / multi-tile grids
    (2, 20, 33, 13, 8, 3), (1, 17, 16, 15, 16, 3), (1, 16, 18, 9, 12, 3), (3, 40, 35, 16, 16, 3), (2, 64, 64, 12, 4, 3),
    (1, 5, 3, 13, 8, 3),
    # ... and output channel counts that are not multiples of four (the last quad of a pixel stored channel by channel)
    (2, 20, 33, 8, 13, 3), (1, 17, 16, 16, 10, 3), (1, 16, 18, 4, 15, 3), (2, 19, 21, 13, 13, 3),
]


@pytest.mark.parametrize('n,h,w,ci,co,ks', CONV_CASES)
def test_conv2d_forward(ops, n, h, w, ci, co, ks):
    x, wt, b = R(n, h, w, ci), R(ks, ks, ci, co) * 0.2, R(co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    close(ops.conv2d(x, wt, b), ref)


@pytest.mark.parametrize('h,w', [(12, 20), (16, 20)])      # 16x20: streamed-filter kernel, 12x20: LDS-staged kernel
def test_conv2d_fused_epilogues(ops, h, w):
    n, ci, co = 2, 16, 24
    x, wt, b, add = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    close(ops.conv2d(x, wt, b, relu=True), np.maximum(ref, 0))
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None, add=add), ref - b + add)


@pytest.mark.parametrize('n,h,w,ci,co', [(2, 40, 33, 48, 48), (1, 32, 16, 48, 192), (1, 33, 17, 192, 48), (1, 64, 16, 24, 32),
                                        (1, 32, 32, 16, 32), (1, 35, 20, 40, 96), (1, 16, 16, 20, 24)])
def test_conv2d_stream_tall_tiles(ops, monkeypatch, n, h, w, ci, co):
    """conv_stream_kernel<..., MT=8> (16x32-pixel tiles, 16/24-channel chunks) is only picked for large grids;
    DL4DS_STREAM_FORCE_TALL makes it take these small ones: forward with fused epilogues, dgrad with accumulate."""
    monkeypatch.setenv('DL4DS_STREAM_FORCE_TALL', '1')
    x, wt, b, add = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    close(ops.conv2d(x, wt, b), ref)
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    dz = R(n, h, w, co)
    gx, _ = _torch_conv_grads(x, wt, dz)
    close(ops.conv2d_dgrad(dz, wt), gx)
    base_x = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt, accumulate_into=base_x), gx + base_x)


def test_conv2d_stream_tall_tiles_depth_to_space(ops, monkeypatch):
    monkeypatch.setenv('DL4DS_STREAM_FORCE_TALL', '1')
    n, h, w, ci, co, r = 2, 34, 20, 48, 192, 2
    x, wt, b = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co)
    ref = N.depth_to_space(N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), r)
    close(ops.conv2d(x, wt, b, d2s=r), ref)
    dz = R(n, h * r, w * r, co // (r * r))
    gx, _ = _torch_conv_grads(x, wt, dz, d2s=r)
    close(ops.conv2d_dgrad(dz, wt, d2s=r), gx)


@pytest.mark.parametrize('sx', ['1', '3', '32'])
@pytest.mark.parametrize('n,h,w,ci,co', [(2, 40, 33, 48, 48), (3, 64, 48, 48, 192), (1, 33, 17, 192, 48), (2, 70, 16, 24, 48),
                                        (1, 35, 20, 48, 96), (2, 32, 32, 48, 40), (5, 32, 16, 24, 24),
                                        # 16x16-pixel tiles (MT = 4): 32/40/48/24-channel chunks, 2-4 cout tiles, ragged Cout (40, 44)
                                        (2, 33, 40, 32, 32), (1, 48, 32, 40, 40), (3, 32, 32, 48, 64), (2, 20, 50, 64, 48),
                                        (1, 64, 16, 24, 32), (2, 17, 33, 80, 96), (1, 32, 32, 96, 44), (4, 16, 16, 32, 128)])
def test_conv2d_stream_producer_consumer(ops, monkeypatch, sx, n, h, w, ci, co):
    """conv_stream_ws_kernel (one persistent 8-wave workgroup per CU: MFMA waves + staging / epilogue waves) is only
    picked for large grids; DL4DS_STREAM_FORCE_WS=<workgroups per XCD> makes it take these small ones with 8, 24 and 256
    workgroups, i.e. with many, a few and at most one item per workgroup: forward with fused epilogues, dgrad with
    accumulate, ReLU mask."""
    monkeypatch.setenv('DL4DS_STREAM_FORCE_WS', sx)
    x, wt, b, add = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    close(ops.conv2d(x, wt, b), ref)
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None, add=add), ref - b + add)
    dz = R(n, h, w, co)
    gx, _ = _torch_conv_grads(x, wt, dz)
    close(ops.conv2d_dgrad(dz, wt), gx)
    base_x = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt, accumulate_into=base_x), gx + base_x)


@pytest.mark.parametrize('sx', ['1', '32'])
@pytest.mark.parametrize('n,h,w,ci,co', [(2, 33, 20, 48, 48), (1, 40, 40, 40, 48), (3, 16, 16, 32, 48), (2, 20, 36, 24, 32), (2, 17, 19, 48, 8),
                                        (1, 32, 32, 32, 16), (2, 24, 33, 24, 24), (1, 16, 16, 40, 40)])
def test_conv2d_stream_producer_consumer_1x1(ops, monkeypatch, sx, n, h, w, ci, co):
    """1x1 layers (TransitionBlock, the projected skips of the residual blocks: blocks.py:208,299) on conv_stream_ws_kernel<1,..>
    (24- .. 48-channel chunks, one to three cout blocks): forward with fused epilogues, dgrad with accumulate; kernel tag asserted."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_STREAM_FORCE_WS', sx)
    x, wt, b, add = R(n, h, w, ci), R(1, 1, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    assert any(t.startswith('conv_stream_ws<1,') for t in tags), tags
    close(got, ref)
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    dz = R(n, h, w, co)
    gx, _ = _torch_conv_grads(x, wt, dz)
    close(ops.conv2d_dgrad(dz, wt), gx)
    base_x = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt, accumulate_into=base_x), gx + base_x)


@pytest.mark.parametrize('sx', ['1', '32'])
@pytest.mark.parametrize('n,h,w,ci,co', [(2, 33, 20, 32, 64), (1, 40, 40, 64, 128), (3, 16, 16, 24, 48), (2, 20, 36, 48, 32), (2, 33, 20, 32, 8), (2, 20, 36, 16, 12),
                                        (1, 32, 32, 128, 96), (2, 17, 17, 32, 40), (2, 24, 33, 16, 32), (1, 16, 16, 80, 48)])
def test_conv2d_stream_producer_consumer_5x5(ops, monkeypatch, sx, n, h, w, ci, co):
    """... and the 5x5 layers (DeconvolutionBlock's 9x9 stride-2 transposed convolutions run as 5x5 convolutions): 32-channel
    chunks (filter ring of 2 fragment sets, 50 group-steps), 24-channel chunks (ring of 3, 75 group-steps) and 16-channel chunks
    (25 group-steps + one idle one); forward with
    fused epilogues, dgrad with accumulate."""
    monkeypatch.setenv('DL4DS_STREAM_FORCE_WS', sx)
    x, wt, b, add = R(n, h, w, ci), R(5, 5, ci, co) * 0.1, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    close(ops.conv2d(x, wt, b), ref)
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    dz = R(n, h, w, co)
    gx, _ = _torch_conv_grads(x, wt, dz)
    close(ops.conv2d_dgrad(dz, wt), gx)
    base_x = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt, accumulate_into=base_x), gx + base_x)


@pytest.mark.parametrize('ci,co', [(64, 128), (32, 64), (32, 32), (128, 256), (48, 64), (16, 32)])
def test_conv2d_stream_producer_consumer_5x5_depth_to_space(ops, monkeypatch, ci, co):
    """The DeconvolutionBlock layers themselves: 5x5 convolution storing through depth_to_space(2) (forward) and reading its
    input through it (dgrad), with groups of 32 / 64 channels (whole n-blocks) and of 16 / 8 channels (narrower than an
    n-block and than a 32-channel chunk: per-quad offsets through the view, several chunks)."""
    monkeypatch.setenv('DL4DS_STREAM_FORCE_WS', '2')
    n, h, w, r = 2, 20, 33, 2
    x, wt, b = R(n, h, w, ci), R(5, 5, ci, co) * 0.1, R(co)
    ref = N.depth_to_space(N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), r)
    close(ops.conv2d(x, wt, b, d2s=r), ref)
    dz = R(n, h * r, w * r, co // (r * r))
    gx, _ = _torch_conv_grads(x, wt, dz, d2s=r)
    close(ops.conv2d_dgrad(dz, wt, d2s=r), gx)


@pytest.mark.parametrize('ci,co', [(48, 192), (48, 32), (24, 32), (48, 96)])
def test_conv2d_stream_producer_consumer_depth_to_space(ops, monkeypatch, ci, co):
    """... through depth_to_space views on the output (forward) and the input (dgrad), also with groups narrower than an
    n-block / a channel chunk (32 couts = 4 groups of 8: the composed upsampling tail of the headline model)."""
    monkeypatch.setenv('DL4DS_STREAM_FORCE_WS', '2')
    n, h, w, r = 2, 34, 20, 2
    x, wt, b = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co)
    ref = N.depth_to_space(N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), r)
    close(ops.conv2d(x, wt, b, d2s=r), ref)
    dz = R(n, h * r, w * r, co // (r * r))
    gx, _ = _torch_conv_grads(x, wt, dz, d2s=r)
    close(ops.conv2d_dgrad(dz, wt, d2s=r), gx)


WINO_CASES = [
    # one pass of 48 / 32 input channels, one .. four chunks of 48 / 32 couts, ragged grids (odd sizes: half tiles)
    (2, 40, 33, 48, 48), (3, 64, 48, 48, 192), (2, 33, 40, 32, 32), (1, 48, 32, 40, 40), (2, 20, 50, 24, 48), (1, 35, 21, 48, 96),
    (1, 17, 19, 24, 32), (5, 32, 16, 24, 24), (2, 32, 32, 48, 40), (1, 18, 16, 32, 64), (1, 16, 16, 28, 36),
    # several passes over the input channels (192 -> 48 is the dgrad of SubpixelConvolution's conv2x)
    (1, 33, 17, 192, 48), (2, 17, 33, 96, 96), (1, 16, 17, 64, 24), (1, 20, 20, 144, 40),
]


@pytest.mark.parametrize('sx', ['', '1'])
@pytest.mark.parametrize('n,h,w,ci,co', WINO_CASES)
def test_conv2d_winograd(ops, monkeypatch, sx, n, h, w, ci, co):
    """conv_wino_kernel (Winograd F(2x2, 3x3), conv_wino.hip) is only picked for grids that give every workgroup several tile
    groups; DL4DS_WINO_FORCE makes it take these small ones ('1': one workgroup per XCD and cout chunk, i.e. many iterations
    of the persistent loop): forward with fused epilogues, dgrad with accumulate.  Same 2e-4 bar as the direct kernels."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_WINO_FORCE', sx or 'all')
    x, wt, b, add = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    assert any(t.startswith('conv_wino<') for t in tags), tags
    close(got, ref)
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None, add=add), ref - b + add)
    dz = R(n, h, w, co)
    gx, _ = _torch_conv_grads(x, wt, dz)
    got, tags = kernel_tags(lambda: ops.conv2d_dgrad(dz, wt))
    assert co < 24 or any(t.startswith('conv_wino<') for t in tags), tags
    close(got, gx)
    base_x = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt, accumulate_into=base_x), gx + base_x)
    monkeypatch.setenv('DL4DS_NO_WINOGRAD', '1')
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    assert not any(t.startswith('conv_wino<') for t in tags), tags
    close(got, ref)


@pytest.mark.parametrize('ci,co', [(48, 192), (48, 32), (24, 32), (48, 96), (32, 128)])
def test_conv2d_winograd_depth_to_space(ops, monkeypatch, ci, co):
    """... through depth_to_space views on the output (forward) and the input (dgrad), also with groups narrower than a cout
    chunk / a channel pass (32 couts = 4 groups of 8: the composed upsampling tail of the headline model)."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_WINO_FORCE', '1')
    n, h, w, r = 2, 34, 20, 2
    x, wt, b = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co)
    ref = N.depth_to_space(N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), r)
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b, d2s=r))
    assert any(t.startswith('conv_wino<') for t in tags), tags
    close(got, ref)
    dz = R(n, h * r, w * r, co // (r * r))
    gx, _ = _torch_conv_grads(x, wt, dz, d2s=r)
    got, tags = kernel_tags(lambda: ops.conv2d_dgrad(dz, wt, d2s=r))
    assert any(t.startswith('conv_wino<') for t in tags), tags
    close(got, gx)


SPLIT_CASES = [
    # 33 .. 48 input channels (or passes of 48), output channels in chunks of 48; ragged grids (widths that are not multiples of the
    # 32-pixel strip, heights that are not multiples of the 32-row segment), several images / chunks / passes
    (2, 40, 33, 48, 48), (3, 64, 48, 48, 192), (1, 48, 32, 40, 40), (1, 35, 21, 48, 96), (2, 32, 32, 48, 40), (1, 70, 100, 40, 48),
    (1, 33, 17, 192, 48), (2, 17, 33, 96, 96), (1, 20, 20, 144, 40), (1, 5, 3, 48, 48), (4, 130, 64, 48, 48),
]


@pytest.mark.parametrize('sx', ['all', '1'])
@pytest.mark.parametrize('n,h,w,ci,co', SPLIT_CASES)
def test_conv2d_split_bf16(ops, monkeypatch, sx, n, h, w, ci, co):
    """Round 6: conv_split_kernel (conv_split.hip) -- the 40 / 48-channel 3x3 layers with every fp32 product as six bf16 MFMA terms
    (operands split exactly into three bf16 parts, fp32 accumulation): forward with the fused epilogues, dgrad with accumulation, at
    the SAME 2e-4 bar as the fp32-pipe kernels and with a measured error of the fp32 MFMA's own size (asserted below against the
    Winograd / direct result of the same call); DL4DS_SPLIT_FORCE makes it take grids and shapes it would leave to them ('1': one
    workgroup per output-channel chunk, i.e. many strips per workgroup; by default it takes the single-pass <= 48 x <= 48 channel layers
    of large grids); DL4DS_NO_SPLIT is the product's A/B switch."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_SPLIT_FORCE', sx)
    x, wt, b, add = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    assert 'conv_split<3,3>' in tags, tags
    close(got, ref)
    e_split = np.abs(got - ref).max() / np.abs(ref).max()
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None, add=add), ref - b + add)
    mask = R(n, h, w, co)
    base = R(n, h, w, co)
    close(ops.conv2d_epilogue(x, wt, b, add=add, mask=mask, relu=True), np.where(mask > 0, np.maximum(ref + add, 0), 0))
    if ci <= 48:
        close(ops.conv2d_epilogue(x, wt, b, mask=mask, accumulate_into=base), base + np.where(mask > 0, ref, 0))
    dz = R(n, h, w, co)
    gx, _ = _torch_conv_grads(x, wt, dz)
    if co > 32 and (co <= 48 or co % 48 == 0) and ci > 32:
        got, tags = kernel_tags(lambda: ops.conv2d_dgrad(dz, wt))
        assert 'conv_split<3,3>' in tags, tags
        close(got, gx)
        if co <= 48:
            base_x = R(*gx.shape)
            close(ops.conv2d_dgrad(dz, wt, accumulate_into=base_x), gx + base_x)
    monkeypatch.setenv('DL4DS_NO_SPLIT', '1')
    got32, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    assert 'conv_split<3,3>' not in tags, tags
    e_f32 = np.abs(got32 - ref).max() / np.abs(ref).max()
    assert e_split < 2e-6 and e_split < 4 * e_f32 + 2e-7, (e_split, e_f32)        # fp32 arithmetic to rounding, not a narrower type


@pytest.mark.parametrize('rows', ['32', '16', '5'])
@pytest.mark.parametrize('n,h,w,ci,co', [(2, 40, 33, 48, 48), (1, 70, 100, 40, 48), (1, 33, 17, 192, 48)])
def test_conv2d_split_bf16_strip_rows(ops, monkeypatch, rows, n, h, w, ci, co):
    """The strip height is a dispatch decision (32 rows; 16 when 32-row strips would leave CUs without one: cfg2 at per-GPU batch 8),
    not arithmetic: every height gives the SAME bits (each output pixel is one fixed-order sum over its 9 x cin products), ragged last strips
    included.  DL4DS_SPLIT_R is the test hook that pins the height (the small grids of the op tests would all take 16 otherwise)."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_SPLIT_FORCE', 'all')
    x, wt, b = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    base = ops.conv2d(x, wt, b)
    monkeypatch.setenv('DL4DS_SPLIT_R', rows)
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    assert 'conv_split<3,3>' in tags, tags
    close(got, ref)
    assert np.array_equal(got, base)
    dz = R(n, h, w, co)
    gx, _ = _torch_conv_grads(x, wt, dz)
    got, tags = kernel_tags(lambda: ops.conv2d_dgrad(dz, wt))
    assert 'conv_split<3,3>' in tags, tags
    close(got, gx)


@pytest.mark.parametrize('ci,co', [(48, 192), (48, 96), (40, 192)])
def test_conv2d_split_bf16_depth_to_space(ops, monkeypatch, ci, co):
    """... through depth_to_space views on the output (forward: SubpixelConvolution's conv2x, blocks.py:414-454) and on the input
    (its dgrad: four passes over 192 gradient channels)."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_SPLIT_FORCE', '1')
    n, h, w, r = 2, 34, 20, 2
    x, wt, b = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co)
    ref = N.depth_to_space(N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), r)
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b, d2s=r))
    assert 'conv_split<3,3>' in tags, tags
    close(got, ref)
    dz = R(n, h * r, w * r, co // (r * r))
    gx, _ = _torch_conv_grads(x, wt, dz, d2s=r)
    got, tags = kernel_tags(lambda: ops.conv2d_dgrad(dz, wt, d2s=r))
    assert 'conv_split<3,3>' in tags, tags
    close(got, gx)


F44_CASES = [
    # F(4x4, 3x3): cout chunks of 32, passes of 48 / 32 input channels; ragged grids (sizes that are not multiples of the 16 x 16 tile group, of the
    # 4 x 4 tile), several images, several chunks, several passes; couts that are not whole chunks (forced)
    (2, 40, 33, 48, 32), (3, 64, 48, 48, 64), (2, 33, 40, 32, 32), (1, 48, 32, 40, 64), (2, 20, 50, 24, 32), (1, 35, 21, 48, 96),
    (1, 17, 19, 24, 32), (5, 32, 16, 32, 64), (2, 32, 32, 48, 48), (1, 16, 16, 28, 36), (1, 33, 17, 192, 32), (2, 17, 33, 96, 96), (1, 16, 17, 64, 40),
]


@pytest.mark.parametrize('sx', ['all', '1'])
@pytest.mark.parametrize('n,h,w,ci,co', F44_CASES)
def test_conv2d_winograd_f44(ops, monkeypatch, sx, n, h, w, ci, co):
    """conv_wino4_kernel (Winograd F(4x4, 3x3), conv_wino4_kernel.h): forward with fused epilogues, dgrad with accumulate, on grids
    the dispatcher would not pick it for (DL4DS_WINO_FORCE as above).  Same 2e-4 bar as the direct kernels (the transform's own
    error is 5e-6)."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_WINO_FORCE', sx)
    monkeypatch.setenv('DL4DS_WINO_F44', 'force')
    x, wt, b, add = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    assert any(t.startswith('conv_wino4<') for t in tags), tags
    close(got, ref)
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None, add=add), ref - b + add)
    dz = R(n, h, w, co)
    gx, _ = _torch_conv_grads(x, wt, dz)
    got, tags = kernel_tags(lambda: ops.conv2d_dgrad(dz, wt))
    assert any(t.startswith('conv_wino') for t in tags), tags
    close(got, gx)
    base_x = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt, accumulate_into=base_x), gx + base_x)


@pytest.mark.parametrize('ci,co', [(48, 128), (48, 32), (24, 32), (32, 128)])
def test_conv2d_winograd_f44_depth_to_space(ops, monkeypatch, ci, co):
    """... through depth_to_space views on the output (forward) and the input (dgrad): 48 -> 4 x 8 is the composed upsampling tail of
    the headline model."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_WINO_FORCE', '1')
    monkeypatch.setenv('DL4DS_WINO_F44', 'force')
    n, h, w, r = 2, 34, 20, 2
    x, wt, b = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co)
    ref = N.depth_to_space(N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), r)
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b, d2s=r))
    assert any(t.startswith('conv_wino4<') for t in tags), tags
    close(got, ref)
    dz = R(n, h * r, w * r, co // (r * r))
    gx, _ = _torch_conv_grads(x, wt, dz, d2s=r)
    got, tags = kernel_tags(lambda: ops.conv2d_dgrad(dz, wt, d2s=r))
    assert any(t.startswith('conv_wino') for t in tags), tags
    close(got, gx)


@pytest.mark.parametrize('sx', ['all', '1'])
@pytest.mark.parametrize('n,h,w,ci,co', WINO_CASES)
def test_conv2d_winograd_wgrad(ops, monkeypatch, sx, n, h, w, ci, co):
    """conv_wino_wgrad_kernel (conv_wino_wgrad.hip): dW = G^T [sum over tiles (B^T d B) . (A dY A^T)] G, also accumulated into
    an existing gradient; DL4DS_WINO_FORCE as above ('1': one workgroup per XCD and chunk pair)."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_WINO_FORCE', sx)
    x, wt, dz = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(n, h, w, co)
    _, gw = _torch_conv_grads(x, wt, dz)
    got, tags = kernel_tags(lambda: ops.conv2d_wgrad(x, dz, 3))
    assert any(t.startswith('conv_wino_wgrad<') for t in tags), tags
    close(got, gw)
    np.testing.assert_array_equal(got, ops.conv2d_wgrad(x, dz, 3))          # bitwise reproducible
    base_w = R(*gw.shape)
    close(ops.conv2d_wgrad(x, dz, 3, accumulate_into=base_w), gw + base_w)
    monkeypatch.setenv('DL4DS_NO_WINOGRAD_WGRAD', '1')
    got, tags = kernel_tags(lambda: ops.conv2d_wgrad(x, dz, 3))
    assert not any(t.startswith('conv_wino') for t in tags), tags
    close(got, gw)


@pytest.mark.parametrize('ci,co', [(48, 192), (48, 32), (24, 32), (48, 96), (32, 128)])
def test_conv2d_winograd_wgrad_depth_to_space(ops, monkeypatch, ci, co):
    """... with the output gradient read through a depth_to_space view (groups of 48, 8, 24 and 32 channels)."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_WINO_FORCE', '1')
    n, h, w, r = 2, 34, 20, 2
    x, wt = R(n, h, w, ci), R(3, 3, ci, co) * 0.2
    dz = R(n, h * r, w * r, co // (r * r))
    _, gw = _torch_conv_grads(x, wt, dz, d2s=r)
    got, tags = kernel_tags(lambda: ops.conv2d_wgrad(x, dz, 3, d2s=r))
    assert any(t.startswith('conv_wino_wgrad<') for t in tags), tags
    close(got, gw)


@pytest.mark.parametrize('ci,co', [(8, 8), (4, 8), (8, 4), (6, 8)])
def test_conv2d_fused_epilogues_pair_path(ops, ci, co):
    n, h, w = 2, 19, 45
    x, wt, b, add = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    close(ops.conv2d(x, wt, b, relu=True), np.maximum(ref, 0))
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None, add=add), ref - b + add)
    dz = R(n, h, w, co)
    gx, _ = _torch_conv_grads(x, wt, dz)
    base_x = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt, accumulate_into=base_x), gx + base_x)


@pytest.mark.parametrize('form', range(16))
@pytest.mark.parametrize('ci,co,h,w', [(8, 8, 37, 70), (5, 8, 19, 45), (16, 16, 37, 40), (13, 16, 20, 33), (8, 1, 35, 45), (1, 1, 19, 70),
                                       (1, 8, 19, 45), (2, 2, 33, 33)])
def test_conv2d_every_epilogue_form(ops, form, ci, co, h, w):
    """The producer / consumer narrow kernels (8 -> 8: conv_narrow_pair_ws, 9..16 -> 16: conv_narrow16_ws) compile one epilogue
    form per combination of residual, ReLU, ReLU mask and accumulation (16 forms each, operands requested before the K loop), and
    the stencil kernels apply the same four switches at run time (second generation: operands fetched before the arithmetic):
    every combination against y = [old +] where(mask > 0, [relu](conv + b + add), 0) on grids with ragged right / bottom tiles
    (several tiles per workgroup, so the XCD-aware walk and the one-tile-ahead descriptors are exercised too)."""
    n = 3
    has_add, relu, has_mask, acc = bool(form & 1), bool(form & 2), bool(form & 4), bool(form & 8)
    x, wt, b = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co)
    add = R(n, h, w, co) if has_add else None
    mask = R(n, h, w, co) if has_mask else None
    old = R(n, h, w, co) if acc else None
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    if has_add:
        ref = ref + add
    if relu:
        ref = np.maximum(ref, 0)
    if has_mask:
        ref = np.where(mask > 0, ref, 0.0)
    if acc:
        ref = ref + old
    from tests.parity import kernel_tags
    got, tags = kernel_tags(lambda: ops.conv2d_epilogue(x, wt, b, add=add, mask=mask, relu=relu, accumulate_into=old))
    want = {(8, 8): 'conv_narrow_pair_ws<', (5, 8): 'conv_narrow_pair_ws<', (16, 16): 'conv_narrow16_ws<', (13, 16): 'conv_narrow16_ws<'}.get(
        (ci, co), 'conv_direct<')
    assert any(t.startswith(want) for t in tags), tags
    close(got, ref)


@pytest.mark.parametrize('ci,co', [(8, 1), (1, 8), (1, 1), (3, 2)])
def test_conv2d_fused_epilogues_stencil_path(ops, ci, co):
    n, h, w = 2, 19, 45
    x, wt, b, add = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    close(ops.conv2d(x, wt, b, relu=True), np.maximum(ref, 0))
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None, add=add), ref - b + add)
    dz = R(n, h, w, co)
    gx, gw = _torch_conv_grads(x, wt, dz)
    base_x, base_w = R(*gx.shape), R(*gw.shape)
    close(ops.conv2d_dgrad(dz, wt, accumulate_into=base_x), gx + base_x)
    close(ops.conv2d_wgrad(x, dz, 3, accumulate_into=base_w), gw + base_w)
    a = ops.conv2d_wgrad(x, dz, 3)
    np.testing.assert_array_equal(a, ops.conv2d_wgrad(x, dz, 3))   # bitwise reproducible


@pytest.mark.parametrize('ci,co,r,h', [(8, 32, 2, 10), (48, 192, 2, 10), (4, 50, 5, 10), (6, 36, 3, 10), (48, 192, 2, 16),
                                        (16, 100, 5, 20), (32, 72, 3, 16)])
def test_conv2d_fused_depth_to_space(ops, ci, co, r, h):
    x, wt, b = R(2, h, 18, ci), R(3, 3, ci, co) * 0.2, R(co)
    ref = N.depth_to_space(N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), r)
    close(ops.conv2d(x, wt, b, d2s=r), ref)


def _torch_conv_grads(x, w, dz, d2s=0):
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    y = T.conv2d(xt, wt)
    if d2s > 1:
        y = T.depth_to_space(y, d2s)
    (y * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    return xt.grad.numpy(), wt.grad.numpy()


@pytest.mark.parametrize('n,h,w,ci,co,ks', CONV_CASES)
def test_conv2d_dgrad_wgrad(ops, n, h, w, ci, co, ks):
    x, wt, dz = R(n, h, w, ci), R(ks, ks, ci, co) * 0.2, R(n, h, w, co)
    gx, gw = _torch_conv_grads(x, wt, dz)
    close(ops.conv2d_dgrad(dz, wt), gx)
    close(ops.conv2d_wgrad(x, dz, ks), gw)


@pytest.mark.parametrize('n,h,w,ci,co,ks', [(4, 8, 8, 128, 256, 5), (16, 8, 8, 256, 132, 3), (2, 16, 16, 64, 128, 3), (5, 7, 9, 80, 68, 3),
                                            (3, 12, 11, 96, 200, 1), (20, 4, 4, 64, 64, 5)])
def test_conv2d_gemm_small_grids(ops, n, h, w, ci, co, ks):
    """conv_gemm_kernel (conv_gemm.hip): small grids with many channels as a GEMM over the flattened pixels of the batch, K split
    over blockIdx.z, partial slabs combined with the epilogue -- the deep levels of unet_pin (sp_preups.py:262-285) and their
    transposed convolutions (blocks.py:508-533).  Forward with every fused epilogue, dgrad with accumulation and ReLU mask,
    ragged pixel counts (M % 128 != 0), cout counts that are not multiples of the 128-wide tile; kernel tag asserted."""
    from tests.parity import kernel_tags
    x, wt, b, add = R(n, h, w, ci), R(ks, ks, ci, co) * 0.05, R(co), R(n, h, w, co)
    xt = torch.tensor(x, dtype=torch.float64)
    ref = (T.conv2d(xt, torch.tensor(wt, dtype=torch.float64)) + torch.tensor(b, dtype=torch.float64)).numpy()
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    assert f'conv_gemm<{ks}>' in tags, tags
    close(got, ref)
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None, add=add), ref - b + add)
    dz = R(n, h, w, co)
    gx, _ = _torch_conv_grads(x, wt, dz)
    got, tags = kernel_tags(lambda: ops.conv2d_dgrad(dz, wt))
    if co >= 64 and co % 16 == 0:
        assert f'conv_gemm<{ks}>' in tags, tags
    close(got, gx)
    base_x = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt, accumulate_into=base_x), gx + base_x)


def test_conv2d_gemm_depth_to_space_both_sides(ops):
    """... storing through a depth_to_space view (the transposed convolution's forward) and reading the output gradient
    through one (its dgrad): 64 -> 4 x 64 channels on an 8 x 8 grid."""
    from tests.parity import kernel_tags
    n, h, w, ci, co, r = 6, 8, 8, 64, 256, 2
    x, wt, b = R(n, h, w, ci), R(3, 3, ci, co) * 0.05, R(co)
    ref = N.depth_to_space(N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)), r)
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b, d2s=r))
    assert 'conv_gemm<3>' in tags, tags
    close(got, ref)
    dz = R(n, h * r, w * r, co // (r * r))
    gx, _ = _torch_conv_grads(x, wt, dz, d2s=r)
    got, tags = kernel_tags(lambda: ops.conv2d_dgrad(dz, wt, d2s=r))
    assert 'conv_gemm<3>' in tags, tags
    close(got, gx)


@pytest.mark.parametrize('n,h,w,ci,co,ks', [(3, 20, 37, 8, 32, 5), (2, 17, 16, 8, 32, 3), (2, 33, 18, 4, 48, 3), (1, 40, 40, 6, 20, 5),
                                            (2, 9, 70, 5, 64, 3), (4, 16, 16, 8, 24, 5)])
def test_conv2d_wgrad_packed_taps(ops, n, h, w, ci, co, ks):
    """Weight gradient of layers with <= 8 input channels and more than 16 outputs (the ConvLSTM2D kernels of the recurrent nets,
    8 -> 32 gate channels, blocks.py:350-355) on conv_wgrad_rows_ws_kernel<1, WCO, KS, PACK>: rows 8..15 of the MFMA's first
    operand carry the same channels one pixel to the right, so one MFMA accumulates two taps.  Ragged grids, several tiles per
    workgroup, channel counts below 8, accumulation, bitwise repeatability."""
    from tests.parity import kernel_tags
    x, dz = R(n, h, w, ci), R(n, h, w, co)
    _, gw = _torch_conv_grads(x, R(ks, ks, ci, co), dz)
    got, tags = kernel_tags(lambda: ops.conv2d_wgrad(x, dz, ks))
    assert any(t.startswith('conv_wgrad_rows<') for t in tags), tags
    close(got, gw)
    base_w = R(*gw.shape)
    close(ops.conv2d_wgrad(x, dz, ks, accumulate_into=base_w), gw + base_w)
    np.testing.assert_array_equal(got, ops.conv2d_wgrad(x, dz, ks))


@pytest.mark.parametrize('n,h,w,ci,co', [(2, 20, 33, 13, 8), (3, 64, 48, 16, 8), (1, 17, 70, 12, 4), (2, 33, 16, 9, 8), (1, 128, 128, 13, 8),
                                         (2, 7, 5, 16, 8)])
def test_conv2d_pair_kernel_sixteen_input_channels(ops, n, h, w, ci, co):
    """Round 5: 9 .. 16 input channels with <= 8 outputs (13 -> 8: ConvBlock_att's first layer in the recurrent nets,
    spt_postups.py:152-157; 16 -> 8: the U-Net decoder's first 512^2 layer, sp_preups.py:262-285) on
    conv_narrow_pair_ws_kernel<2, EPI, C16 = true> -- two pixels per MFMA column, 16-channel k-slots: 48 MFMAs per 32 pixels instead of
    conv_narrow16_ws's 72 with half of their rows idle.  Forward with bias / ReLU / residual, as a dgrad with and without accumulation;
    ragged grids, channel counts and pixel pitches (13 channels: 16-byte loads at dword alignment)."""
    from tests.parity import kernel_tags
    x, wt, b, add = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    assert any(t.startswith('conv_narrow_pair16_ws<') for t in tags), tags
    close(got, ref)
    close(ops.conv2d(x, wt, b, relu=True), np.maximum(ref, 0))
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None, add=add), ref - b + add)
    # as the dgrad of a co -> ci layer (dz has ci channels, dx co): plain and accumulating
    wt2 = R(3, 3, co, ci) * 0.2
    dz = R(n, h, w, ci)
    gx, _ = _torch_conv_grads(R(n, h, w, co), wt2, dz)
    got, tags = kernel_tags(lambda: ops.conv2d_dgrad(dz, wt2))
    assert any(t.startswith('conv_narrow_pair16_ws<') for t in tags), tags
    close(got, gx)
    base_x = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt2, accumulate_into=base_x), gx + base_x)


@pytest.mark.parametrize('n,h,w,ci,co', [(2, 20, 33, 13, 8), (3, 64, 48, 16, 8), (1, 17, 70, 12, 4), (2, 33, 16, 9, 8), (1, 128, 128, 13, 8),
                                         (2, 7, 5, 16, 8), (2, 40, 36, 10, 4), (64, 64, 64, 13, 8)])
def test_conv2d_wgrad_channel_slices(ops, n, h, w, ci, co):
    """Round 6: the 3x3 weight gradient of 9 .. 16 input channels with <= 8 outputs (13 -> 8: ConvBlock_att's first layer in the
    recurrent nets, spt_postups.py:152-157; 16 -> 8 U-Net decoder layers, sp_preups.py:262-285) as two launches of
    conv_narrow_wgrad_kernel<8> on the channel slices [0, 8) and [8, C) -- plain views of the same pixels at the tensor's pitch
    (13 floats: 16-byte loads at dword alignment) -- whose slab sums scatter into the [9][C][Cout] gradient (the bias gradient
    comes from the first slice: covered by the recurrent-net model tests).  Ragged grids / channel counts, accumulation into dW and db, bitwise repeatability."""
    from tests.parity import kernel_tags
    x, dz = R(n, h, w, ci), R(n, h, w, co)
    _, gw = _torch_conv_grads(x, R(3, 3, ci, co), dz)
    got, tags = kernel_tags(lambda: ops.conv2d_wgrad(x, dz, 3))
    assert tags.get('conv_narrow_wgrad<8>') == 2 and not any(t.startswith('conv_wgrad_rows<') for t in tags), tags
    close(got, gw)
    base_w = R(*gw.shape)
    close(ops.conv2d_wgrad(x, dz, 3, accumulate_into=base_w), gw + base_w)
    np.testing.assert_array_equal(got, ops.conv2d_wgrad(x, dz, 3))


@pytest.mark.parametrize('n,h,w,ci,co', [(2, 20, 33, 8, 13), (3, 64, 48, 8, 16), (1, 17, 16, 4, 15), (2, 33, 70, 5, 9), (1, 128, 128, 8, 13)])
def test_conv2d_narrow16_eight_input_channels(ops, n, h, w, ci, co):
    """Round 5: <= 8 input channels with 9 .. 16 outputs (8 -> 13: the dgrad of ConvBlock_att's first layer in the recurrent nets,
    spt_postups.py:152-157) on conv_narrow16_ws_kernel<NR, EPI, C8 = true> -- k-slots 0, 1 carry the channel quads of tap 2j, k-slots
    2, 3 those of tap 2j + 1: five MFMA groups per output row instead of nine.  Forward with bias / ReLU / residual, dgrad (the
    transposed layer ci <- co runs the 16-channel form) with and without accumulation, ragged grids and channel counts."""
    from tests.parity import kernel_tags
    x, wt, b, add = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    if ci % 4 == 0:                                  # (ragged input channel counts through the op-level API take the fallback kernel)
        assert any(t.startswith('conv_narrow16_ws<') for t in tags), tags
    close(got, ref)
    close(ops.conv2d(x, wt, b, relu=True), np.maximum(ref, 0))
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    # as a dgrad: dz with ci' = co channels -> dx with co' = ci ... and the other way round (8 gradient channels -> 13)
    wt2 = R(3, 3, co, ci) * 0.2
    dz = R(n, h, w, ci)
    gx, _ = _torch_conv_grads(R(n, h, w, co), wt2, dz)
    close(ops.conv2d_dgrad(dz, wt2), gx)
    base_x = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt2, accumulate_into=base_x), gx + base_x)


@pytest.mark.parametrize('sx', ['1', '32'])
@pytest.mark.parametrize('n,h,w,co,ks', [(3, 20, 37, 32, 5), (2, 64, 64, 32, 5), (2, 17, 33, 32, 3), (4, 64, 64, 32, 3), (1, 40, 24, 64, 5),
                                         (2, 16, 16, 48, 5)])
def test_conv2d_stream_eight_input_channels(ops, monkeypatch, sx, n, h, w, co, ks):
    """Round 5: 8 input channels with >= 16 outputs (the ConvLSTM2D cells' input convolutions, 8 -> 32 gate channels at 5x5 and 3x3,
    blocks.py:350-355) on conv_stream_ws_kernel<KS, 2, NT, 4> -- one 8-channel chunk, two k-steps per LDS read -- instead of the
    fallback conv_igemm_kernel.  Forced onto small / ragged grids with few and many workgroups; bias, ReLU and residual epilogues."""
    from tests.parity import kernel_tags
    monkeypatch.setenv('DL4DS_STREAM_FORCE_WS', sx)
    x, wt, b, add = R(n, h, w, 8), R(ks, ks, 8, co) * 0.2, R(co), R(n, h, w, co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    got, tags = kernel_tags(lambda: ops.conv2d(x, wt, b))
    if ks == 5 or co % 32 == 0:
        assert any(t.startswith(f'conv_stream_ws<{ks},2,') for t in tags), tags
    close(got, ref)
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None, add=add), ref - b + add)


@pytest.mark.parametrize('h,w', [(9, 17), (16, 20)])        # 16x20: dgrad through the streamed-filter kernel
def test_conv2d_grads_through_d2s_and_accumulate(ops, h, w):
    n, ci, co, r = 2, 48, 192, 2
    x, wt = R(n, h, w, ci), R(3, 3, ci, co) * 0.2
    dz = R(n, h * r, w * r, co // (r * r))
    gx, gw = _torch_conv_grads(x, wt, dz, d2s=r)
    close(ops.conv2d_dgrad(dz, wt, d2s=r), gx)
    close(ops.conv2d_wgrad(x, dz, 3, d2s=r), gw)
    base_x, base_w = R(*gx.shape), R(*gw.shape)
    close(ops.conv2d_dgrad(dz, wt, d2s=r, accumulate_into=base_x), gx + base_x)
    close(ops.conv2d_wgrad(x, dz, 3, d2s=r, accumulate_into=base_w), gw + base_w)


def test_wgrad_large_reduction(ops):
    # K = N*H*W = 32768 pixels, tiny M x N : split over strips + deterministic slab reduce
    n, h, w, ci, co = 2, 128, 128, 8, 8
    x, dz = R(n, h, w, ci), R(n, h, w, co)
    _, gw = _torch_conv_grads(x, R(3, 3, ci, co), dz)
    a = ops.conv2d_wgrad(x, dz, 3)
    close(a, gw)
    np.testing.assert_array_equal(a, ops.conv2d_wgrad(x, dz, 3))   # bitwise reproducible


@pytest.mark.parametrize('c', [1, 8, 24, 48, 100])
def test_bias_act_backward(ops, c):
    dy, y = R(2, 11, 13, c), R(2, 11, 13, c)
    dz, db = ops.bias_act_bwd(dy, y)
    ref = dy * (y > 0)
    close(dz, ref)
    close(db, ref.astype(np.float64).sum(axis=(0, 1, 2)))
    dz2, db2 = ops.bias_act_bwd(dy, None)
    close(dz2, dy)
    close(db2, dy.astype(np.float64).sum(axis=(0, 1, 2)))


def test_depth_to_space_roundtrip(ops):
    x = R(2, 5, 7, 36)
    for r in (2, 3):
        y = ops.depth_to_space(x, r)
        np.testing.assert_array_equal(y, N.depth_to_space(x, r))
        np.testing.assert_array_equal(ops.space_to_depth(y, r), x)


def test_maxpool(ops):
    for shape in ((2, 8, 12, 5), (1, 7, 9, 3)):
        x = R(*shape)
        y = ops.maxpool2(x)
        np.testing.assert_array_equal(y, N.max_pool2(x))
        dy = R(*y.shape)
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        (T.max_pool2(xt) * torch.tensor(dy, dtype=torch.float64)).sum().backward()
        close(ops.maxpool2_bwd(x, y, dy), xt.grad.numpy())


@pytest.mark.parametrize('h,w,ho,wo', [(6, 7, 24, 28), (8, 8, 16, 16), (9, 5, 27, 20), (16, 16, 4, 4)])
def test_resize_bilinear(ops, h, w, ho, wo):
    x = R(2, h, w, 3)
    close(ops.resize_bilinear(x, ho, wo), N.resize_bilinear(x.astype(np.float64), ho, wo))
    dy = R(2, ho, wo, 3)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    (T.resize_bilinear(xt, ho, wo) * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    close(ops.resize_bilinear_bwd(dy, h, w), xt.grad.numpy())


@pytest.mark.parametrize('n,h,w_,c,f', [(3, 9, 11, 2, 2), (13, 20, 17, 2, 2), (5, 8, 8, 3, 2), (2, 6, 7, 4, 5)])
def test_localconv(ops, n, h, w_, c, f):
    """LocallyConnected2D 1x1 (blocks.py:322-328): forward, and the backward kernel whose batch sum is split over four lanes
    of a wave (2 -> 2 channels compiled in, other counts at run time; batch sizes that do not divide by four)."""
    x, w, b = R(n, h, w_, c), R(h, w_, c, f), R(h, w_, f)
    close(ops.localconv(x, w, b), N.locally_connected_1x1(x.astype(np.float64), w, b))
    dy = R(n, h, w_, f)
    xt, wt, bt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, w, b))
    (T.locally_connected_1x1(xt, wt, bt) * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    dx, dw, db = ops.localconv_bwd(x, w, dy)
    close(dx, xt.grad.numpy())
    close(dw, wt.grad.numpy())
    close(db, bt.grad.numpy())


@pytest.mark.parametrize('shape', [(2, 16, 20, 8), (3, 7, 9, 12), (2, 3, 6, 10, 8), (1, 4, 64, 64, 48)])
def test_channel_attention(ops, shape):
    c = shape[-1]
    cr = c // 4
    x = R(*shape)
    w1, b1, w2, b2 = R(1, 1, c, cr), R(cr), R(1, 1, cr, c), R(c)
    from oracle import models as M

    class P(dict):
        def get(self, ops_, name, shape, init='glorot'):
            return self[name]
    for backend, conv in ((N, np.asarray), (T, lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True))):
        pass
    pn = P({'a/conv1/kernel': w1.astype(np.float64), 'a/conv1/bias': b1.astype(np.float64),
            'a/conv2/kernel': w2.astype(np.float64), 'a/conv2/bias': b2.astype(np.float64)})
    ref = M.channel_attention(N, pn, 'a', x.astype(np.float64), c)
    close(ops.channel_attention(x, w1, b1, w2, b2), ref)
    dy = R(*shape)
    pt = P({k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in pn.items()})
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    (M.channel_attention(T, pt, 'a', xt, c) * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    dx, g1, gb1, g2, gb2 = ops.channel_attention_bwd(x, dy, w1, b1, w2, b2)
    close(dx, xt.grad.numpy())
    close(g1, pt['a/conv1/kernel'].grad.numpy())
    close(gb1, pt['a/conv1/bias'].grad.numpy())
    close(g2, pt['a/conv2/kernel'].grad.numpy())
    close(gb2, pt['a/conv2/bias'].grad.numpy())


@pytest.mark.parametrize('kind', ['mae', 'mse'])
def test_pixel_losses(ops, kind):
    yt, yp = R(2, 33, 35, 1), R(2, 33, 35, 1)
    t = torch.tensor(yp, dtype=torch.float64, requires_grad=True)
    lv = getattr(T, kind)(torch.tensor(yt, dtype=torch.float64), t)
    lv.backward()
    a, g = ops.loss(kind, yt, yp)
    assert a == pytest.approx(float(lv), rel=1e-5)
    close(g, t.grad.numpy())


def test_bce(ops):
    p = rng.random((16, 1)).astype(np.float32) * 0.98 + 0.01
    for label in (0.0, 1.0):
        t = torch.tensor(p, dtype=torch.float64, requires_grad=True)
        lv = T.bce(torch.full_like(t, label), t)
        lv.backward()
        a, g = ops.bce(p, label)
        assert a == pytest.approx(float(lv), rel=1e-5)
        close(g, t.grad.numpy())
    assert ops.bce(np.full((4, 1), 0.5, np.float32), 1.0)[0] == pytest.approx(np.log(2.0), rel=1e-6)


def test_adam_matches_keras_form(ops):
    w, g, m, v = R(1001), R(1001), R(1001) * 0.1, np.abs(R(1001)) * 0.1
    for t in (1, 7):
        rw, rm, rv = N.adam_step(w.astype(np.float64), 0.5 * g.astype(np.float64), m.astype(np.float64),
                                 v.astype(np.float64), t, 1e-3)
        aw, am, av = ops.adam(w, g, m, v, t, 1e-3, grad_scale=0.5)
        close(aw, rw, 1e-6)
        close(am, rm, 1e-6)
        close(av, rv, 1e-6)


@pytest.mark.parametrize('n,h,w,ci,co,s', [(2, 6, 7, 4, 3, 2), (1, 8, 8, 16, 8, 2), (1, 4, 5, 24, 16, 2),
                                           (1, 5, 4, 6, 4, 4), (1, 3, 3, 5, 2, 5), (1, 4, 4, 8, 8, 8)])
def test_conv2d_transpose_fwd_dgrad_wgrad(ops, n, h, w, ci, co, s):
    x, wt = R(n, h, w, ci), R(9, 9, co, ci) * 0.1
    ref = N.conv2d_transpose(x.astype(np.float64), wt.astype(np.float64), s)
    close(ops.conv2d_transpose(x, wt, s), ref)
    close(ops.conv2d_transpose(x, wt, s, relu=True), np.maximum(ref, 0))
    dz = R(*ref.shape)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wtt = torch.tensor(wt, dtype=torch.float64, requires_grad=True)
    (T.conv2d_transpose(xt, wtt, s) * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    close(ops.conv2d_transpose_dgrad(dz, wt, s), xt.grad.numpy())
    close(ops.conv2d_transpose_wgrad(x, dz, 9, s), wtt.grad.numpy())


@pytest.mark.parametrize('kind', ['dssim', 'dssim_mae', 'dssim_mse', 'dssim_mae_mse'])
@pytest.mark.parametrize('case', ['positive', 'negative_min', 'pred_sets_range', 'multichannel'])
def test_dssim_losses(ops, kind, case):
    c = 2 if case == 'multichannel' else 1
    yt = rng.random((2, 29, 37, c)).astype(np.float32)
    yp = (yt + 0.1 * rng.standard_normal(yt.shape)).astype(np.float32)
    if case == 'positive':
        yp = np.abs(yp) + 0.01
    elif case == 'negative_min':
        yp = yp - 0.3                       # min(pred) < 0 -> shift branch + argmin routing
        yt = yt - 0.1
    elif case == 'pred_sets_range':
        yp[1, 5, 7, 0] = 3.0                # max(pred) > max(true) -> drange gradient reaches the prediction
        yp[0, 20, 3, 0] = -1.0
    t = torch.tensor(yp, dtype=torch.float64, requires_grad=True)
    lv = getattr(T, kind)(torch.tensor(yt, dtype=torch.float64), t)
    lv.backward()
    a, g = ops.loss(kind, yt, yp)
    assert a == pytest.approx(float(lv.detach()), rel=2e-4, abs=1e-6)
    close(g, t.grad.numpy(), 1e-3)


@pytest.mark.parametrize('kind', ['msdssim', 'msdssim_mae', 'msdssim_mae_mse'])
@pytest.mark.parametrize('case', ['positive', 'negative_min', 'pred_sets_range', 'multichannel_odd'])
def test_msdssim_losses(ops, kind, case):
    """tf.image.ssim_multiscale-based losses (losses.py:92-149): four scales, odd sizes (SYMMETRIC padding of the
    pooling), min-shift and dynamic-range gradients -- loss and gradient against the torch fp64 oracle."""
    c = 2 if case == 'multichannel_odd' else 1
    h, w = (93, 101) if case == 'multichannel_odd' else (96, 104)
    yt = rng.random((2, h, w, c)).astype(np.float32)
    yp = (yt + 0.1 * rng.standard_normal(yt.shape)).astype(np.float32)
    if case == 'positive':
        yp = np.abs(yp) + 0.01
    elif case == 'negative_min':
        yp = yp - 0.3
        yt = yt - 0.1
    elif case == 'pred_sets_range':
        yp[1, 5, 7, 0] = 3.0
        yp[0, 20, 3, 0] = -1.0
    t = torch.tensor(yp, dtype=torch.float64, requires_grad=True)
    lv = getattr(T, kind)(torch.tensor(yt, dtype=torch.float64), t)
    lv.backward()
    a, g = ops.loss(kind, yt, yp)
    assert a == pytest.approx(float(lv.detach()), rel=2e-4, abs=1e-6)
    close(g, t.grad.numpy(), 1e-3)
    a2, g2 = ops.loss(kind, yt, yp)
    assert a2 == a
    np.testing.assert_array_equal(g, g2)                    # deterministic reductions


def untie(x, pre_fn, thr=2e-4):
    """Move the inputs whose fp64 pre-activation lies within ``thr`` of zero: in fp32 such an element takes either
    ReLU branch, so the comparison would test the tie and not the kernel."""
    for _ in range(8):
        bad = np.abs(pre_fn(x.astype(np.float64))) < thr
        if not bad.any():
            return x
        x = x.copy()
        x[bad] += np.float32(0.03)
    raise AssertionError('could not remove the ReLU ties')


NORM_SHAPES = [(2, 9, 7, 3), (1, 16, 16, 64), (3, 5, 11, 20), (2, 4, 6, 130), (1, 33, 17, 256), (2, 3, 5, 1), (1, 8, 8, 1000)]


@pytest.mark.parametrize('shape', NORM_SHAPES)
@pytest.mark.parametrize('relu', [False, True])
def test_layernorm(ops, shape, relu):
    """LayerNormalization(axis=-1) (+ fused ReLU), blocks.py:69-71,158-159: y, dx, dgamma, dbeta vs torch fp64
    autograd; scalar (C % 4 != 0) and float4 lane layouts, C below / above one wavefront of packs."""
    c = shape[-1]
    x = (rng.standard_normal(shape) * 2 + 0.7).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(c)).astype(np.float32)
    dy = rng.standard_normal(shape).astype(np.float32)
    for eps in (1e-3, 1e-6):
        if relu and c > 1:
            x = untie(x, lambda v: N.layer_norm(v, gamma.astype(np.float64), beta.astype(np.float64), eps))
        tx, tg, tb = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, gamma, beta)]
        ty = T.layer_norm(tx, tg, tb, eps)
        if relu:
            ty = torch.relu(ty)
        ty.backward(torch.tensor(dy, dtype=torch.float64))
        y, dx, dg, db = ops.layernorm(x, gamma, beta, eps=eps, relu=relu, dy=dy)
        close(y, ty.detach().numpy(), 1e-4)
        if c == 1:          # a single channel normalises to exactly beta: dx is identically 0, up to rounding / sqrt(eps)
            assert np.abs(dx).max() < 1e-6 / np.sqrt(eps)
        else:
            close(dx, tx.grad.numpy(), 1e-3)
        close(dg, tg.grad.numpy(), 1e-3)
        close(db, tb.grad.numpy(), 1e-3)


@pytest.mark.parametrize('shape', NORM_SHAPES)
@pytest.mark.parametrize('relu', [False, True])
def test_batchnorm(ops, shape, relu):
    """BatchNormalization(axis=-1), momentum 0.99, eps 1e-3 (blocks.py:66-68): training-mode output, gradients and
    moving-average update (Bessel-corrected variance), then the inference-mode output; a channel offset of 50 sigma
    checks the shifted-sum statistics (without ReLU: at that offset fp32 cannot resolve the sign of the smallest
    pre-activations)."""
    c = shape[-1]
    x = (rng.standard_normal(shape) * 0.5 + (0.3 if relu else 25.0) * rng.standard_normal(c)).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(c)).astype(np.float32)
    if relu:
        def pre(v):
            mu, var = N.channel_moments(v)
            return N.batch_norm(v, gamma.astype(np.float64), beta.astype(np.float64), mu, var, 1e-3)
        x = untie(x, pre)
    mm = rng.standard_normal(c).astype(np.float32)
    mv = (0.5 + rng.random(c)).astype(np.float32)
    dy = rng.standard_normal(shape).astype(np.float32)
    tx, tg, tb = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, gamma, beta)]
    mu, var = T.channel_moments(tx)
    ty = T.batch_norm(tx, tg, tb, mu, var, 1e-3)
    if relu:
        ty = torch.relu(ty)
    ty.backward(torch.tensor(dy, dtype=torch.float64))
    y, mm2, mv2, dx, dg, db = ops.batchnorm(x, gamma, beta, mm, mv, training=True, relu=relu, dy=dy)
    n = x.size // c
    close(y, ty.detach().numpy(), 2e-4)
    close(dx, tx.grad.numpy(), 1e-3)
    close(dg, tg.grad.numpy(), 1e-3)
    close(db, tb.grad.numpy(), 1e-3)
    np.testing.assert_allclose(mm2, 0.99 * mm + 0.01 * mu.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mv2, 0.99 * mv + 0.01 * var.detach().numpy() * n / (n - 1), rtol=1e-4, atol=1e-6)
    yi, mm3, mv3 = ops.batchnorm(x, gamma, beta, mm, mv, training=False, relu=relu)
    ref = (x.astype(np.float64) - mm) / np.sqrt(mv.astype(np.float64) + 1e-3) * gamma + beta
    close(yi, np.maximum(ref, 0) if relu else ref, 1e-4)
    np.testing.assert_array_equal(mm3, mm)
    np.testing.assert_array_equal(mv3, mv)


@pytest.mark.parametrize('shape', [(2, 9, 7, 3), (1, 16, 16, 8), (2, 5, 20, 20), (1, 33, 17, 64), (1, 6, 6, 130), (1, 3, 2, 4),
                                   (2, 40, 35, 48)])
def test_depthwise_conv7(ops, shape):
    """DepthwiseConv2D(7, 'same') of ConvNextBlock (blocks.py:143-144): forward, input gradient (mirrored taps, with
    and without accumulation), kernel and bias gradients vs torch fp64; grids smaller than the kernel, channel counts
    off the float4 path, more channel packs than one wgrad block holds."""
    c = shape[-1]
    x, k, b, dy = R(*shape), R(7, 7, c, 1) * 0.2, R(c), R(*shape)
    tx, tk, tb = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, k, b)]
    ty = T.depthwise_conv2d(tx, tk, tb)
    ty.backward(torch.tensor(dy, dtype=torch.float64))
    close(ops.dwconv(x, k, b), ty.detach().numpy())
    close(ops.dwconv(x, k, None), N.depthwise_conv2d(x.astype(np.float64), k.astype(np.float64)))
    y, gx, gk, gb = ops.dwconv(x, k, b, dy=dy)
    close(gx, tx.grad.numpy())
    close(gk, tk.grad.numpy(), 1e-3)
    close(gb, tb.grad.numpy(), 1e-3)
    base = R(*shape)
    _, gx2, _, _ = ops.dwconv(x, k, b, dy=dy, accumulate_into=base)
    close(gx2, tx.grad.numpy() + base)


@pytest.mark.parametrize('shape,out', [((2, 5, 7, 3), (10, 14)), ((1, 8, 6, 4), (20, 15)), ((2, 9, 9, 1), (27, 27)), ((1, 12, 10, 2), (5, 4))])
def test_resize_nearest_in_graph(shape, out):
    """Resizing(interpolation='nearest') (tf.image.resize half-pixel nearest): forward and input gradient through a one-op
    graph + identity 1x1 convolution, integer and fractional ratios, down-sampling included."""
    import dl4ds_amd.graph as G
    from dl4ds_amd.training import SupervisedEngine
    n, h, w, c = shape
    g = G.GraphBuilder()
    x_in = g.input(h, w, c)
    y = g.conv2d(g.resize(x_in, out[0], out[1], interpolation='nearest'), 'id', c, 1, use_bias=False)
    g.finalize(y, seed=0)
    m = G.Model(g, 'resize', [(h, w, c)])
    eye = np.eye(c, dtype=np.float32).reshape(1, 1, c, c)
    m.set_weights({'id/kernel': eye})
    x = R(*shape)
    ref = N.resize_nearest(x.astype(np.float64), *out)
    np.testing.assert_array_equal(m([x]), ref.astype(np.float32))
    # gradient w.r.t. the identity kernel = sum over pixels of resized(x)_ci * dy_co: exercises the forward; the input
    # gradient is exercised by the model tests (rc_interpolation='nearest')
    yt = R(n, out[0], out[1], c)
    eng = SupervisedEngine(m, loss='mse', learning_rate=1e-3)
    _, grads = eng.loss_and_grads([x], yt)
    d = 2.0 * (ref - yt.astype(np.float64)) / ref.size
    gw = np.einsum('nhwi,nhwo->io', ref, d).reshape(1, 1, c, c)
    close(grads['id/kernel'], gw, 1e-3)


@pytest.mark.parametrize('shape,out', [((2, 5, 7, 3), (10, 14)), ((1, 8, 6, 4), (32, 24)), ((2, 9, 9, 1), (27, 27)), ((1, 12, 10, 2), (5, 4)),
                                       ((1, 6, 5, 2), (15, 8))])
@pytest.mark.parametrize('method', ['bicubic', 'lanczos3', 'lanczos5', 'gaussian', 'mitchellcubic'])
def test_resize_bicubic_in_graph(shape, out, method):
    """Resizing(interpolation='lanczos3' | 'lanczos5' | 'gaussian' | 'mitchellcubic') (tf.image.resize's ScaleAndTranslate
    family: spans clamped into the image, normalised) and Resizing(interpolation='bicubic') (tf.image.resize bicubic: Keys A = -0.5 from a 1024-step table, out-of-image taps
    dropped and renormalised): forward, and the gradient THROUGH the resize (the table-driven transpose kernel) via the
    kernel of a 1x1 convolution in front of it -- integer and fractional ratios, down-sampling included."""
    import dl4ds_amd.graph as G
    from dl4ds_amd.training import SupervisedEngine
    n, h, w, c = shape
    g = G.GraphBuilder()
    x_in = g.input(h, w, c)
    z = g.conv2d(x_in, 'pre', c, 1, use_bias=False)
    y = g.conv2d(g.resize(z, out[0], out[1], interpolation=method), 'post', c, 1, use_bias=False)
    g.finalize(y, seed=0)
    m = G.Model(g, 'resize', [(h, w, c)])
    k1, k2 = (R(1, 1, c, c) * 0.5 + np.eye(c, dtype=np.float32)), (R(1, 1, c, c) * 0.5 + np.eye(c, dtype=np.float32))
    m.set_weights({'pre/kernel': k1, 'post/kernel': k2})
    x = R(*shape)
    xt = torch.tensor(x, dtype=torch.float64)
    t1 = torch.tensor(k1.reshape(c, c), dtype=torch.float64, requires_grad=True)
    t2 = torch.tensor(k2.reshape(c, c), dtype=torch.float64, requires_grad=True)
    ref = (T.resize_bicubic(xt @ t1, *out) if method == 'bicubic' else T.resize_scale_translate(xt @ t1, *out, method)) @ t2
    close(m([x]), ref.detach().numpy(), 1e-5)
    yt = R(n, out[0], out[1], c)
    ((ref - torch.tensor(yt, dtype=torch.float64)) ** 2).mean().backward()
    eng = SupervisedEngine(m, loss='mse', learning_rate=1e-3)
    _, grads = eng.loss_and_grads([x], yt)
    close(grads['post/kernel'].reshape(c, c), t2.grad.numpy(), 1e-4)
    close(grads['pre/kernel'].reshape(c, c), t1.grad.numpy(), 1e-4)


@pytest.mark.parametrize('shape,over_time', [((3, 1, 80, 72, 12), False), ((2, 1, 64, 64, 6), False), ((2, 3, 48, 40, 16), True),
                                             ((2, 1, 9, 7, 5), False)])
def test_global_average_pooling_in_graph(shape, over_time):
    """GlobalAveragePooling2D / 3D (discriminator.py:72-74): the coalesced chunked kernel (feature maps of >= 4096
    pixels; float4 and scalar channel packs) and the per-(n, c) kernel for small maps, forward and gradient."""
    import dl4ds_amd.graph as G
    from dl4ds_amd.training import SupervisedEngine
    n, t, h, w, c = shape
    g = G.GraphBuilder()
    x_in = g.input(h, w, c, nmul=t)
    feat = g.conv2d(x_in, 'id', c, 1, use_bias=False)
    y = g.gap(feat, 'gap', over_time=over_time)
    g.finalize(y, seed=0)
    m = G.Model(g, 'gap', [(t, h, w, c) if t > 1 else (h, w, c)])
    m.set_weights({'id/kernel': np.eye(c, dtype=np.float32).reshape(1, 1, c, c)})
    x = R(*((n, t, h, w, c) if t > 1 else (n, h, w, c)))
    out = m([x])
    ref = x.astype(np.float64).mean(axis=(1, 2, 3) if (t > 1 and over_time) else ((2, 3) if t > 1 else (1, 2)))
    close(out.reshape(ref.shape), ref, 1e-5)
    # d loss / d kernel through the pooling backward: loss = mse(gap(x W), yt)
    yt = R(*out.shape)
    eng = SupervisedEngine(m, loss='mse', learning_rate=1e-3)
    _, grads = eng.loss_and_grads([x], yt)
    d = 2.0 * (ref - yt.reshape(ref.shape).astype(np.float64)) / ref.size         # (n[, t], c)
    gw = (ref.reshape(-1, c).T @ d.reshape(-1, c)).reshape(1, 1, c, c)             # gap(x W) = mean(x) W
    close(grads['id/kernel'], gw, 1e-3)


def _random_conv_case(i):
    r = np.random.default_rng(9000 + i)
    pick = lambda xs: xs[int(r.integers(len(xs)))]
    ks = pick([1, 3, 3, 3, 5, 7])
    n = int(pick([1, 2, 3]))
    h, w = int(r.integers(3, 41)), int(r.integers(3, 41))
    chans = [1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 32, 40, 44, 48, 56, 64, 72, 96, 130]
    ci, co = int(pick(chans)), int(pick(chans))
    if ks >= 5:                      # keep the CPU oracle quick
        ci, co = min(ci, 32), min(co, 48)
    d2s = 2 if (ks == 3 and co % 4 == 0 and r.integers(4) == 0) else 0
    return n, h, w, ci, co, ks, d2s


@pytest.mark.parametrize('i', range(48))
def test_conv2d_random_shapes(ops, i):
    """48 seeded draws of (batch, grid, channels, kernel size, depth_to_space): forward with the fused epilogue (bias,
    residual add where allowed, ReLU), dgrad with accumulation, wgrad -- whatever kernel the dispatcher picks (stencil,
    narrow, pair, stream incl. zero-padded filters, LDS-staged fallback incl. split-K and the two-wave-set form)."""
    n, h, w, ci, co, ks, d2s = _random_conv_case(i)
    x, wt, b = R(n, h, w, ci), R(ks, ks, ci, co) * (0.5 / ks), R(co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    if d2s:
        close(ops.conv2d(x, wt, b, relu=True, d2s=d2s), np.maximum(N.depth_to_space(ref, d2s), 0))
        dz = R(n, h * d2s, w * d2s, co // (d2s * d2s))
    else:
        add = R(n, h, w, co)
        close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
        dz = R(n, h, w, co)
    gx, gw = _torch_conv_grads(x, wt, dz, d2s=d2s)
    base = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt, d2s=d2s, accumulate_into=base), gx + base)
    close(ops.conv2d_wgrad(x, dz, ks, d2s=d2s), gw, 5e-4)


@pytest.mark.parametrize('n,h,w,ci,co', [(3, 40, 36, 26, 13), (2, 33, 17, 13, 26), (2, 64, 64, 24, 2), (1, 19, 23, 2, 24),
                                        (2, 21, 30, 50, 25), (1, 5, 7, 7, 3), (1, 16, 16, 61, 64), (4, 32, 32, 10, 6)])
def test_pointwise_conv_odd_channels(ops, n, h, w, ci, co):
    """1x1 layers whose channel counts are not multiples of four (densenet transitions, TransitionLast 26 -> 13,
    LocalizedConvBlock's 24 -> 2; csrc/conv_point.hip: contiguous staging, MFMA, contiguous store): forward with bias +
    residual add + ReLU, forward without an epilogue, dgrad with accumulation (tile tails of 1..255 pixels included)."""
    x, wt, b = R(n, h, w, ci), R(1, 1, ci, co) * 0.3, R(co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    add = R(n, h, w, co)
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, None), ref - b.astype(np.float64))
    dz = R(n, h, w, co)
    gx, gw = _torch_conv_grads(x, wt, dz, d2s=0)
    base = R(*gx.shape)
    close(ops.conv2d_dgrad(dz, wt, accumulate_into=base), gx + base)
    close(ops.conv2d_wgrad(x, dz, 1), gw, 5e-4)


@pytest.mark.parametrize('n,h,w,ci,co', [(3, 45, 70, 5, 8), (2, 33, 31, 3, 4), (2, 64, 64, 6, 8), (1, 16, 16, 7, 8), (2, 40, 36, 13, 13),
                                        (1, 37, 29, 10, 6)])
def test_narrow_kernels_ragged_input_channels(ops, n, h, w, ci, co):
    """3x3 layers whose input channel count is not a multiple of four through the float4-staged narrow kernels (the
    discriminator's 5 -> 8 first layer in the producer / consumer pair kernel, 13 -> 13 in conv_narrow<16>): dword-aligned
    16-byte loads, the quad that reaches beyond the last pixel of the tensor included."""
    x, wt, b = R(n, h, w, ci), R(3, 3, ci, co) * 0.2, R(co)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    add = R(n, h, w, co)
    close(ops.conv2d(x, wt, b, add=add, relu=True), np.maximum(ref + add, 0))
    close(ops.conv2d(x, wt, b), ref)


@pytest.mark.parametrize('shape', [(5, 40, 36, 1), (3, 24, 50, 2), (4, 9, 8, 1)])
def test_image_metrics(shape):
    """compute_metrics' reductions (metrics.py:166-262) on the device vs the numpy restatement: PSNR / SSIM / MAE / RMSE /
    Pearson per test pair, RMSE / bias / Pearson per grid point; a grid below the 11x11 SSIM window reports NaN there."""
    from dl4ds_amd.metrics import image_metrics, compute_metrics
    y = (R(*shape) * 2 + 1).astype(np.float32)
    p = (y + 0.3 * R(*shape)).astype(np.float32)
    m = image_metrics(y, p)
    ref = N.image_metrics(y, p) if min(shape[1:3]) >= 11 else None
    if ref is None:
        assert np.isnan(m['ssim']).all()
        d = p.astype(np.float64) - y
        np.testing.assert_allclose(m['mae'], np.abs(d).reshape(shape[0], -1).mean(1), rtol=1e-5)
        np.testing.assert_allclose(m['rmse_map'], np.sqrt((d ** 2).mean(0)), rtol=1e-5, atol=1e-7)
        return
    assert m['drange'] == pytest.approx(ref['drange'], rel=1e-6)
    for k in ('mae', 'mse', 'rmse', 'psnr', 'pearson'):
        np.testing.assert_allclose(m[k], ref[k], rtol=2e-5, err_msg=k)
    np.testing.assert_allclose(m['ssim'], ref['ssim'], rtol=2e-4, err_msg='ssim')
    for k in ('rmse_map', 'bias_map', 'pearson_map'):
        np.testing.assert_allclose(m[k], ref[k], rtol=2e-4, atol=2e-6, err_msg=k)
    rmse_map, corr_map, nmb, full = compute_metrics(y, p, verbose=False)
    assert rmse_map.shape == shape[1:] and full['summary']['PSNR'][0] == pytest.approx(ref['psnr'].mean(), rel=1e-5)
    np.testing.assert_allclose(nmb, ref['bias_map'] / (y.mean() * 100), rtol=2e-4, atol=1e-8)


# ------------------------------------------------------------------------------------------------ ConvLSTM2D (a6)
CONVLSTM_CASES = [
    # B, T, H, W, Cin, F, KS, relu -- the cfg4 shapes (T = 8, 64^2: 5x5 1 -> 8 and 3x3 8 -> 8), wider filters, ragged grids
    (2, 8, 64, 64, 1, 8, 5, True), (2, 8, 64, 64, 8, 8, 3, True), (1, 8, 64, 64, 2, 4, 5, False),
    (1, 8, 64, 64, 4, 16, 3, True), (2, 8, 64, 64, 3, 16, 5, False), (1, 8, 64, 64, 8, 4, 3, False),
    (2, 8, 37, 23, 2, 8, 5, True), (1, 8, 19, 50, 5, 4, 3, True), (3, 3, 9, 7, 1, 16, 3, False),
]


@pytest.mark.parametrize('B,Tn,H,W,C,F,KS,relu', CONVLSTM_CASES)
def test_conv_lstm2d(B, Tn, H, W, C, F, KS, relu):
    """ConvLSTM2D(F, k, 'same', return_sequences=True) [+ ReLU] as ONE op (blocks.py:350-355; gates i, f, c, o,
    hard-sigmoid recurrent activation, h0 = c0 = 0): output and -- through an MSE loss -- dX, dK (input kernel), dU
    (recurrent kernel) and db against the fp64 torch oracle, every gradient at its own scale."""
    import ctypes
    from dl4ds_amd import _lib
    from dl4ds_amd.graph import GraphBuilder, Model
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import assert_matches_reference, banded_reference
    r = np.random.default_rng(1000 * KS + 10 * F + C + H)
    gb = GraphBuilder()
    xin = gb.input(H, W, C, nmul=Tn, requires_grad=True)
    out = gb.convlstm(xin, 'lstm', F, KS, Tn, activation='relu' if relu else None)
    gb.finalize(out, seed=1)
    model = Model(gb, 'convlstm_only', [(Tn, H, W, C)])
    w = model.get_weights()
    w['lstm/bias'] = (w['lstm/bias'] + 0.1 * r.standard_normal(4 * F)).astype(np.float32)
    # inputs large enough that the hard sigmoids saturate in places (both clip branches of the backward pass run)
    w['lstm/kernel'] = (w['lstm/kernel'] * 2.0).astype(np.float32)
    model.set_weights(w)
    x = (1.5 * r.standard_normal((B, Tn, H, W, C))).astype(np.float32)
    # targets = the oracle's own output +- (0.5 ... 1.5), the sign + for 80 % of the entries (oracle/reference.py): weight gradients
    # that are sums which do not cancel -- against pure-noise targets every entry of dK / dU is a random walk over the pixels, and
    # the handful of gate pre-activations that sit on a hard-sigmoid clip point (2 M gates: there always are some) then move
    # whole filter columns by percents in ANY single-precision evaluation
    with torch.no_grad():
        t64 = lambda a: torch.tensor(np.asarray(a, np.float64))
        o64 = T.conv_lstm2d(t64(x), t64(w['lstm/kernel']), t64(w['lstm/recurrent_kernel']), t64(w['lstm/bias']))
        o64 = (T.relu(o64) if relu else o64).numpy()
    from tests.parity import targets_clear_of_the_kink
    y = targets_clear_of_the_kink(o64, r)
    def call(dt):
        t = lambda a: torch.tensor(np.asarray(a, dt), requires_grad=True)
        xt, kt, ut, bt = t(x), t(w['lstm/kernel']), t(w['lstm/recurrent_kernel']), t(w['lstm/bias'])
        out = T.conv_lstm2d(xt, kt, ut, bt)
        if relu:
            out = T.relu(out)
        loss = ((out - torch.tensor(y.astype(dt))) ** 2).mean()
        gx, gk, gu, gb_ = torch.autograd.grad(loss, [xt, kt, ut, bt])
        return float(loss), {'x': gx, 'lstm/kernel': gk, 'lstm/recurrent_kernel': gu, 'lstm/bias': gb_}, out.detach()
    # reference = mid-point of the evaluations with the hard-sigmoid / ReLU kinks displaced by +/- 4e-6, tests/parity.py
    ref = banded_reference(call)
    got = model([x])
    close(got, ref['pred'])
    eng = SupervisedEngine(model, loss='mse', learning_rate=1e-3)
    l_hip, g_hip = eng.loss_and_grads([x], y)
    assert l_hip == pytest.approx(ref['loss'], rel=1e-4)
    p = ctypes.c_void_p()
    _lib.check(_lib.lib().dl4ds_graph_tensor_ptr(gb.h, xin.id, 1, ctypes.byref(p)))
    dx = np.empty(x.shape, np.float32)
    _lib.check(_lib.lib().dl4ds_memcpy_d2h(dx.ctypes.data, p, dx.nbytes))
    assert_matches_reference(dict(g_hip, x=dx), ref, what=(B, Tn, H, W, C, F, KS, relu))


def test_device_side_error_word_fails_the_next_host_wait_once():
    """ADVICE r3 (convlstm_seq.hip): a persistent kernel that gives up a bounded spin must not let the step pass for a valid one.
    It raises a sticky, host-visible error word; every host-side wait checks it after the stream has drained.  Here a
    one-thread kernel raises the same word the ConvLSTM kernel would: dl4ds_sync fails with the explanation, and the wait
    after that is clean again."""
    import dl4ds_amd._lib as L
    lib = L.lib()
    L.check(lib.dl4ds_sync())
    L.check(lib.dl4ds_debug_raise_device_error(1))
    with pytest.raises(L.Dl4dsHipError, match='persistent ConvLSTM kernel gave up'):
        L.check(lib.dl4ds_sync())
    L.check(lib.dl4ds_sync())
