"""CPU model of the arithmetic of the relevance convolutions' cross products (csrc/conv_f16x3.h, x6_split / x6_pack and the weight
packer of csrc/lrpx_core.hip): fp16 hi halves + block-scaled fp6 e2m3 fields of the values and of the residuals.  Checks the
properties DESIGN.md section 5.1e states - no GPU needed; the GPU side is pinned by tests/test_gpu_vgg.py
(test_packed_f16f6_weight_layout decodes what the packer wrote, test_conv_rule_f16f8 bounds the kernels' error)."""
import numpy as np


def e2m3_values():
    """the 32 non-negative e2m3 values: subnormals 0 .. 0.875 in steps of 1/8, then [1, 2) 1/8, [2, 4) 1/4, [4, 8) 1/2 (max 7.5)"""
    return np.array([m * 0.125 if e == 0 else (1 + m * 0.125) * 2.0 ** (e - 1) for e in range(4) for m in range(8)])


def quant_e2m3(v):
    """round to nearest even onto the e2m3 grid, saturating at +-7.5 (v_cvt_scalef32_pk32_fp6_f16 after the division)"""
    a = np.minimum(np.abs(v), 7.5)
    q = np.where(a < 2, np.rint(a * 8) / 8, np.where(a < 4, np.rint(a * 4) / 4, np.rint(a * 2) / 2))
    return np.sign(v) * np.minimum(q, 7.5)


def block_scale(xmax):
    """2^(e-127) with e the exponent field of xmax * 16/15 / 4 in fp32 (x6_split)"""
    bs = np.float32(xmax) * np.float32(16.0 / 15.0 * 0.25)
    e = (np.asarray(bs, dtype=np.float32).view(np.uint32) >> 23) & 0xFF
    return 2.0 ** (e.astype(np.int64) - 127)


def split_slice(x):
    """x: (..., 16) float32 values already scaled into the fp16 range -> hi (fp16 as float64), fp6 fields of x and of (x - hi) * 2^11,
    block scale per slice"""
    x = x.astype(np.float32)
    hi = x.astype(np.float16).astype(np.float32)
    r = ((x - hi) * np.float32(2048)).astype(np.float16).astype(np.float64)
    s = block_scale(np.abs(x).max(axis=-1, keepdims=True))
    return hi.astype(np.float64), quant_e2m3(x.astype(np.float64) / s), quant_e2m3(r / s), s


def test_e2m3_grid_and_rounding():
    g = e2m3_values()
    assert len(np.unique(g)) == 32 and g.max() == 7.5 and g[1] == 0.125
    assert np.all(np.isin(quant_e2m3(np.linspace(-9, 9, 7201)), np.concatenate([g, -g])))
    # ties go to the even code: 0.0625 -> 0, 0.1875 -> 0.25, 4.25 -> 4, 4.75 -> 5 (the probes of profiles/r03_mfma_f6.txt)
    assert list(quant_e2m3(np.array([0.0625, 0.1875, 4.25, 4.75, 7.9, -30.0]))) == [0.0, 0.25, 4.0, 5.0, 7.5, -7.5]


def test_block_scale_puts_the_maximum_below_saturation():
    rs = np.random.RandomState(0)
    m = np.exp(rs.uniform(np.log(1e-20), np.log(3e4), 200000)).astype(np.float32)
    top = m.astype(np.float64) / block_scale(m)
    assert top.min() >= 3.75 - 1e-6 and top.max() <= 7.5 + 1e-6
    assert block_scale(np.float32(0.0)) == 2.0 ** -127          # an all-zero slice: exponent field 0, every field 0


def test_residual_fits_the_block_of_its_value():
    rs = np.random.RandomState(1)
    x = (rs.randn(100000).astype(np.float32) * np.exp(3 * rs.randn(100000)).astype(np.float32)) * 3e3
    x = x[np.abs(x) < 3.2e4]
    hi = x.astype(np.float16).astype(np.float32)
    big = np.abs(x) >= 2.0 ** -14                                # fp16 normals: 11 significant bits
    assert np.all(np.abs((x - hi) * 2048)[big] <= np.abs(x)[big])


def test_cross_products_of_heavy_tailed_slices():
    """sum_c x_c w_c = hi.hi_w + [x . w_res + r . w] up to the 2^-22 term r . w_res.  Each fp6 factor is off by at most 2^-4 of
    itself (worst case: the bottom of a binade; 1/16 of a block unit for subnormals, i.e. <= 2^-5.9 of its block's maximum), so the
    bracket - two products of two rounded factors, scaled by 2^-11 - is off by at most
        2^-13 sum |x_c w_c|  +  2^-15.9 (max|x| sum |w_c| + max|w| sum |x_c|)
    rigorously; the rounding errors are independent, the observed error is ~30x below that.  Heavy-tailed operands, outliers of
    1e6 inside slices included (tiny entries beside an outlier are the second term)."""
    rs = np.random.RandomState(2)
    n, K = 4000, 16
    x = rs.randn(n, K) * np.exp(4 * rs.randn(n, K))
    x[::7, 3] *= 1e6                                             # an outlier in every 7th slice
    x = (x / np.abs(x).max() * 3.0e4).astype(np.float32)         # per-map scale: the maximum just below 2^15
    w = rs.randn(n, K) * np.exp(1.5 * rs.randn(n, K))
    w = (w / np.abs(w).max() * 3.0e4).astype(np.float32)
    hx, qx, qr, sx = split_slice(x)
    hw, qw, qwr, sw = split_slice(w)
    exact = (x.astype(np.float64) * w.astype(np.float64)).sum(-1)
    cross = ((qx * qwr).sum(-1) + (qr * qw).sum(-1)) * sx[:, 0] * sw[:, 0] * 2.0 ** -11
    got = (hx * hw).sum(-1) + cross
    xa, wa = np.abs(x.astype(np.float64)), np.abs(w.astype(np.float64))
    sxw = (xa * wa).sum(-1)
    # + the absolute floor of fp16 itself: entries below 2^-14 (2^-29 of the map's maximum 2^15) have denormal hi halves, their
    # residuals can exceed the slice's block and saturate: <= 2^-24 per entry, times the other factor
    bound = (2.0 ** -13 * sxw + 2.0 ** -15.9 * (xa.max(-1) * wa.sum(-1) + wa.max(-1) * xa.sum(-1)) + 2.0 ** -20 * sxw +
             2.0 ** -24 * (wa.sum(-1) + xa.sum(-1)))
    err = np.abs(got - exact)
    assert np.all(err <= bound), float((err / bound).max())
    # typical size, on benign slices (entries within e^+-2 of each other, all in the normal fp16 range): the cross products bring the
    # fp16-grade result (1e-4 of sum |x w|) down by a factor of > 10 (median 7e-6), 99 % of the 16-term slices to below 2^-14 (a conv
    # output sums 36 - 288 such slices with independent errors)
    xb = (rs.randn(n, K) * np.exp(rs.randn(n, K))).astype(np.float32) * np.float32(300.0)
    wb = (rs.randn(n, K) * np.exp(rs.randn(n, K))).astype(np.float32) * np.float32(300.0)
    hxb, qxb, qrb, sxb = split_slice(xb)
    hwb, qwb, qwrb, swb = split_slice(wb)
    exb = (xb.astype(np.float64) * wb.astype(np.float64)).sum(-1)
    sb = (np.abs(xb.astype(np.float64)) * np.abs(wb.astype(np.float64))).sum(-1)
    gotb = (hxb * hwb).sum(-1) + ((qxb * qwrb).sum(-1) + (qrb * qwb).sum(-1)) * sxb[:, 0] * swb[:, 0] * 2.0 ** -11
    eb, nb = np.abs(gotb - exb) / sb, np.abs((hxb * hwb).sum(-1) - exb) / sb
    assert np.median(eb) < 0.1 * np.median(nb) and np.quantile(eb, 0.99) < 2.0 ** -14
    print("fp6 cross products: worst error / rigorous bound %.3f (hostile slices); benign slices: error with / without them median "
          "%.1e / %.1e, 99 %% %.1e of sum |x w|" % ((err / bound).max(), np.median(eb), np.median(nb), np.quantile(eb, 0.99)))
