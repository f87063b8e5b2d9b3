"""GPU parity of the few-row dense kernel (csrc/dense_small.hip) that carries the decoder's per-step epsilon-rule
contractions  x * (W^T (r / z~))  (models/gridTDmodel.py:744-765 `lrp_linear_eps`), through the C ABI
(lrpx_conv_mfma, taps = 1), against plain fp32/fp64 PyTorch on the CPU."""
import pytest
import torch

import lrp_amd  # noqa: F401
from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import ops as o
    return o


@pytest.mark.parametrize("rows,k,n,n_src", [(320, 512, 1536, 16), (37, 96, 64, 5), (2048, 1024, 160, 64)])
def test_dense_rel_few_rows(ops, rows, k, n, n_src):
    """out[row][j] = X[src(row)][j] * sum_i A[row][i] W[i][j]   (PACK_DENSE_T: W is (k rows, n cols))"""
    from lrp_amd import _lib
    g = torch.Generator().manual_seed(rows + k)
    a = torch.randn(rows, k, generator=g)
    w = torch.randn(k, n, generator=g) * 0.05
    x = torch.randn(n_src, n, generator=g)
    src = torch.randint(0, n_src, (rows,), generator=g).to(torch.int32)
    n_pad = -(-n // 32) * 32
    wp = ops.pack_weights(w.cuda(), k, n, 1, _lib.PACK_DENSE_T, 32)
    out = torch.full((rows, n), float("nan"), device="cuda")
    ops.conv_mfma(a.cuda(), wp, rows, 0, k, n_pad, 1, _lib.EPI_REL, pix_per_map=1, oc_split=n, x=x.cuda(),
                  map2img=src.cuda(), out0=out)
    torch.cuda.synchronize()
    want = x[src.long()].double() * (a.double() @ w.double())
    assert rel_err(out.cpu(), want) < 2e-6
    # ... with the addend U (one row per map here): x * (A W + U) - K >= 512 runs the four-wave K-split kernel (dense_ks_kernel<REL>, round 6)
    u = torch.randn(rows, n, generator=g) * 3.0
    out2 = torch.full((rows, n), float("nan"), device="cuda")
    ops.conv_mfma(a.cuda(), wp, rows, 0, k, n_pad, 1, _lib.EPI_REL, pix_per_map=1, oc_split=n, x=x.cuda(), u=u.cuda(),
                  map2img=src.cuda(), out0=out2)
    torch.cuda.synchronize()
    want2 = x[src.long()].double() * (a.double() @ w.double() + u.double())
    assert rel_err(out2.cpu(), want2) < 2e-6


@pytest.mark.parametrize("rows,k,n,relu", [(320, 512, 1536, 0), (16, 64, 96, 1), (320, 2048, 1536, 0), (100, 1056, 64, 0)])
def test_dense_plain_few_rows(ops, rows, k, n, relu):
    """out = A W^T + b (PACK_DENSE: W is (n rows, k cols)), optional ReLU"""
    from lrp_amd import _lib
    g = torch.Generator().manual_seed(rows * 3 + k)
    a = torch.randn(rows, k, generator=g)
    w = torch.randn(n, k, generator=g) * 0.05
    b = torch.randn(n, generator=g)
    n_pad = -(-n // 32) * 32
    wp = ops.pack_weights(w.cuda(), n, k, 1, _lib.PACK_DENSE, 32)
    out = torch.full((rows, n), float("nan"), device="cuda")
    ops.conv_mfma(a.cuda(), wp, rows, 0, k, n_pad, 1, _lib.EPI_PLAIN, pix_per_map=1, oc_split=n, relu=relu,
                  bias=b.cuda(), out0=out)
    torch.cuda.synchronize()
    want = a.double() @ w.double().t() + b.double()
    if relu:
        want = want.clamp(min=0)
    assert rel_err(out.cpu(), want) < 2e-6


@pytest.mark.parametrize("n_maps,P,k,n,n_img", [(40, 196, 512, 512, 4), (23, 36, 512, 2048, 5), (3, 36, 64, 96, 2), (320, 196, 512, 512, 16)])
def test_dense_f16x3_rel_many_rows(ops, n_maps, P, k, n, n_img):
    """The (word, pixel) epsilon rules of the decoders on the fp16 matrix cores (csrc/dense_f16x3.hip):
        r = X[img(map), p] * (sum_i A[map, p, i] W[i, :] + U[map]),  out0 = r,  out1 = r / z~(Zdiv[img(map), p])
    (models/gridTDmodel.py:1125-1128, models/aoamodel.py:1135-1148) against fp64: per map <= 2e-6 of max|r| although the maps'
    magnitudes are spread over 1e-12 .. 1e12 and their entries over e^+-6 (per-map power-of-two operand scale); row counts that
    are no multiple of the 128-row tile, maps that straddle tiles, 96 / 2048 columns; nothing is written behind the outputs;
    out1_amax = the exact per-map maximum of out1."""
    from lrp_amd import _lib
    g = torch.Generator().manual_seed(n_maps * 7 + P)
    a = torch.randn(n_maps, P, k, generator=g) * torch.exp(1.5 * torch.randn(n_maps, P, k, generator=g))
    a = a * torch.logspace(-12, 12, n_maps).view(-1, 1, 1)
    a[n_maps // 2] = 0.0                                                       # an all-zero map (amax = 0)
    w = torch.randn(k, n, generator=g) * 0.05
    x = torch.randn(n_img, P, n, generator=g)
    u = torch.randn(n_maps, n, generator=g) * a.abs().amax(dim=(1, 2)).view(-1, 1) * 0.3
    z = torch.randn(n_img, P, n, generator=g)
    z[0, 0, :5] = 0.0                                                          # z == 0 -> 0.01 (epsilon stabiliser)
    m2i = torch.randint(0, n_img, (n_maps,), generator=g).to(torch.int32)
    n_pad = -(-n // 32) * 32
    wp = ops.pack_weights_f16x2(w.cuda(), k, n, _lib.PACK_BWD_PLAIN, taps=1)
    rows = n_maps * P
    guard = 128 * n
    buf0 = torch.full((rows * n + guard,), 777.0, device="cuda")
    buf1 = torch.full((rows * n + guard,), 777.0, device="cuda")
    amax_in = ops.amax_maps(a.cuda(), n_maps)
    amax_out = torch.zeros(n_maps, dtype=torch.int32, device="cuda")
    ops.conv_mfma(a.cuda(), wp, n_maps, 0, k, n_pad, 1, _lib.EPI_REL, pix_per_map=P, oc_split=n, x=x.cuda(), u=u.cuda(),
                  zdiv=z.cuda(), stab=_lib.STAB_EPS, map2img=m2i.cuda(), out0=buf0[:rows * n], out1=buf1[:rows * n], f16x3=1,
                  in_amax=amax_in, out1_amax=amax_out)
    torch.cuda.synchronize()
    assert (buf0[rows * n:] == 777.0).all() and (buf1[rows * n:] == 777.0).all()
    got0, got1 = buf0[:rows * n].view(n_maps, P, n).cpu().double(), buf1[:rows * n].view(n_maps, P, n).cpu().double()
    xs, zs = x[m2i.long()].double(), z[m2i.long()].double()
    want0 = xs * (a.double() @ w.double() + u.double().unsqueeze(1))
    zt = zs + 0.01 * torch.sign(zs)
    zt[zt == 0] = 0.01
    want1 = want0 / zt
    worst = 0.0
    for m in range(n_maps):
        s0 = want0[m].abs().max().item()
        if s0 == 0:
            assert got0[m].abs().max().item() == 0 and got1[m].abs().max().item() == 0
            continue
        e0 = ((got0[m] - want0[m]).abs().max() / s0).item()
        e1 = ((got1[m] - want1[m]).abs().max() / want1[m].abs().max()).item()
        worst = max(worst, e0, e1)
        assert e0 < 2e-6 and e1 < 4e-6, (m, e0, e1)
        assert amax_out[m:m + 1].view(torch.float32).item() == buf1[:rows * n].view(n_maps, P, n)[m].abs().max().item()
    print(f"dense f16x3 ({n_maps} maps x {P} rows, {k} -> {n}): worst map {worst:.2e} of its maximum")


@pytest.mark.parametrize("n_maps,P,k,n,n_img", [(40, 196, 512, 512, 4), (23, 36, 512, 2048, 5), (3, 36, 64, 96, 2), (320, 196, 512, 512, 16),
                                                 (1, 5, 32, 8, 1), (47, 100, 128, 1000, 5)])
def test_dense_bf16x6_rel_many_rows(ops, n_maps, P, k, n, n_img):
    """The same rules in the DEFAULT arithmetic of the path (conv mode 1: nothing narrower than fp32) - operands split exactly into
    three bf16 parts, six products on v_mfma_f32_32x32x16_bf16 (csrc/dense_f16x3.hip, B6; weights from lrpx_pack_weights_bf16x3 with
    taps = 1): no operand scale, no in_amax, fp32 range.  Against fp64, per map <= 2e-6 of max|r| with the maps' magnitudes spread over
    1e-30 .. 1e30 (beyond fp16's range even behind a per-map scale: entries e^+-6 around it); out0 (with and without the addend U) and
    out1 = r / z~ (epsilon stabiliser, z = 0 included); any number of rows (5 .. 62 720; off the 96 / 128-row tiles), 8 / 96 / 1000 /
    2048 columns; nothing written behind the outputs; a map's result is the same bits whatever else is in the batch."""
    from lrp_amd import _lib
    g = torch.Generator().manual_seed(n_maps * 13 + P)
    a = torch.randn(n_maps, P, k, generator=g) * torch.exp(1.5 * torch.randn(n_maps, P, k, generator=g))
    a = a * torch.logspace(-30, 30, n_maps).view(-1, 1, 1) if n_maps > 1 else a
    if n_maps > 2:
        a[n_maps // 2] = 0.0
    w = torch.randn(k, n, generator=g) * 0.05
    x = torch.randn(n_img, P, n, generator=g)
    u = torch.randn(n_maps, n, generator=g) * a.abs().amax(dim=(1, 2)).view(-1, 1) * 0.3
    z = torch.randn(n_img, P, n, generator=g)
    z[0, 0, :5] = 0.0
    m2i = torch.randint(0, n_img, (n_maps,), generator=g).to(torch.int32)
    n_pad = -(-n // 32) * 32
    wp = ops.pack_weights_bf16x3(w.cuda(), k, n, _lib.PACK_BWD_PLAIN, taps=1)
    rows = n_maps * P
    guard = 128 * n
    xs, zs = x[m2i.long()].double(), z[m2i.long()].double()
    zt = zs + 0.01 * torch.sign(zs)
    zt[zt == 0] = 0.01
    ad, xd, ud, zd, md = a.cuda(), x.cuda(), u.cuda(), z.cuda(), m2i.cuda()
    worst = 0.0
    for with_u, with_o1 in ((True, False), (False, False), (True, True), (False, True)):
        buf = torch.full((rows * n + guard,), 777.0, device="cuda")
        kw = dict(out1=buf[:rows * n], zdiv=zd, stab=_lib.STAB_EPS) if with_o1 else dict(out0=buf[:rows * n])
        ops.conv_mfma(ad, wp, n_maps, 0, k, n_pad, 1, _lib.EPI_REL, pix_per_map=P, oc_split=n, x=xd, u=ud if with_u else None,
                      map2img=md, bf16x6=1, **kw)
        torch.cuda.synchronize()
        assert (buf[rows * n:] == 777.0).all()
        got = buf[:rows * n].view(n_maps, P, n).cpu().double()
        want = xs * (a.double() @ w.double() + (u.double().unsqueeze(1) if with_u else 0.0))
        if with_o1:
            want = want / zt
        for m in range(n_maps):
            s0 = want[m].abs().max().item()
            if s0 == 0:
                assert got[m].abs().max().item() == 0
                continue
            e = ((got[m] - want[m]).abs().max() / s0).item()
            worst = max(worst, e)
            assert e < (4e-6 if with_o1 else 2e-6), (with_u, with_o1, m, e)      # (fp32 accumulation over K = 512: the bounds of the f16x3 test above)
        if n_maps >= 3 and not with_o1:          # the first maps alone (another grid, another tile height): the same bits
            sub = max(1, n_maps // 3)
            buf2 = torch.full((sub * P * n,), 777.0, device="cuda")
            ops.conv_mfma(ad[:sub].contiguous(), wp, sub, 0, k, n_pad, 1, _lib.EPI_REL, pix_per_map=P, oc_split=n, x=xd,
                          u=ud[:sub].contiguous() if with_u else None, map2img=md[:sub].contiguous(), bf16x6=1, out0=buf2)
            torch.cuda.synchronize()
            assert torch.equal(buf2, buf[:sub * P * n])
    print(f"dense bf16x6 ({n_maps} maps x {P} rows, {k} -> {n}): worst map {worst:.2e} of its maximum")


def test_dense_bf16x6_refuses_what_it_is_not_built_for(ops):
    """K off the 32-chunk, two outputs at once, a PLAIN epilogue: EINVAL with a message, no launch"""
    from lrp_amd import _lib
    a = torch.randn(2, 40, 48, device="cuda")
    w = torch.randn(48, 64, device="cuda")
    x = torch.randn(1, 40, 64, device="cuda")
    m2i = torch.zeros(2, dtype=torch.int32, device="cuda")
    wp = ops.pack_weights_bf16x3(w, 48, 64, _lib.PACK_BWD_PLAIN, taps=1)
    out = torch.empty(2, 40, 64, device="cuda")
    with pytest.raises(ValueError, match="multiple of 32"):
        ops.conv_mfma(a, wp, 2, 0, 48, 64, 1, _lib.EPI_REL, pix_per_map=40, oc_split=64, x=x, map2img=m2i, out0=out, bf16x6=1)
    a2 = torch.randn(2, 40, 64, device="cuda")
    wp2 = ops.pack_weights_bf16x3(torch.randn(64, 64, device="cuda"), 64, 64, _lib.PACK_BWD_PLAIN, taps=1)
    with pytest.raises(ValueError, match="ONE output"):
        ops.conv_mfma(a2, wp2, 2, 0, 64, 64, 1, _lib.EPI_REL, pix_per_map=40, oc_split=64, x=x, zdiv=x, stab=_lib.STAB_EPS, map2img=m2i,
                      out0=out, out1=torch.empty_like(out), bf16x6=1)
    with pytest.raises(ValueError, match="REL epilogue"):
        ops.conv_mfma(a2, wp2, 2, 0, 64, 64, 1, _lib.EPI_PLAIN, pix_per_map=40, oc_split=64, out0=out, bf16x6=1)


@pytest.mark.parametrize("per", [36 * 512, 196 * 512, 1024, 2048, 3072, 4096, 6144, 8192, 100, 4 * 1037])
def test_amax_maps_exact_for_every_block_shape(ops, per):
    """lrpx_amax_maps: float bits of max|.| per map, exact, for every float4-per-thread instantiation (256 * ITER float4 dividing
    a map, ITER = 9 / 7 / 1 / 2 / 3 / 4 / 6 / 8) and for map sizes no block divides; a map of zeros gives 0; negative maxima count"""
    n_maps = 37
    g = torch.Generator().manual_seed(per)
    a = torch.randn(n_maps, per, generator=g) * torch.logspace(-20, 20, n_maps).view(-1, 1)
    a[5] = 0.0
    a[7, per // 2] = -3e30
    got = ops.amax_maps(a.cuda(), n_maps).view(torch.float32).cpu()
    assert torch.equal(got, a.abs().amax(dim=1))


@pytest.mark.parametrize("n_maps,P,k,n,n_img", [(120, 36, 512, 2048, 7), (41, 196, 512, 512, 4), (131, 33, 64, 288, 3), (47, 100, 128, 1000, 5),
                                                 (401, 36, 128, 2048, 9)])
def test_dense_f16x3_n256_tiles_match_the_128_tiles_bitwise(ops, n_maps, P, k, n, n_img):
    """dense_f16x3_n256_kernel (128 x 256 tiles, waves side by side: taken for >= 4096 rows, >= 256 columns and ONE output) against
    dense_f16x3_kernel (taken when both outputs are asked for): the same products in the same order per accumulator, so out0,
    out1 and out1_amax are BIT-identical, with and without the addend U; row counts off the 128-row tile, 288 / 1000 columns (a
    partial column block: waves without a column tile of their own), an all-zero map; nothing written behind the outputs.  The
    launcher picks 96-row tiles for the first four shapes (one round of workgroups either way) and 128-row tiles for the last one.
    And EPI_PLAIN (scores with bias) through the same tiles against fp64."""
    from lrp_amd import _lib
    g = torch.Generator().manual_seed(n_maps * 11 + P)
    a = torch.randn(n_maps, P, k, generator=g) * torch.exp(1.5 * torch.randn(n_maps, P, k, generator=g))
    a = (a * torch.logspace(-10, 10, n_maps).view(-1, 1, 1)).cuda()
    a[n_maps // 3] = 0.0
    w = torch.randn(k, n, generator=g) * 0.05
    x = torch.randn(n_img, P, n, generator=g).cuda()
    u = (torch.randn(n_maps, n, generator=g) * 0.3).cuda() * a.abs().amax(dim=(1, 2)).view(-1, 1)
    z = torch.randn(n_img, P, n, generator=g)
    z[0, 0, :5] = 0.0
    z = z.cuda()
    m2i = torch.randint(0, n_img, (n_maps,), generator=g).to(torch.int32).cuda()
    n_pad = -(-n // 32) * 32
    rows = n_maps * P
    assert rows >= 4096 and n_pad >= 256
    wp = ops.pack_weights_f16x2(w.cuda(), k, n, _lib.PACK_BWD_PLAIN, taps=1)
    amax_in = ops.amax_maps(a, n_maps)

    def run(uu, want0, want1):
        b0 = torch.full((rows * n + 4096,), 777.0, device="cuda") if want0 else None
        b1 = torch.full((rows * n + 4096,), 777.0, device="cuda") if want1 else None
        am = torch.zeros(n_maps, dtype=torch.int32, device="cuda")
        ops.conv_mfma(a, wp, n_maps, 0, k, n_pad, 1, _lib.EPI_REL, pix_per_map=P, oc_split=n, x=x, u=uu, zdiv=z if want1 else None,
                      stab=_lib.STAB_EPS, map2img=m2i, out0=b0[:rows * n] if want0 else None, out1=b1[:rows * n] if want1 else None,
                      f16x3=1, in_amax=amax_in, out1_amax=am if want1 else None)
        torch.cuda.synchronize()
        for b in (b0, b1):
            assert b is None or (b[rows * n:] == 777.0).all()
        return (b0[:rows * n] if want0 else None), (b1[:rows * n] if want1 else None), am

    for uu in (u, None):
        ref0, ref1, ref_am = run(uu, True, True)            # both outputs: the 128 x 128 kernel
        got0, _, _ = run(uu, True, False)                   # one output: the 128 x 256 kernel
        _, got1, got_am = run(uu, False, True)
        assert torch.equal(got0, ref0) and torch.equal(got1, ref1) and torch.equal(got_am, ref_am), ("addend" if uu is not None else "no addend")
        assert ref0.abs().max().item() > 0
    # PLAIN: out = A W + b, per-map operand scale
    b = (torch.randn(n, generator=g) * 0.1).cuda()
    buf = torch.full((rows * n + 4096,), 333.0, device="cuda")
    a1 = a * torch.logspace(10, -10, n_maps, device="cuda").view(-1, 1, 1)       # maps back to O(1): the bias matters
    ops.conv_mfma(a1, wp, n_maps, 0, k, n_pad, 1, _lib.EPI_PLAIN, pix_per_map=P, oc_split=n, bias=b, out0=buf[:rows * n], f16x3=1,
                  in_amax=ops.amax_maps(a1, n_maps))
    torch.cuda.synchronize()
    assert (buf[rows * n:] == 333.0).all()
    want = a1.double().view(rows, k) @ w.double().cuda() + b.double()
    err = ((buf[:rows * n].view(rows, n).double() - want).abs().amax(dim=1) / want.abs().amax(dim=1)).max().item()
    print(f"dense f16x3 n256 PLAIN ({rows} x {k} -> {n}): worst row {err:.2e} of its maximum")
    assert err < 2e-6


_N256_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
import lrp_amd
from lrp_amd import ops, _lib
g = torch.Generator().manual_seed(77)
n_maps, P, k, n, n_img = 130, 36, 256, 512, 6
a = (torch.randn(n_maps, P, k, generator=g) * torch.logspace(-6, 6, n_maps).view(-1, 1, 1)).cuda()
w = (torch.randn(k, n, generator=g) * 0.05).cuda()
x = torch.randn(n_img, P, n, generator=g).cuda()
u = torch.randn(n_maps, n, generator=g).cuda() * a.abs().amax(dim=(1, 2)).view(-1, 1) * 0.3
m2i = torch.randint(0, n_img, (n_maps,), generator=g).to(torch.int32).cuda()
wp = ops.pack_weights_f16x2(w, k, n, _lib.PACK_BWD_PLAIN, taps=1)
out = torch.empty(n_maps, P, n, device="cuda")
ops.conv_mfma(a, wp, n_maps, 0, k, n, 1, _lib.EPI_REL, pix_per_map=P, oc_split=n, x=x, u=u, map2img=m2i, out0=out, f16x3=1,
              in_amax=ops.amax_maps(a, n_maps))
torch.cuda.synchronize()
torch.save(out.cpu(), sys.argv[1])
"""


def test_dense_n256_switch_off_gives_the_same_bits(tmp_path):
    """LRPX_DENSE_N256=0 (the switch is read once per process: a child each) sends the many-row rule with one output to the 128 x 128
    kernel, the default to the 128 x 256 one: bit-identical results"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, env in (("default", {}), ("off", {"LRPX_DENSE_N256": "0"})):
        f = tmp_path / f"{name}.pt"
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, "-c", _N256_CHILD % root, str(f)], env=e, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(torch.load(f))
    assert outs[0].abs().max().item() > 0 and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("rows,k,n,n_src", [(320, 512, 1536, 16), (1280, 512, 2048, 64), (37, 96, 64, 5), (640, 1024, 160, 640)])
def test_dense_f16x3_rel_few_rows(ops, rows, k, n, n_src):
    """the lock-step gate rules on the fp16 matrix cores (dense_small_f16x3_kernel): out[row] = X[src(row)] * (A[row] W + U[row]),
    row scales found in-kernel - rows 1e-12 .. 1e12 apart, an all-zero row, row counts off the 32-row tile - against fp64"""
    from lrp_amd import _lib
    g = torch.Generator().manual_seed(rows + k)
    a = torch.randn(rows, k, generator=g) * torch.exp(1.5 * torch.randn(rows, k, generator=g)) * torch.logspace(-12, 12, rows).view(-1, 1)
    a[rows // 3] = 0.0
    w = torch.randn(k, n, generator=g) * 0.05
    x = torch.randn(n_src, n, generator=g)
    u = torch.randn(rows, n, generator=g) * a.abs().amax(dim=1, keepdim=True) * 0.2
    src = torch.randint(0, n_src, (rows,), generator=g).to(torch.int32)
    n_pad = -(-n // 32) * 32
    wp = ops.pack_weights_f16x2(w.cuda(), k, n, _lib.PACK_BWD_PLAIN, taps=1)
    buf = torch.full((rows * n + 4096,), 555.0, device="cuda")
    ops.conv_mfma(a.cuda(), wp, rows, 0, k, n_pad, 1, _lib.EPI_REL, pix_per_map=1, oc_split=n, x=x.cuda(), u=u.cuda(),
                  map2img=src.cuda(), out0=buf[:rows * n], f16x3=1)
    torch.cuda.synchronize()
    assert (buf[rows * n:] == 555.0).all()
    got = buf[:rows * n].view(rows, n).cpu().double()
    want = x[src.long()].double() * (a.double() @ w.double() + u.double())
    scale = want.abs().amax(dim=1)
    live = scale > 0
    err = ((got - want).abs().amax(dim=1)[live] / scale[live]).max().item()
    assert got[~live].abs().max().item() == 0 if (~live).any() else True
    print(f"dense f16x3, few rows ({rows} x {k} -> {n}): worst row {err:.2e} of its maximum")
    assert err < 2e-6


@pytest.mark.parametrize("rows,k,n", [(640, 512, 11027), (320, 512, 9586), (130, 64, 96)])
def test_dense_f16x3_plain_scores(ops, rows, k, n):
    """the (T,V) score block of a trace on the fp16 matrix cores (dense_f16x3_kernel<EPI_PLAIN>): out = A W^T + b with a per-row
    operand scale, rows spread over 1e-6 .. 1e6, against fp64: <= 2e-6 of a row's maximum; nothing written behind the output"""
    from lrp_amd import _lib
    g = torch.Generator().manual_seed(rows + n)
    a = torch.randn(rows, k, generator=g) * torch.logspace(-6, 6, rows).view(-1, 1)
    w = torch.randn(n, k, generator=g) * 0.05
    b = torch.randn(n, generator=g) * 0.1
    wp = ops.pack_weights_f16x2(w.cuda(), n, k, _lib.PACK_FWD, taps=1)
    n_pad = -(-n // 32) * 32
    buf = torch.full((rows * n + 4096,), 333.0, device="cuda")
    ac = a.cuda()
    ops.conv_mfma(ac, wp, rows, 0, k, n_pad, 1, _lib.EPI_PLAIN, pix_per_map=1, oc_split=n, bias=b.cuda(), out0=buf[:rows * n],
                  f16x3=1, in_amax=ops.amax_maps(ac, rows))
    torch.cuda.synchronize()
    assert (buf[rows * n:] == 333.0).all()
    got = buf[:rows * n].view(rows, n).cpu().double()
    want = a.double() @ w.double().t() + b.double()
    err = ((got - want).abs().amax(dim=1) / want.abs().amax(dim=1)).max().item()
    print(f"dense f16x3 PLAIN ({rows} x {k} -> {n}): worst row {err:.2e} of its maximum")
    assert err < 2e-6
