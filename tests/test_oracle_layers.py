"""Pin the CPU oracle (oracle/lrp_oracle.py) against layer-level outputs of the reference's own
LRPtools rule classes (tests/golden/layers.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_err
from oracle import lrp_oracle as O

G = np.load(os.path.join(GOLDEN, "layers.npz"))
T = lambda k: torch.from_numpy(G[k])


def test_conv_alpha1beta0_rule():
    # LRPtools/lrp_modules.py:124-150, incl. a region where Z == 0 exactly
    r = O.conv_alpha1beta0(T("conv_x"), T("conv_w"), T("conv_rout"))
    assert rel_err(r, T("conv_rin")) < 1e-5


def test_maxpool_rule_ties_and_zero_windows():
    r = O.maxpool_rule(T("pool_x"), T("pool_rout"))
    ref = T("pool_rin")
    assert rel_err(r, ref) < 1e-6
    # tie -> first element of the window gets everything; all-zero window -> nothing
    assert (r[0, 0, 0:2, 0:2].flatten()[1:] == 0).all() and r[0, 0, 0, 0] != 0
    assert (r[0, 1, 2:4, 2:4] == 0).all()


def test_mini_network_through_hooks():
    # conv-relu-conv-relu-pool-conv-relu via add_lrp/compute_lrp (LRPtools/lrp_wrapper.py:37-87)
    x = T("mini_x")
    a0 = F.relu(F.conv2d(x, T("mini_w0"), T("mini_b0"), padding=1))
    a1 = F.relu(F.conv2d(a0, T("mini_w2"), T("mini_b2"), padding=1))
    p = F.max_pool2d(a1, 2, 2)
    r = O.conv_alpha1beta0(p, T("mini_w5"), T("mini_target"))
    r = O.maxpool_rule(a1, r)
    r = O.conv_alpha1beta0(a0, T("mini_w2"), r)
    r = O.conv_alpha1beta0(x, T("mini_w0"), r)
    assert rel_err(r, T("mini_r")) < 1e-5


def test_eps_rules():
    # models/gridTDmodel.py:744-765 dense / eye, z containing an exact zero
    d = O.eps_dense(T("eps_r"), T("eps_x"), T("eps_z"), T("eps_w"))
    assert rel_err(d, T("eps_dense_out")) < 1e-5
    e = O.eps_identity(T("eps_r"), T("eps_eye_x"), T("eps_z"))
    assert torch.equal(e, T("eps_eye_out"))      # elementwise path is bit-exact
    assert torch.equal(O.safe_divide(T("eps_r"), T("eps_z")), T("safe_div"))


def test_linear_module_rule_matches_formula():
    # LRPtools/lrp_modules.py:9-37 (not exercised by VGG16; pinned for completeness)
    x = T("lin_x").clone()
    x[x == 0] = -1e-6
    z = x @ T("lin_w").t()
    z = z + 0.01 * z.sign()
    z[z == 0] = 0.01
    r = x * ((T("lin_rout") / z) @ T("lin_w"))
    assert rel_err(r, T("lin_rin")) < 1e-5


M4 = np.load(os.path.join(GOLDEN, "m4.npz"))
M = lambda k: torch.from_numpy(M4[k])


def test_m4_linear_rule_with_inplace_nudge():
    # LRPtools/lrp_modules.py:9-37 at 5 x 70 -> 41 with a zero input row, a zero weight row (Z == 0 -> 0.01)
    r, x_after = O.linear_eps_rule(M("lin_x"), M("lin_w"), M("lin_rout"))
    assert rel_err(r, M("lin_rin")) < 1e-5
    assert torch.equal(x_after, M("lin_x_after"))            # quirk (h): zeros became -1e-6 on the saved input
    assert (M("lin_x") == 0).sum() > 70 and (x_after == 0).sum() == 0


def test_m4_batchnorm_rules():
    # :197-246; channel 2 of the 2-d case has b == 0 and zero inputs (0 / (0 + 1e-7) = 0), channel 4 a negative gamma
    args = [M("bn2_" + k) for k in ("gamma", "beta", "mean", "var")]
    r = O.batchnorm_rule(M("bn2_x"), M("bn2_rout"), *args, float(M4["bn2_eps"]))
    assert torch.equal(r, M("bn2_rin"))                      # elementwise: bit-exact
    assert (r[:, 2, 1:3] == 0).all()
    args = [M("bn1_" + k) for k in ("gamma", "beta", "mean", "var")]
    r = O.batchnorm_rule(M("bn1_x"), M("bn1_rout"), *args, 1e-5)
    assert tuple(r.shape) == (6, 4, 6) and torch.equal(r, M("bn1_rin"))      # the reference's broadcasting quirk
    r = O.batchnorm_rule(M("bn1_x3"), M("bn1_rout3"), *args, 1e-5)
    assert tuple(r.shape) == (6, 6, 5) and torch.equal(r, M("bn1_rin3"))


def test_m4_add_flatten_dropout_rules():
    r1, r2 = O.add_rule(M("add_x1"), M("add_x2"), M("add_rout"))
    assert torch.equal(r1, M("add_r1")) and torch.equal(r2, M("add_r2"))
    assert torch.equal(r1[0, 1], 0.5 * M("add_rout")[0, 1])                  # zero sums: half each (:262-272)
    assert torch.equal(M("flat_rin").reshape(3, 16), M("flat_rout"))         # Flatten: a view (:282-291)
    assert torch.equal(M("drop_rin"), M("drop_r"))                           # Dropout: passes relevance_input (:248-254)


def test_avgpool_rule_vs_reference_pool2d():
    # LRPtools/lrp_modules.py:172-195 on nn.AvgPool2d modules (tests/golden/avgpool.npz, make_golden.py:gen_avgpool): 2x2 with an
    # all-zero window, overlapping windows with and without counted padding, ceil_mode, divisor_override, a global pool
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden import AVGPOOL_CASES
    A = np.load(os.path.join(GOLDEN, "avgpool.npz"))
    for name, shape, kw in AVGPOOL_CASES:
        x, r_out, want = (torch.from_numpy(A[f"{name}_{k}"]) for k in ("x", "rout", "rin"))
        assert tuple(x.shape) == shape
        kw = {k: v for k, v in kw.items() if k != "divisor_override"}      # the reference's clone drops it (lrp_modules.py:176-177)
        got = O.avgpool_rule(x, r_out, **kw)
        assert torch.equal(got, want), (name, rel_err(got, want))       # same summation orders as ATen's loops: bit-exact
    x = torch.from_numpy(A["k2_x"])
    assert (x[0, 1, 2:4, 2:4] == 0).all() and (torch.from_numpy(A["k2_rin"])[0, 1, 2:4, 2:4] == 0).all()   # Z == 0: 0 * (R / 1e-7)


def test_toy_residual_net_through_the_rules():
    # tests/golden/toy_resnet.npz: the reference's add_lrp / compute_lrp on Conv-BN-ReLU, a skip through the explicit Add
    # module, MaxPool, Flatten, Linear (make_golden.py:gen_toy).  The oracle's rule restatements walked by hand in the order
    # autograd drives the reference's hooks; the relevance of relu1's output is the SUM of its two consumers' (Add and conv2)
    import sys
    import torch.nn as nn
    sys.path.insert(0, GOLDEN)
    from make_golden import toy_resnet

    class _Add(nn.Module):
        def forward(self, x, y):
            return x + y

    class _Flat(nn.Module):
        def forward(self, x):
            return x.view(x.size(0), -1)
    g = np.load(os.path.join(GOLDEN, "toy_resnet.npz"))
    net = toy_resnet(np.random.RandomState(int(g["seed"])), _Add, _Flat)
    x = torch.from_numpy(g["x"])
    bn = lambda m, r, xin: O.batchnorm_rule(xin, r, m.weight.data, m.bias.data, m.running_mean, m.running_var, m.eps)
    with torch.no_grad():
        c1 = net.conv1(x); b1 = net.bn1(c1); x1 = net.relu1(b1)
        c2 = net.conv2(x1); b2 = net.bn2(c2); y = net.relu2(b2)
        z = x1 + y; p = net.pool(z); f = p.view(p.size(0), -1)
        logits = net.fc(f)
    assert rel_err(logits, g["logits"]) < 1e-6
    total = 0
    for key, want in (("target", "r1"), ("target2", "r2")):
        r, _ = O.linear_eps_rule(f.clone(), net.fc.weight.data, torch.from_numpy(g[key]))
        r = O.maxpool_rule(z, r.view(p.shape))
        r_x1, r_y = O.add_rule(x1, y, r)
        r = bn(net.bn2, r_y, c2)
        r_x1 = r_x1 + O.conv_alpha1beta0(x1, net.conv2.weight.data, r)
        r = bn(net.bn1, r_x1, c1)
        total = total + O.conv_alpha1beta0(x, net.conv1.weight.data, r)
        assert rel_err(total, g[want]) < 1e-5, want          # the second call returns the `.grad` running sum
