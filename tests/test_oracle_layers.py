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
