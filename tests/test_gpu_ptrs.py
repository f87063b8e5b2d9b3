"""The C ABI refuses pointers the GPU cannot address with an error code instead of a page fault (SURVEY 8(b)
"Errors": `int` status, distinct codes).  Round 5's intermittent abort was a CPU tensor's `data_ptr()` reaching
`lrpx_vgg16_forward`; the Python shim has refused host tensors since (`_lib.ptr`), but INTEGRATION.md 2(c) tells a maintainer to bind
the library directly - so these tests go around the shim: raw ctypes, raw addresses."""
import ctypes as C
import os
import subprocess
import sys

import pytest
import torch

import lrp_amd  # noqa: F401

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EINVAL = 1


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import _lib, ops, weights
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    raw.lrpx_last_error_string.restype = C.c_char_p
    sd = weights.make_gridtd_state(seed=0, vocab_size=64)
    names = [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")]
    vgg = ops.Vgg16([torch.from_numpy(sd[k]).cuda() for k in names], [torch.from_numpy(sd[k.replace(".weight", ".bias")]).cuda() for k in names])
    img = torch.from_numpy(weights.make_images(0, 1))
    vgg.forward(img.cuda())            # allocates the trace
    torch.cuda.synchronize()
    return lib, raw, vgg, img


def test_vgg16_forward_with_a_host_image_pointer_returns_einval(env):
    lib, raw, vgg, img = env
    host = img.contiguous()            # pageable host memory: what a CPU tensor's data_ptr() is
    rc = raw.lrpx_vgg16_forward(C.c_void_p(vgg.packed.data_ptr()), C.c_void_p(host.data_ptr()), C.c_int(1),
                                C.c_void_p(vgg.trace.data_ptr()), C.c_void_p(0), C.c_void_p(0))
    msg = raw.lrpx_last_error_string().decode()
    assert rc == EINVAL, (rc, msg)
    assert "img_nchw" in msg and "lrpx_vgg16_forward" in msg, msg
    # the context is alive and the same call with the device image succeeds
    dev = img.cuda()
    rc = raw.lrpx_vgg16_forward(C.c_void_p(vgg.packed.data_ptr()), C.c_void_p(dev.data_ptr()), C.c_int(1),
                                C.c_void_p(vgg.trace.data_ptr()), C.c_void_p(0), C.c_void_p(0))
    torch.cuda.synchronize()
    assert rc == 0, raw.lrpx_last_error_string().decode()


def test_relevance_and_helpers_name_the_bad_argument(env):
    lib, raw, vgg, img = env
    r_host = torch.rand(1, 196, 512)
    ws = torch.empty(int(lib.lrpx_vgg16_workspace_bytes(1)) // 4, device="cuda")
    out = torch.empty(1, 3, 224, 224, device="cuda")
    raw.lrpx_vgg16_relevance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = raw.lrpx_vgg16_relevance(vgg.packed.data_ptr(), vgg.trace.data_ptr(), 1, r_host.data_ptr(), None, 1, ws.data_ptr(), out.data_ptr(), None)
    msg = raw.lrpx_last_error_string().decode()
    assert rc == EINVAL and "r_feat_nhwc" in msg, (rc, msg)
    raw.lrpx_cumsum_maps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p]
    host_maps = torch.zeros(2, 16)
    rc = raw.lrpx_cumsum_maps(host_maps.data_ptr(), out.data_ptr(), 1, 2, 16, None)
    assert rc == EINVAL and "`in`" in raw.lrpx_last_error_string().decode()
    # pinned host memory is device-accessible: not refused
    pinned = torch.zeros(2, 16).pin_memory()
    dst = torch.empty(2, 16, device="cuda")
    rc = raw.lrpx_cumsum_maps(pinned.data_ptr(), dst.data_ptr(), 1, 2, 16, None)
    torch.cuda.synchronize()
    assert rc == 0, raw.lrpx_last_error_string().decode()


def test_python_shim_refuses_host_tensors_before_the_abi(env):
    from lrp_amd import _lib
    with pytest.raises(TypeError):
        _lib.ptr(torch.zeros(4))
    with pytest.raises(TypeError):
        _lib.ptr_at(torch.zeros(4, dtype=torch.float64, device="cuda"), 1)


CHILD = r'''
import ctypes as C, sys, torch
sys.path.insert(0, %r)
import lrp_amd
from lrp_amd import _lib
_lib.load()
raw = C.CDLL(_lib.LIB_PATH)
raw.lrpx_last_error_string.restype = C.c_char_p
raw.lrpx_relu.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
y = torch.empty(64, device="cuda")
x = torch.ones(64)
rc = raw.lrpx_relu(x.data_ptr(), y.data_ptr(), 64, None)
print("RC", rc, raw.lrpx_last_error_string().decode())
''' % ROOT


def test_microsecond_entry_points_validate_under_LRPX_CHECK_PTRS(env):
    """the decoder's small steps skip the query by default (a microsecond each on a 5-microsecond launch); LRPX_CHECK_PTRS=1 turns
    it on for every entry point - one child process, the switch latches on first use"""
    e = dict(os.environ, LRPX_CHECK_PTRS="1")
    p = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RC")][0]
    assert line.split()[1] == "1" and "lrpx_relu" in line and "`x`" in line, line
