"""GPU: kernel-level parity of libmarl_hip.so (called through the C ABI) against plain
torch fp32 / the golden known-answer vectors."""
import ctypes as C

import numpy as np
import pytest
import torch as th
import torch.nn.functional as F

from tests.util import GOLDEN

pytestmark = pytest.mark.gpu


def _lib():
    from marlclassification_amd import _lib

    return _lib.load(), _lib.check


def _p4(x):
    return (x + 3) & ~3


@pytest.fixture(params=[1, 0], ids=["bf16x6", "f32mfma"])
def mfma_split(request):
    """Both forms of the matrix kernels: 1 = fp32 operands split into three bf16 terms, six bf16
    MFMA products (gemm_split.hip, the default); 0 = v_mfma_f32_32x32x2_f32 (gemm.hip)."""
    lib, check = _lib()
    check(lib.marl_tune(b"mfma_split", request.param))
    yield request.param
    check(lib.marl_tune(b"mfma_split", 1))


def _padded(t, ld):
    out = th.zeros(t.shape[0], ld, device=t.device)
    out[:, : t.shape[1]] = t
    return out


@pytest.mark.parametrize("m,n,k", [(96, 256, 160), (665, 92, 183), (37, 45, 24), (1, 4, 25),
                                   (4096, 512, 368), (3000, 130, 7), (300, 2048, 624),
                                   (32768 + 77, 368, 200)])  # last: many row blocks, ragged M and N
@pytest.mark.parametrize("acc", [0, 1])
@pytest.mark.parametrize("weights", [0, 1], ids=["plainB", "weightB"])
def test_gemm_nt(device, mfma_split, m, n, k, acc, weights):
    lib, check = _lib()
    g = th.Generator().manual_seed(m * 7 + n * 3 + k)
    a = th.randn(m, k, generator=g)
    b = th.randn(n, k, generator=g)
    bias = th.randn(n, generator=g)
    c0 = th.randn(m, n, generator=g)
    ad, bd = _padded(a.to(device), _p4(k) + 4), _padded(b.to(device), _p4(k))
    ldc = n + 3
    cd = th.zeros(m, ldc, device=device)
    cd[:, :n] = c0.to(device)
    bias_d = bias.to(device)
    if weights:  # B as a weight matrix: pre-split image, the path every product of the episode takes
        img = th.zeros(lib.marl_gemm_weight_image_bytes(n, k) // 4 + 64, device=device)
        # (A as in the episode: a slice of a larger finite workspace - the kernel may read up to
        # 112 bytes past the last row's K columns)
        ws = th.zeros(ad.numel() + 64, device=device)
        ws[: ad.numel()] = ad.flatten()
        check(lib.marl_gemm_nt_weights(ws.data_ptr(), ad.shape[1], bd.data_ptr(), bd.shape[1],
                                       bias_d.data_ptr(), cd.data_ptr(), ldc, m, n, k, acc,
                                       img.data_ptr(), None))
    else:
        check(lib.marl_gemm_nt(ad.data_ptr(), ad.shape[1], bd.data_ptr(), bd.shape[1],
                               bias_d.data_ptr(), cd.data_ptr(), ldc, m, n, k, acc, None))
    th.cuda.synchronize()
    ref = a.double() @ b.double().t() + bias.double() + (c0.double() if acc else 0)
    err = (cd[:, :n].cpu().double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-6 * max(1.0, scale) * max(1, k) ** 0.5, (err, scale)
    assert th.equal(cd[:, n:].cpu(), th.zeros(m, ldc - n)), "wrote outside [M, N]"


def test_gemm_nt_is_transpose_detecting(device, mfma_split):
    # asymmetric operands: A = identity-like rows selects rows of B exactly
    lib, check = _lib()
    m, n, k = 64, 96, 64
    a = th.eye(m, k)
    b = th.arange(n * k, dtype=th.float32).view(n, k) / 7.0
    cd = th.zeros(m, n, device=device)
    ad, bd = a.to(device), b.to(device)
    check(lib.marl_gemm_nt(ad.data_ptr(), k, bd.data_ptr(), k, None, cd.data_ptr(), n, m, n, k, 0,
                           None))
    th.cuda.synchronize()
    assert th.equal(cd.cpu(), b.t().contiguous()[:m])


@pytest.mark.parametrize("rows,ni,nj", [(665, 92, 183), (5000, 16, 27), (20000, 200, 130),
                                        (33, 1, 24), (70000, 8, 9), (4096, 1024, 368),
                                        (9001, 300, 257), (65536, 384, 256)])
@pytest.mark.parametrize("waves", [8, 4])
def test_gemm_tn(device, mfma_split, rows, ni, nj, waves):
    """waves: 8 = the 512-thread form of the bf16x6 kernel (one operand per thread while staging,
    32 x 64 per wave; the default), 4 = the 256-thread form (knob tn_split_waves)."""
    lib, check = _lib()
    if waves == 4 and not mfma_split:
        pytest.skip("the fp32-MFMA kernel has one form")
    check(lib.marl_tune(b"tn_split_waves", waves))
    g = th.Generator().manual_seed(rows + ni + nj)
    a = th.randn(rows, ni, generator=g)
    b = th.randn(rows, nj, generator=g)
    ad, bd = _padded(a.to(device), _p4(ni)), _padded(b.to(device), _p4(nj) + 8)
    ldc = _p4(nj)
    cd = th.zeros(ni, ldc, device=device)
    sb = lib.marl_gemm_tn_scratch(ni, nj, rows)
    scratch = th.zeros(sb // 4 + 16, device=device)
    check(lib.marl_gemm_tn(ad.data_ptr(), ad.shape[1], bd.data_ptr(), bd.shape[1], cd.data_ptr(),
                           ldc, ni, nj, rows, scratch.data_ptr(), sb, None))
    ref = a.double().t() @ b.double()
    err = (cd[:, :nj].cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * max(1.0, ref.abs().max().item()) * rows ** 0.5, err
    # deterministic: same bits on a second run
    cd2 = th.zeros_like(cd)
    check(lib.marl_gemm_tn(ad.data_ptr(), ad.shape[1], bd.data_ptr(), bd.shape[1], cd2.data_ptr(),
                           ldc, ni, nj, rows, scratch.data_ptr(), sb, None))
    assert th.equal(cd, cd2)
    check(lib.marl_tune(b"tn_split_waves", 8))


@pytest.mark.parametrize("m,n", [(95, 24), (4096, 384), (7, 1), (300, 1000)])
def test_ln_silu_fwd(device, m, n):
    lib, check = _lib()
    g = th.Generator().manual_seed(m + n)
    z = th.randn(m, n, generator=g) * 3
    gamma = th.randn(n, generator=g)
    beta = th.randn(n, generator=g)
    zd = _padded(z.to(device), _p4(n))
    out = th.zeros(m, _p4(n) + 4, device=device)
    stats = th.zeros(m, 2, device=device)
    gd, bd = gamma.to(device), beta.to(device)
    check(lib.marl_ln_silu_fwd(zd.data_ptr(), zd.shape[1], gd.data_ptr(), bd.data_ptr(),
                               out.data_ptr(), out.shape[1], stats.data_ptr(), m, n, None))
    th.cuda.synchronize()
    ref = F.silu(F.layer_norm(z, (n,), gamma, beta, 1e-5))
    assert th.allclose(out[:, :n].cpu(), ref, rtol=1e-5, atol=2e-6)
    assert th.equal(out[:, n:].cpu(), th.zeros(m, out.shape[1] - n))


def test_patch_gather_and_transition_kats(device):
    from marlclassification_amd.engine import HipEngine
    from tests.util import CASES, model_spec

    z = np.load(GOLDEN + "/g5_unit_kats.npz")
    eng = HipEngine(model_spec(CASES["g1_conftest"]), device)
    img, pos = th.from_numpy(z["crop_img"]), th.from_numpy(z["crop_pos"])
    obs = eng.patch_gather(img.to(device), pos.to(device), 5)
    assert th.equal(obs.cpu(), th.from_numpy(z["crop_obs"]))  # bit-exact copy, non-square image
    table = z["tr_table"].tolist()
    p0 = th.from_numpy(z["tr_pos"])
    acts = th.arange(len(table)).view(-1, 1)
    new = eng.transition(p0.to(device), acts.to(device), table, [10, 10], 5)
    assert th.equal(new.cpu(), th.from_numpy(z["tr_new"]))
