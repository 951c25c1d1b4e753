"""-m gpu: every hand-written kernel, called through the C ABI, against a torch fp32 reference of the same op on the
same (16-bit-rounded) inputs.  Tolerances: 16-bit outputs carry one rounding of the output type (bf16 2^-8, f16 2^-11
relative) on top of fp32 accumulation-order noise; fp32 outputs only the latter."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import DT, dev, ptr, stream, call, rel_err, keep_mask

# "x3" = the parity mode: fp32 tensors, Linear as three bf16 MFMA products of hi/lo-split operands (csrc/x3.hip), fp32 attention
TOL16 = {"bf16": 1.2e-2, "f16": 2e-3, "x3": 1e-4}


def test_hw_layouts():
    """lane maps of mfma 16x16x32 / 32x32x16 and ds_read_b64_tr_b16 that every kernel is built on"""
    out = torch.zeros(4096, dtype=torch.int32, device=dev())
    call("mmhip_op_probe_layouts", ptr(out), stream())
    o = out.cpu().numpy()
    lanes = np.arange(64)
    # 16x16x32: D[i][j] = (i+1)*(j+1)*32 with col = lane&15, row = 4*(lane>>4) + reg
    for reg in range(4):
        i, j = 4 * (lanes >> 4) + reg, lanes & 15
        assert (o[lanes * 4 + reg] == (i + 1) * (j + 1) * 32).all()
    # 32x32x16: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    for reg in range(16):
        i, j = (reg & 3) + 8 * (reg >> 2) + 4 * (lanes >> 5), lanes & 31
        assert (o[1024 + lanes * 16 + reg] == (i + 1) * (j + 1) * 32).all()
    # transposing read: lane (group g, index i) receives column i of rows 4g..4g+3 of its group's block
    for e in range(4):
        g, i = lanes >> 4, lanes & 15
        assert (o[2048 + lanes * 4 + e] == ((4 * g + e) * 64 + i) % 256).all()
    # k-order of the 16x16x32 operands: D[i][j] = (j+1) + 64*(j+17)
    for reg in range(4):
        j = lanes & 15
        assert (o[2304 + lanes * 4 + reg] == (j + 1) + 64 * (j + 17)).all()


def _gemm_ref(A, B, bias, act, resid, mulg, keep, scale):
    v = A.float() @ B.float().t()
    if bias is not None:
        v = v + bias
    pre = v.clone()
    if act == 1:
        v = torch.nn.functional.gelu(v)
    if act == 2:
        v = torch.tanh(v)
    if mulg is not None:
        u = mulg.float()
        v = v * (0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi))
    if keep is not None:
        v = v * keep.to(v.device).float() * scale
    if resid is not None:
        v = v + resid.float()
    return v, pre


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,slow", [(256, 256, 128, 0), (200, 128, 64, 0), (1000, 768, 768, 0), (8192, 2304, 768, 0),
                                        (512, 768, 3072, 0), (96, 48, 40, 1), (64, 768, 768, 0),
                                        (1000, 768, 768, 96), (8192, 2304, 768, 96), (300, 192, 128, 96),
                                        (300, 128, 128, 144), (8192, 3072, 768, 144), (1000, 768, 2304, 144),
                                        (200, 96, 64, 160), (8192, 768, 768, 160), (1000, 2304, 768, 160), (300, 480, 128, 0),
                                        (300, 96, 64, 192), (8192, 768, 3072, 192), (1000, 2304, 768, 192),
                                        (200, 256, 64, 208), (1000, 768, 768, 208), (8192, 2304, 768, 208), (300, 512, 128, 208),
                                        (200, 128, 64, 224), (1000, 768, 768, 224), (8192, 2304, 768, 224), (12608, 768, 3072, 224), (300, 384, 192, 224),
                                        (8192, 3072, 768, 240), (12608, 2304, 768, 240), (8192, 3072, 64, 240), (70000, 256, 128, 240),
                                        (8192, 3072, 768, 256), (8192, 768, 3072, 256), (12608, 2304, 768, 256), (16384, 768, 2304, 256),
                                        (40000, 128, 64, 256), (35000, 128, 192, 256),
                                        (200, 192, 64, 272), (1000, 768, 768, 272), (8192, 3072, 768, 272), (12608, 768, 3072, 272), (300, 384, 128, 272),
                                        (8192, 3072, 768, 288), (12608, 768, 768, 288), (8192, 3072, 64, 288), (70000, 192, 128, 288), (16384, 2304, 192, 288),
                                        (256, 256, 320, 208), (256, 128, 320, 224), (256, 192, 320, 272), (35000, 256, 192, 240), (35000, 384, 192, 288),
                                        (4096, 768, 3072, 320), (1152, 768, 2048, 320), (200, 128, 64, 320), (300, 256, 128, 320),
                                        (4096, 768, 3072, 336), (1152, 768, 3072, 336), (200, 128, 64, 336), (300, 384, 192, 336), (4096, 768, 2304, 0)])
def test_gemm_nt_epilogues(dt, M, N, K, slow):
    """`slow`: bit 0 forces the generic kernel, bits 4.. pick the tile variant (16 = 128x128, 96 = 128x192, 144 = role-specialised 256x128
    (MFMA waves + LDS-DMA loader waves), 160 = 128x96, 192 = role-specialised 256x96, 208 / 224 = deep-pipelined 256x256 / 256x128
    (gemm8.hip: interleaved K-loop schedule, counted vmcnt across raw barriers, register epilogue, bias from LDS), 240 / 256 = the same,
    persistent, 272 / 288 = deep-pipelined 256x192, one-shot / persistent, 320 / 336 = 128x128 on a 4- / 3-deep LDS ring (round 4: long K on
    grids of <= 256 tiles; (4096, 768, 2304, 0) takes the 3-deep one by the measured rule))"""
    code, tdt = DT[dt]
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(tdt).to(dev())
    B = (torch.randn(N, K, generator=g) * 0.05).to(tdt).to(dev())
    bias = torch.randn(N, generator=g).to(dev())
    resid = torch.randn(M, N, generator=g).to(tdt).to(dev())
    mulg = torch.randn(M, N, generator=g).to(tdt).to(dev())
    for variant in ("plain", "bias_gelu_aux", "bias_drop_resid", "mulgrad_resid", "bias_tanh_f32"):
        C_ = torch.zeros(M, N, dtype=torch.float32 if variant == "bias_tanh_f32" else tdt, device=dev())
        aux = torch.zeros(M, N, dtype=tdt, device=dev())
        kw = dict(bias=None, act=0, aux=None, mulg=None, p=0.0, resid=None, f32=0)
        if variant == "bias_gelu_aux":
            kw.update(bias=bias, act=1, aux=aux)
        if variant == "bias_drop_resid":
            kw.update(bias=bias, p=0.1, resid=resid)
        if variant == "mulgrad_resid":
            kw.update(mulg=mulg, resid=resid)
        if variant == "bias_tanh_f32":
            kw.update(bias=bias, act=2, f32=1)
        seed, sid = 0x123456789ABCDEF, 21
        call("mmhip_op_gemm_nt", code, ptr(A), K, ptr(B), K, ptr(C_), N, M, N, K, ptr(kw["bias"]), kw["act"], ptr(kw["aux"]), N,
             ptr(kw["mulg"]), N, kw["p"], seed, sid, ptr(kw["resid"]), N, kw["f32"], slow, stream())
        torch.cuda.synchronize()
        keep, scale = (None, 1.0)
        if kw["p"] > 0:
            keep, scale = keep_mask((M, N), sid, seed, kw["p"])
        ref, pre = _gemm_ref(A, B, kw["bias"], kw["act"], kw["resid"], kw["mulg"], keep, scale)
        tol = TOL16[dt] if dt == "x3" else (2e-5 if kw["f32"] else TOL16[dt])
        assert rel_err(C_, ref) < tol, (variant, rel_err(C_, ref))
        if kw["aux"] is not None:
            assert rel_err(aux, pre) < TOL16[dt]
        if kw["p"] > 0:      # the very same elements are dropped
            z = (C_.float().cpu() - resid.float().cpu()).abs() < 1e-6
            assert (z == ~keep).float().mean() > 0.99      # kept values below half an output ulp of the residual also read as "dropped"


@pytest.mark.parametrize("dt,slow", [("bf16", 0), ("bf16", 16), ("bf16", 224), ("f16", 256), ("x3", 0)])
def test_gelu_epilogue_is_the_erf_gelu(dt, slow):
    """the activation of the FC1 epilogue alone: with B = I the accumulator IS the (16-bit exact) input, so the fp32 output is mm_gelu(x) --
    checked against x Phi(x) in double over [-9, 9].  mm_gelu is the division-free tail form (csrc/mmhip_common.h; tools/gelu_fit.py):
    |error| <= 4.8e-7 by construction; the bound below adds fp32 rounding of the result."""
    code, tdt = DT[dt]
    M, N = 512, 128
    g = torch.Generator(device="cpu").manual_seed(5)
    x = ((torch.rand(M, N, generator=g) * 18 - 9).to(torch.bfloat16 if dt == "x3" else tdt)).to(tdt)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-4, -1e-4, 5.5, -5.5, 30.0, -30.0]).to(tdt)
    A, B = x.to(dev()), torch.eye(N, dtype=tdt, device=dev())
    C_ = torch.zeros(M, N, dtype=torch.float32, device=dev())
    call("mmhip_op_gemm_nt", code, ptr(A), N, ptr(B), N, ptr(C_), N, M, N, N, None, 1, None, 0, None, 0, 0.0, 0, 0, None, 0, 0 if dt == "x3" else 1, slow, stream())
    torch.cuda.synchronize()
    xd = x.double()
    ref = xd * 0.5 * torch.erfc(-xd / math.sqrt(2.0))
    err = (C_.double().cpu() - ref).abs()
    assert (err <= 6e-7 + 1.2e-7 * ref.abs()).all(), err.max().item()
    # a NaN pre-activation stays NaN (the division-free form clamps with v_med3_f32, which drops a NaN operand: the kernel adds 0 * x back)
    A[1, 3] = float("nan")
    call("mmhip_op_gemm_nt", code, ptr(A), N, ptr(B), N, ptr(C_), N, M, N, N, None, 1, None, 0, None, 0, 0.0, 0, 0, None, 0, 0 if dt == "x3" else 1, slow, stream())
    torch.cuda.synchronize()
    assert torch.isnan(C_[1, 3]).item() and torch.isfinite(C_[0]).all().item()


@pytest.mark.parametrize("dt,slow", [("bf16", 0), ("bf16", 16), ("bf16", 224), ("f16", 256), ("x3", 0)])
def test_gelu_grad_epilogue_is_the_erf_gelu_derivative(dt, slow):
    """the dFC2 epilogue alone: B = I and A = 1 make the fp32 output mm_gelu_grad2(u) -- against Phi(u) + u phi(u) in double over [-9, 9]
    (|error| <= 1.3e-6 by construction, csrc/mmhip_common.h)"""
    code, tdt = DT[dt]
    M, N = 512, 128
    g = torch.Generator(device="cpu").manual_seed(6)
    u = ((torch.rand(M, N, generator=g) * 18 - 9).to(torch.bfloat16 if dt == "x3" else tdt)).to(tdt)
    u[0, :8] = torch.tensor([0.0, -0.0, 1e-4, -1e-4, 5.5, -5.5, 30.0, -30.0]).to(tdt)
    B, U = torch.eye(N, dtype=tdt, device=dev()), u.to(dev())
    A = torch.ones(M, N, dtype=tdt, device=dev())          # A . I^T: the accumulator is exactly 1 everywhere
    C_ = torch.zeros(M, N, dtype=torch.float32, device=dev())
    call("mmhip_op_gemm_nt", code, ptr(A), N, ptr(B), N, ptr(C_), N, M, N, N, None, 0, None, 0, ptr(U), N, 0.0, 0, 0, None, 0, 0 if dt == "x3" else 1, slow, stream())
    torch.cuda.synchronize()
    ud = u.double()
    ref = 0.5 * torch.erfc(-ud / math.sqrt(2.0)) + ud * torch.exp(-ud * ud / 2) / math.sqrt(2 * math.pi)
    err = (C_.double().cpu() - ref).abs()
    assert err.max().item() <= 2e-6, err.max().item()


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,Nn,Nc,slow", [(64, 128, 128, 0), (256, 256, 384, 0), (8192, 768, 768, 0), (1024, 2304, 768, 0),
                                          (96, 40, 72, 1), (64, 128, 128, 16), (256, 256, 384, 16),])
def test_gemm_tn(dt, M, Nn, Nc, slow):
    code, tdt = DT[dt]
    g = torch.Generator(device="cpu").manual_seed(M + Nn)
    A = (torch.randn(M, Nn, generator=g) * 0.1).to(tdt).to(dev())
    B = (torch.randn(M, Nc, generator=g) * 0.5).to(tdt).to(dev())
    C_ = torch.full((Nn, Nc), 3.0, device=dev())
    cs = torch.full((Nn,), 7.0, device=dev())
    call("mmhip_op_gemm_tn", code, ptr(A), Nn, ptr(B), Nc, ptr(C_), Nc, M, Nn, Nc, 0, slow, ptr(cs), stream())
    ref = A.double().t() @ B.double()
    ref_cs = A.double().sum(0)                     # the bias gradient that goes with dW: column sums of the dY operand
    tol = 5e-5 if dt == "x3" else 3e-5            # x3: operands carry 16 of their 24 mantissa bits through the matrix cores
    assert rel_err(C_, ref) < tol and rel_err(cs, ref_cs) < tol
    call("mmhip_op_gemm_tn", code, ptr(A), Nn, ptr(B), Nc, ptr(C_), Nc, M, Nn, Nc, 1, slow, ptr(cs), stream())
    assert rel_err(C_, 2 * ref) < tol and rel_err(cs, 2 * ref_cs) < tol
    call("mmhip_op_gemm_tn", code, ptr(A), Nn, ptr(B), Nc, ptr(C_), Nc, M, Nn, Nc, 0, slow, None, stream())
    assert rel_err(C_, ref) < tol and rel_err(cs, 2 * ref_cs) < tol         # NULL: column sums untouched


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("rows,width", [(37, 768), (8192, 768), (130, 1024)])
def test_layernorm_fwd_bwd(dt, rows, width):
    code, tdt = DT[dt]
    g = torch.Generator(device="cpu").manual_seed(rows)
    x = (torch.randn(rows, width, generator=g) * 2 + 0.3).to(tdt).to(dev())
    gamma = (1 + 0.1 * torch.randn(width, generator=g)).to(dev())
    beta = (0.1 * torch.randn(width, generator=g)).to(dev())
    dy = torch.randn(rows, width, generator=g).to(tdt).to(dev())
    dres = torch.randn(rows, width, generator=g).to(tdt).to(dev())
    y = torch.zeros_like(x)
    mean, rstd = torch.zeros(rows, device=dev()), torch.zeros(rows, device=dev())
    call("mmhip_op_layernorm_fwd", code, ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), rows, width, 1e-5, stream())
    xf = x.float().requires_grad_(True)
    gf, bf = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xf, (width,), gf, bf, 1e-5)
    assert rel_err(y, ref.detach()) < TOL16[dt]
    dx = torch.zeros_like(x)
    dg, db = torch.zeros(width, device=dev()), torch.zeros(width, device=dev())
    call("mmhip_op_layernorm_bwd", code, ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dres), ptr(dg), ptr(db), rows, width, stream())
    ref.backward(dy.float())
    assert rel_err(dx, xf.grad + dres.float()) < TOL16[dt]
    assert rel_err(dg, gf.grad) < 1e-4 and rel_err(db, bf.grad) < 1e-4


def _attn_ref(qkv, maskbias, posts, S, heads, keep, scale, dctx=None):
    H = heads * 64
    x = qkv.float().view(posts, S, 3, heads, 64).requires_grad_(dctx is not None)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(-1, -2) * 0.125
    if maskbias is not None:
        s = s + maskbias.view(posts, 1, 1, S)
    lse = torch.logsumexp(s, dim=-1)
    a = torch.softmax(s, dim=-1)
    if keep is not None:
        a = a * keep.to(a.device).float().view_as(a) * scale
    ctx = (a @ v).permute(0, 2, 1, 3).reshape(posts * S, H)
    if dctx is None:
        return ctx, lse, None
    ctx.backward(dctx.float())
    return ctx.detach(), lse.detach(), x.grad.reshape(posts * S, 3 * H)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("posts,S,heads,masked,p", [(3, 128, 12, True, 0.0), (2, 64, 2, True, 0.1), (2, 197, 12, False, 0.0),
                                                    (5, 32, 1, False, 0.0), (2, 50, 3, True, 0.0), (64, 128, 12, True, 0.1)])
def test_attention_fwd_bwd(dt, posts, S, heads, masked, p):
    code, tdt = DT[dt]
    H = heads * 64
    g = torch.Generator(device="cpu").manual_seed(S + posts)
    qkv = torch.randn(posts * S, 3 * H, generator=g).to(tdt).to(dev())
    maskbias = None
    if masked:
        lens = torch.randint(1, S + 1, (posts,), generator=g)
        lens[0] = S
        maskbias = torch.where(torch.arange(S)[None, :] < lens[:, None], 0.0, float("-inf")).to(dev()).contiguous()
    ctx = torch.zeros(posts * S, H, dtype=tdt, device=dev())
    lse = torch.zeros(posts, heads, S, device=dev())
    seed, sid = 77, 16
    call("mmhip_op_attn_fwd", code, ptr(qkv), ptr(maskbias), ptr(ctx), ptr(lse), posts, S, heads, p, seed, sid, stream())
    keep, scale = (None, 1.0) if p == 0 else keep_mask((posts, heads, S, S), sid, seed, p)
    bwd = S <= 128
    dctx = torch.randn(posts * S, H, generator=g).to(tdt).to(dev()) if bwd else None
    ref_ctx, ref_lse, ref_dqkv = _attn_ref(qkv, maskbias, posts, S, heads, keep, scale, dctx)
    assert rel_err(ctx, ref_ctx) < TOL16[dt], rel_err(ctx, ref_ctx)
    assert (lse - ref_lse).abs().max().item() < 2e-3
    if bwd:
        dqkv = torch.full((posts * S, 3 * H), float("nan"), dtype=tdt, device=dev())
        call("mmhip_op_attn_bwd", code, ptr(qkv), ptr(maskbias), ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), posts, S, heads, p, seed, sid, stream())
        assert torch.isfinite(dqkv.float()).all()
        for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
            e = rel_err(dqkv[:, sl], ref_dqkv[:, sl])
            assert e < 2.5 * TOL16[dt], (name, e)


# ---- parity mode (dtype code MMHIP_F32 at the op level): the same operators on fp32 tensors
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (200, 128, 64), (1000, 768, 768), (96, 48, 40), (8192, 2304, 768), (64, 768, 3072), (300, 132, 96), (130, 512, 768), (12608, 768, 3072), (700, 384, 192),
                                   (64, 3072, 768), (128, 768, 768), (100, 176, 256), (5, 16, 128), (128, 2304, 3072)])          # the last five: few rows, K split over a workgroup's waves (gemm_nt_x3_small_kernel)
def test_x3_gemm_nt(M, N, K):
    test_gemm_nt_epilogues("x3", M, N, K, 0)


@pytest.mark.parametrize("M,Nn,Nc", [(64, 128, 128), (256, 256, 384), (8192, 768, 768), (1024, 2304, 768), (96, 40, 72), (4, 768, 3072), (100, 768, 768)])
def test_x3_gemm_tn(M, Nn, Nc):
    test_gemm_tn("x3", M, Nn, Nc, 0)


@pytest.mark.parametrize("rows,width", [(37, 768), (8192, 768), (130, 1024)])
def test_x3_layernorm(rows, width):
    test_layernorm_fwd_bwd("x3", rows, width)


@pytest.mark.parametrize("posts,S,heads,masked,p", [(3, 128, 12, True, 0.0), (2, 64, 2, True, 0.1), (2, 197, 12, False, 0.0), (5, 32, 1, False, 0.0),
                                                    (2, 50, 3, True, 0.0), (8, 128, 12, True, 0.1)])
def test_x3_attention(posts, S, heads, masked, p):
    test_attention_fwd_bwd("x3", posts, S, heads, masked, p)


def test_colsum_and_casts():
    x = torch.randn(1000, 2304, device=dev()).to(torch.bfloat16)
    out = torch.zeros(2304, device=dev())
    call("mmhip_op_colsum", 0, ptr(x), 1000, 2304, 2304, ptr(out), stream())
    assert rel_err(out, x.float().sum(0)) < 1e-5
    src = torch.randn(2304, 768, device=dev())
    d1 = torch.zeros(2304, 768, dtype=torch.bfloat16, device=dev())
    d2 = torch.zeros(768, 2304, dtype=torch.bfloat16, device=dev())
    call("mmhip_op_cast", 0, ptr(src), ptr(d1), src.numel(), 0, 0, stream())
    call("mmhip_op_cast", 0, ptr(src), ptr(d2), 0, 2304, 768, stream())
    assert torch.equal(d1, src.to(torch.bfloat16)) and torch.equal(d2, src.to(torch.bfloat16).t().contiguous())


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_gemm_tn_group_matches_single_problems(dt):
    """mmhip_op_gemm_tn_group: many C (+)= A^T B problems of different shapes in few launches (fast grouped tiles and generic shapes mixed),
    accumulate into non-zero C and column sums -- against fp64"""
    import ctypes as C
    from smtc_amd import _lib
    code, tdt = DT[dt]
    g = torch.Generator(device="cpu").manual_seed(11)
    shapes = [(4096, 768, 768), (4096, 3072, 768), (1152, 768, 2048), (4096, 768, 3072), (80, 3, 768), (144, 768, 4), (4096, 2304, 768), (1152, 768, 768),
              (256, 256, 128), (4096, 768, 768), (4096, 768, 768)]
    keep, refs = [], []
    arr = (_lib.TNProblem * len(shapes))()
    for i, (M, Nn, Nc) in enumerate(shapes):
        A = (torch.randn(M, Nn, generator=g) * 0.1).to(tdt).to(dev())
        B = (torch.randn(M, Nc, generator=g) * 0.5).to(tdt).to(dev())
        C0 = torch.randn(Nn, Nc, generator=g).to(dev())
        cs = torch.randn(Nn, generator=g).to(dev()) if Nn % 4 == 0 else None
        refs.append((C0.double().cpu() + A.double().cpu().t() @ B.double().cpu(), None if cs is None else cs.double().cpu() + A.double().cpu().sum(0)))
        keep.append((A, B, C0, cs))
        arr[i] = _lib.TNProblem(A.data_ptr(), B.data_ptr(), C0.data_ptr(), M, Nn, Nc, Nn, Nc, Nc, None if cs is None else cs.data_ptr())
    call("mmhip_op_gemm_tn_group", code, C.cast(arr, C.c_void_p), len(shapes), 1, stream())
    torch.cuda.synchronize()
    for (A, B, C0, cs), (rc, rcs), shp in zip(keep, refs, shapes):
        assert rel_err(C0, rc) < 3e-5, shp
        if cs is not None:
            assert rel_err(cs, rcs) < 3e-5, shp
