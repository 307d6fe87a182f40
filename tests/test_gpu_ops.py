"""GPU parity tests of the building-block kernels, called through the C ABI, against plain fp32
torch CPU math of the same op. Run on the MI355X box with `pytest -m gpu`."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from refil_amd._lib import (GEMM_A_OUTC, GEMM_ACCUM, GEMM_B_OUTC, GEMM_COLSUM_A, GEMM_RELU, GEMM_RELU_BWD, MASK_ENTITY,
                            MASK_INTERACT, MASK_OBS, MASK_OBS_INTERACT, MASK_OBS_WITHIN, MASK_WITHIN)

DEV = "cuda"


def _close(a, b, tol=2e-5, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = max(b.abs().max().item(), 1e-6)
    assert err <= tol * ref, f"{what}: max abs err {err:.3e} vs ref max {ref:.3e} (tol {tol})"


def _rowmap(r, grp, gstride, off):
    return (r // grp) * gstride + (r % grp) + off if grp else r


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,pad", [(1000, 128, 84, 4), (300, 64, 128, 0), (777, 22, 64, 0), (130, 384, 128, 0),
                                       (257, 40, 10, 1), (64, 32, 33, 3)])
def test_gemm_linear_forward(M, N, K, pad):
    import hip_ops
    torch.manual_seed(M + N + K)
    lda = K + pad
    x = torch.randn(M, lda)
    W = torch.randn(N, K) / math.sqrt(K)
    b = torch.randn(N)
    mask = (torch.rand(50) < 0.3).to(torch.uint8)
    ref = torch.relu(x[:, :K] @ W.t() + b)
    ref[mask[torch.arange(M) % 50].bool()] = 0
    y = torch.full((M, N), float("nan"), device=DEV)
    hip_ops.gemm(x.to(DEV), W.to(DEV), y, M, N, K, lda, K, N, flags=GEMM_RELU, bias=b.to(DEV), rowmask=mask.to(DEV),
                 rowmask_mod=50)
    _close(y, ref, what="linear fwd")


def test_gemm_row_remap_and_batch():
    import hip_ops
    torch.manual_seed(1)
    na, ne, R, w, nets = 3, 7, 41, 16, 4
    x1 = torch.randn(R * ne, nets * w)
    W = torch.randn(nets, w, w)
    ref = torch.stack([x1.view(R, ne, nets * w)[:, :na, n * w:(n + 1) * w].reshape(R * na, w) @ W[n].t() for n in range(nets)])
    y = torch.zeros(nets, R * na, w, device=DEV)
    hip_ops.gemm(x1.to(DEV), W.to(DEV), y, R * na, w, w, nets * w, w, w, batch=nets, sA=w, sB=w * w, sC=R * na * w,
                 a_map=(na, ne, 0))
    _close(y, ref, what="Q projection (agent rows, batched)")


def test_gemm_dx_relu_bwd_accum_cmap():
    import hip_ops
    torch.manual_seed(2)
    na, ne, R, w = 3, 5, 37, 24
    dkv = torch.randn(R * ne, 2 * w)
    dq = torch.randn(R * na, w)
    Win = torch.randn(3 * w, w) / 5
    x1 = torch.randn(R * ne, w)
    ref = dkv @ Win[w:]
    refq = dq @ Win[:w]
    ref.view(R, ne, w)[:, :na] += refq.view(R, na, w)
    ref = ref * (x1 > 0)
    dx1 = torch.full((R * ne, w), float("nan"), device=DEV)
    Wd, x1d = Win.to(DEV), x1.to(DEV)
    hip_ops.gemm(dkv.to(DEV), Wd[w:], dx1, R * ne, w, 2 * w, 2 * w, w, w, flags=GEMM_B_OUTC | GEMM_RELU_BWD, aux=x1d)
    hip_ops.gemm(dq.to(DEV), Wd, dx1, R * na, w, w, w, w, w, flags=GEMM_B_OUTC | GEMM_RELU_BWD | GEMM_ACCUM, aux=x1d,
                 c_map=(na, ne, 0))
    _close(dx1, ref, what="dx1")


@pytest.mark.parametrize("Rr,N,K,splits,batch", [(5000, 128, 84, 7, 1), (3000, 22, 64, 5, 1), (2000, 96, 48, 1, 2),
                                                 (4097, 192, 64, 16, 3)])
def test_gemm_dw_split_colsum(Rr, N, K, splits, batch):
    import hip_ops
    torch.manual_seed(Rr)
    dy = torch.randn(batch, Rr, N)
    x = torch.randn(batch, Rr, K)
    ref_w = torch.einsum("brn,brk->bnk", dy, x)
    ref_b = dy.sum(1)
    dW = torch.full((batch, N, K), float("nan"), device=DEV)
    db = torch.full((batch, N), float("nan"), device=DEV)
    partial = torch.empty(batch * splits * (N * K + N) + 16, device=DEV)
    hip_ops.gemm(dy.to(DEV), x.to(DEV), dW, N, K, Rr, N, K, K, flags=GEMM_A_OUTC | GEMM_B_OUTC | GEMM_COLSUM_A, colsum=db,
                 partial=partial, batch=batch, splits=splits, sA=Rr * N, sB=Rr * K, sC=N * K, sColsum=N)
    _close(dW, ref_w, tol=5e-5, what="dW")
    _close(db, ref_b, tol=5e-5, what="db")


def test_gemm_dw_bmap_hprev():
    """dW_hh = dgh^T h_prev with h_prev read from the [gb, T1+1, na, H] buffer through a row map."""
    import hip_ops
    torch.manual_seed(5)
    GB, T1, na, H = 3, 6, 4, 64
    hsx = torch.randn(GB, T1 + 1, na, H)
    dgh = torch.randn(GB * T1 * na, 3 * H)
    hprev = hsx[:, :T1].reshape(-1, H)
    ref = dgh.t() @ hprev
    dW = torch.zeros(3 * H, H, device=DEV)
    partial = torch.empty(4 * (3 * H * H + 3 * H), device=DEV)
    hip_ops.gemm(dgh.to(DEV), hsx.to(DEV), dW, 3 * H, H, GB * T1 * na, 3 * H, H, H, flags=GEMM_A_OUTC | GEMM_B_OUTC,
                 partial=partial, splits=4, b_map=(T1 * na, (T1 + 1) * na, 0))
    _close(dW, ref, tol=5e-5, what="dW_hh")


@pytest.mark.parametrize("N,K,splits", [(192, 64, 9), (128, 128, 5), (256, 128, 16), (512, 84, 7), (100, 52, 3)])
def test_gemm_dw_stream_rowmaps(N, K, splits):
    """The streaming weight-gradient kernel (gemm_dw.hip; long reductions, N_out >= 96): row maps on both operands
    (agent rows of a [R, ne] buffer; h_{t-1} rows of the [gb, T1+1, na] buffer), bias gradient, odd row counts."""
    import hip_ops
    torch.manual_seed(N + K + splits)
    na, ne, R = 3, 5, 1499
    dyb = torch.randn(R * ne, N)                 # only the first na of every ne rows are used
    xb = torch.randn(R * ne, K)
    dy = dyb.view(R, ne, N)[:, :na].reshape(R * na, N)
    x = xb.view(R, ne, K)[:, :na].reshape(R * na, K)
    ref_w, ref_b = dy.t() @ x, dy.sum(0)
    dW = torch.full((N, K), float("nan"), device=DEV)
    db = torch.full((N,), float("nan"), device=DEV)
    partial = torch.empty(splits * (N * K + N) + 16, device=DEV)
    hip_ops.gemm(dyb.to(DEV), xb.to(DEV), dW, N, K, R * na, N, K, K, flags=GEMM_A_OUTC | GEMM_B_OUTC | GEMM_COLSUM_A, colsum=db,
                 partial=partial, splits=splits, a_map=(na, ne, 0), b_map=(na, ne, 0))
    _close(dW, ref_w, tol=5e-5, what="dW (stream, row maps)")
    _close(db, ref_b, tol=5e-5, what="db (stream)")


# ---- weight-resident kernel (gemm_wres.hip): taken for M >= 2048, M % 32 == 0, N % 32 == 0, short reductions ----
@pytest.fixture(params=[0, 6], ids=["fp32", "bf16x6"])
def wres_mode(request):
    """Both arithmetic forms of the weight-resident kernel: v_mfma_f32_32x32x2_f32 on the fp32 operands, or the 3-way bf16 split of
    both operands on v_mfma_f32_32x32x16_bf16 x 6 (refil_set_tuning("wres_split", 6) / REFIL_WRES_SPLIT=6)."""
    from refil_amd import _lib
    _lib.check(_lib.lib().refil_set_tuning(b"wres_split", request.param), "refil_set_tuning")
    yield request.param
    _lib.check(_lib.lib().refil_set_tuning(b"wres_split", -1), "refil_set_tuning")


@pytest.mark.parametrize("M,N,K,bt,scale", [(8192, 128, 128, False, 0.0), (8192, 128, 128, True, 0.0), (4096, 256, 84, False, 0.0),
                                            (4096, 128, 256, True, 0.0), (8192, 128, 128, False, 2.0), (4096, 64, 200, False, 1.0)])
def test_wres_split_accuracy(M, N, K, bt, scale):
    """The bf16 x 6 form is as accurate as the fp32 matrix instruction: both against an fp64 product of the same fp32 operands.
    scale > 0: operands with a wide dynamic range (x exp(scale N(0,1))), so that the three pieces of a split sit at very different
    exponents. Bar: rms error <= 1.1 x, max error <= 1.5 x that of the fp32 path (measured: 0.9-1.05 x rms)."""
    import hip_ops
    from refil_amd import _lib
    torch.manual_seed(M + N + K + int(bt))
    x = torch.randn(M, K)
    W = torch.randn(K, N) / math.sqrt(K) if bt else torch.randn(N, K) / math.sqrt(K)
    if scale > 0:
        x = x * torch.exp(scale * torch.randn(M, K))
        W = W * torch.exp(scale * torch.randn(*W.shape))
    ref = x.double() @ (W.double() if bt else W.double().t())
    errs = {}
    try:
        for mode in (0, 6):
            _lib.check(_lib.lib().refil_set_tuning(b"wres_split", mode), "refil_set_tuning")
            y = torch.full((M, N), float("nan"), device=DEV)
            if bt:
                hip_ops.gemm(x.to(DEV), W.to(DEV), y, M, N, K, K, N, N, flags=GEMM_B_OUTC)
            else:
                hip_ops.gemm(x.to(DEV), W.to(DEV), y, M, N, K, K, K, N)
            d = y.cpu().double() - ref
            errs[mode] = (d.pow(2).mean().sqrt().item(), d.abs().max().item())
    finally:
        _lib.check(_lib.lib().refil_set_tuning(b"wres_split", -1), "refil_set_tuning")
    print(f"fp32 MFMA: rms {errs[0][0]:.3e} max {errs[0][1]:.3e}; bf16 x 6: rms {errs[6][0]:.3e} max {errs[6][1]:.3e}; |ref| rms {ref.pow(2).mean().sqrt().item():.3e}")
    assert errs[6][0] <= 1.1 * errs[0][0] and errs[6][1] <= 1.5 * errs[0][1], errs


def test_wres_split_edge_operands():
    """Non-finite, huge, tiny and subnormal operands through refil_gemm in both arithmetic forms (include/refil_hip.h
    refil_set_tuning): (1) a row holding such an operand never disturbs the OTHER rows (bit for bit); (2) Inf / NaN operands give a
    non-finite result in both forms -- the split form NaN where the fp32 instruction gives Inf (x - bf16(x) = Inf - Inf); (3) |x| >= 3.3962e38
    (half-way between bf16's largest finite value and 2^128: the top 0.2 % of the fp32 range) rounds its hi piece to Inf in the split
    form where the fp32 instruction may still give a finite value: the one documented difference; (4) tiny / subnormal operands give finite results within the fp32 unit roundoff of
    the row's scale + the flush-to-zero floor of a subnormal piece."""
    import hip_ops
    from refil_amd import _lib
    torch.manual_seed(11)
    M, N, K = 4096, 128, 128
    x = torch.randn(M, K)
    W = torch.randn(N, K) / math.sqrt(K)
    special = {3: float("inf"), 40: float("-inf"), 77: float("nan"), 130: 3.4e38, 131: -3.3965e38, 200: 3.395e38, 260: 1e-38, 261: 1.2e-38,
               300: 1e-41, 301: -3e-44, 350: 1e30, 351: 1e-30}
    xs = x.clone()
    for r, v in special.items():
        xs[r, 5] = v
    xs[400, :] = 1e-39 * torch.randn(K)              # a whole row of subnormals
    special[400] = None
    clean = torch.ones(M, dtype=torch.bool)
    clean[list(special)] = False
    ref = xs.double() @ W.double().t()
    out = {}
    try:
        for mode in (0, 6):
            _lib.check(_lib.lib().refil_set_tuning(b"wres_split", mode), "refil_set_tuning")
            ya = torch.full((M, N), 5.0, device=DEV); yb = torch.full((M, N), 5.0, device=DEV)
            hip_ops.gemm(x.to(DEV), W.to(DEV), ya, M, N, K, K, K, N)
            hip_ops.gemm(xs.to(DEV), W.to(DEV), yb, M, N, K, K, K, N)
            ya, yb = ya.cpu(), yb.cpu()
            assert torch.equal(ya[clean], yb[clean]), "a special operand in one row changed another row"
            out[mode] = yb
    finally:
        _lib.check(_lib.lib().refil_set_tuning(b"wres_split", -1), "refil_set_tuning")
    for mode in (0, 6):
        y = out[mode]
        for r in (3, 40, 77):
            assert not torch.isfinite(y[r]).any(), f"form {mode}: Inf / NaN operand in row {r} gave a finite result"
        for r in (200, 350, 351, 260, 261, 300, 301, 400):              # finite in both forms, to fp32 accuracy of the row's scale
            assert torch.isfinite(y[r]).all(), f"form {mode}: row {r} not finite"
            scale = (xs[r].abs().double() @ W.abs().double().t())
            err = (y[r].double() - ref[r]).abs()
            assert (err <= 4e-7 * scale + 1e-37).all(), f"form {mode}: row {r} max err {err.max().item():.3e} (scale {scale.max().item():.3e})"
    assert torch.isinf(out[0][3]).all() and torch.isnan(out[6][3]).all()            # Inf: fp32 instruction Inf, split form NaN
    assert torch.isfinite(out[0][130]).all() and not torch.isfinite(out[6][130]).any()     # 3.4e38: rounds to Inf in bf16
    assert torch.isfinite(out[0][131]).all() and not torch.isfinite(out[6][131]).any()


@pytest.mark.parametrize("M,N,K,batch", [(4096, 128, 128, 1), (2048 + 64, 256, 128, 2), (6400, 512, 84, 1), (4096, 192, 64, 1),
                                         (4096, 128, 148, 1), (4096, 512, 200, 1), (2048, 64, 256, 2), (4096, 256, 132, 1),
                                         (2560, 64, 128, 1), (3200, 32, 128, 3), (2048, 128, 16, 1), (4096, 96, 40, 1)])
def test_gemm_wres_forward(M, N, K, batch, wres_mode):
    import hip_ops
    torch.manual_seed(M + N + K)
    x = torch.randn(M, batch * K)
    W = torch.randn(batch, N, K) / math.sqrt(K)
    b = torch.randn(batch, N)
    mask = (torch.rand(48) < 0.3).to(torch.uint8)
    ref = torch.stack([torch.relu(x[:, n * K:(n + 1) * K] @ W[n].t() + b[n]) for n in range(batch)])
    ref[:, mask[torch.arange(M) % 48].bool()] = 0
    y = torch.full((batch, M, N), float("nan"), device=DEV)
    hip_ops.gemm(x.to(DEV), W.to(DEV), y, M, N, K, batch * K, K, N, flags=GEMM_RELU, bias=b.to(DEV), rowmask=mask.to(DEV),
                 rowmask_mod=48, batch=batch, sA=K, sB=N * K, sC=M * N, sBias=N)
    _close(y, ref, what="wres forward")
    # no bias / no mask / no relu, agent-row gather on the input and scatter on the output
    na, ne = 4, 8
    R = M // na
    xe = torch.randn(R * ne, K)
    ref2 = xe.view(R, ne, K)[:, :na].reshape(R * na, K) @ W[0].t()
    y2 = torch.zeros(R * ne, N, device=DEV)
    hip_ops.gemm(xe.to(DEV), W[0].contiguous().to(DEV), y2, R * na, N, K, K, K, N, a_map=(na, ne, 0), c_map=(na, ne, 0))
    _close(y2.view(R, ne, N)[:, :na].reshape(R * na, N), ref2, what="wres forward (row maps)")
    assert y2.view(R, ne, N)[:, na:].abs().max().item() == 0.0


@pytest.mark.parametrize("M,N,K", [(4096, 128, 128), (2304, 128, 32), (4096, 128, 64), (2560, 64, 96), (4096, 256, 128)])
def test_gemm_wres_dx(M, N, K, wres_mode):
    """dx[M,N] = dy[M,K] W[K,N]  (N = layer input width, K = layer output width = the reduction)."""
    import hip_ops
    torch.manual_seed(M + N + K + 1)
    dy = torch.randn(M, K)
    W = torch.randn(K, N) / math.sqrt(K)
    mask = (torch.rand(40) < 0.3).to(torch.uint8)
    ref = dy @ W
    ref[mask[torch.arange(M) % 40].bool()] = 0
    dx = torch.full((M, N), float("nan"), device=DEV)
    hip_ops.gemm(dy.to(DEV), W.to(DEV), dx, M, N, K, K, N, N, flags=GEMM_B_OUTC, rowmask=mask.to(DEV), rowmask_mod=40)
    _close(dx, ref, what="wres dx")


@pytest.mark.parametrize("M,N,K", [(4096, 128, 256), (2048, 64, 192), (4096, 128, 128), (2560, 128, 64), (2048, 64, 52)])
def test_gemm_wres_dx_relu_bwd_accum(M, N, K, wres_mode):
    import hip_ops
    torch.manual_seed(M + N + K + 2)
    na, ne = 4, 8
    R = M // ne
    dy = torch.randn(M, K)
    W = torch.randn(K, N) / math.sqrt(K)
    x1 = torch.randn(M, N)
    dq = torch.randn(R * na, K)
    Wq = torch.randn(K, N) / math.sqrt(K)
    ref = (dy @ W) * (x1 > 0)
    ref.view(R, ne, N)[:, :na] += (dq @ Wq).view(R, na, N) * (x1.view(R, ne, N)[:, :na] > 0)
    dx = torch.full((M, N), float("nan"), device=DEV)
    x1d = x1.to(DEV)
    hip_ops.gemm(dy.to(DEV), W.to(DEV), dx, M, N, K, K, N, N, flags=GEMM_B_OUTC | GEMM_RELU_BWD, aux=x1d)
    if R * na >= 2048:
        hip_ops.gemm(dq.to(DEV), Wq.to(DEV), dx, R * na, N, K, K, N, N, flags=GEMM_B_OUTC | GEMM_RELU_BWD | GEMM_ACCUM, aux=x1d,
                     c_map=(na, ne, 0))
        _close(dx, ref, what="wres dx relu-bwd + accumulate")
    else:
        _close(dx, (dy @ W) * (x1 > 0), what="wres dx relu-bwd")


# ------------------------------------------------------------------------------------------------
# attention core
# ------------------------------------------------------------------------------------------------
def _masks(code, obs, em_t, em0, gb, na):
    """bool [B,T1,na,ne] pre-mask for a variant code (closed forms of SURVEY.md section 8a-5)."""
    B, T1, ne, _ = obs.shape
    inact0 = em0.bool()
    act_pair = (~inact0)[:, :, None] & (~inact0)[:, None, :]
    same = (act_pair & (gb.bool()[:, :, None] == gb.bool()[:, None, :]))[:, None, :na, :].expand(B, T1, na, ne)
    in0 = (~act_pair)[:, None, :na, :].expand(B, T1, na, ne)
    om = obs.bool()[:, :, :na, :]
    emt = em_t.bool()
    ent = emt[:, :, :na, None] | emt[:, :, None, :]
    return {MASK_OBS: om, MASK_OBS_WITHIN: ~same | om, MASK_OBS_INTERACT: same | om, MASK_ENTITY: ent,
            MASK_WITHIN: ~same, MASK_INTERACT: same | in0}[code]


def _attn_ref(q, k, v, masks, heads):
    """q [R,na,w], k,v [R,ne,w], masks list of bool [R,na,ne] -> list of [R,na,w] (attention.py:48-64)."""
    R, na, w = q.shape
    ne = k.shape[1]
    hd = w // heads
    qh = q.reshape(R, na, heads, hd).permute(0, 2, 1, 3)
    kh = k.reshape(R, ne, heads, hd).permute(0, 2, 3, 1)
    vh = v.reshape(R, ne, heads, hd).permute(0, 2, 1, 3)
    logits = (qh @ kh) / torch.tensor(float(hd)).sqrt()
    outs = []
    for m in masks:
        ml = logits.masked_fill(m[:, None], float("-inf"))
        wts = torch.softmax(ml, dim=3)
        wts = torch.where(torch.isnan(wts), torch.zeros_like(wts), wts)
        outs.append((wts @ vh).permute(0, 2, 1, 3).reshape(R, na, w))
    return outs


@pytest.mark.parametrize("ne,na,heads,hd,variants", [
    (32, 16, 4, 32, [MASK_OBS, MASK_OBS_WITHIN, MASK_OBS_INTERACT]),
    (16, 8, 4, 16, [MASK_ENTITY, MASK_WITHIN, MASK_INTERACT]),
    (7, 5, 3, 8, [MASK_OBS, MASK_OBS_WITHIN, MASK_OBS_INTERACT]),
    (6, 3, 2, 4, [MASK_ENTITY]),
    (48, 24, 4, 32, [MASK_ENTITY, MASK_WITHIN, MASK_INTERACT]),
    (64, 32, 2, 32, [MASK_OBS]),
    (40, 40, 1, 64, [MASK_ENTITY, MASK_WITHIN]),      # tile shape without a matrix-core instantiation -> VALU fallback
    (48, 12, 4, 32, [MASK_OBS, MASK_OBS_WITHIN, MASK_OBS_INTERACT]),     # 3 key tiles, one agent tile
    (64, 16, 2, 32, [MASK_ENTITY, MASK_WITHIN, MASK_INTERACT]),          # 4 key tiles, one agent tile
    (20, 4, 2, 12, [MASK_OBS, MASK_OBS_INTERACT]),                       # head dim that is not a whole 16-channel tile
    (30, 20, 4, 24, [MASK_ENTITY]),                                      # two agent tiles, partial second channel tile
])
@pytest.mark.parametrize("force_valu", [False, True])
def test_attention_forward_backward(ne, na, heads, hd, variants, force_valu, monkeypatch):
    monkeypatch.setenv("REFIL_ATTN_VALU", "1" if force_valu else "0")   # matrix-core kernel vs generic VALU fallback
    _attention_case(ne, na, heads, hd, variants, ne * 100 + na)


def _attn_fuzz_cases(n, seed=31):
    import random
    rnd = random.Random(seed)
    groups = [[MASK_OBS, MASK_OBS_WITHIN, MASK_OBS_INTERACT], [MASK_ENTITY, MASK_WITHIN, MASK_INTERACT]]
    out = []
    for i in range(n):
        ne = rnd.randint(1, 64)
        na = rnd.randint(1, min(ne, 32))
        g = rnd.choice(groups)
        out.append((ne, na, rnd.choice([1, 2, 3, 4]), rnd.choice([4, 8, 12, 16, 20, 24, 28, 32]), g[:rnd.randint(1, 3)], 7000 + i))
    return out


@pytest.mark.parametrize("ne,na,heads,hd,variants,seed", _attn_fuzz_cases(int(__import__("os").environ.get("REFIL_FUZZ_ATTN_N", "16"))))
def test_attention_random_shapes(ne, na, heads, hd, variants, seed, monkeypatch):
    """Attention-core fuzz: any entity / agent count up to 64 / 32, head widths 4..32, 1-3 mask variants, whichever kernel the dispatcher
    takes (nine matrix-core tile shapes or the vector-ALU kernel), forward and backward against torch autograd."""
    monkeypatch.delenv("REFIL_ATTN_VALU", raising=False)
    _attention_case(ne, na, heads, hd, variants, seed)


def _attention_case(ne, na, heads, hd, variants, seed):
    import hip_ops
    torch.manual_seed(seed)
    B, T1 = 3, 4
    R, w = B * T1, heads * hd
    q = torch.randn(R, na, w, requires_grad=True)
    kv = torch.randn(R, ne, 2 * w, requires_grad=True)
    obs = (torch.rand(B, T1, ne, ne) < 0.4).to(torch.uint8)
    em_t = (torch.rand(B, T1, ne) < 0.3).to(torch.uint8)
    obs[0, 0, 0, :] = 1          # a fully masked row
    em_t[1, :, :] = 1            # an episode with every entity inactive
    em0 = em_t[:, 0].contiguous()
    gb = (torch.rand(B, ne) < 0.5).to(torch.uint8)
    masks = [_masks(c, obs, em_t, em0, gb, na).reshape(R, na, ne) for c in variants]
    outs = _attn_ref(q, kv[:, :, :w], kv[:, :, w:], masks, heads)
    nv = len(variants)
    dO = torch.randn(nv, R, na, w)
    loss = sum((o * dO[i]).sum() for i, o in enumerate(outs))
    loss.backward()

    qd, kvd = q.detach().reshape(R * na, w).to(DEV), kv.detach().reshape(R * ne, 2 * w).to(DEV)
    d = hip_ops.attn_desc(qd, kvd, kvd[:, w:], w, 2 * w, R, T1, ne, na, heads, hd, variants, obs_mask=obs.to(DEV),
                          ent_mask=em_t.reshape(R, ne).to(DEV), ent_mask0=em0.to(DEV), group_bits=gb.to(DEV))
    O = torch.full((nv, R * na, w), float("nan"), device=DEV)
    hip_ops.attn_forward(d, O, w, R * na * w)
    for i in range(nv):
        _close(O[i].reshape(R, na, w), outs[i], what=f"attn fwd variant {i}")
    dQ = torch.full((R * na, w), float("nan"), device=DEV)
    dKV = torch.full((R * ne, 2 * w), float("nan"), device=DEV)
    hip_ops.attn_backward(d, dO.reshape(nv, R * na, w).to(DEV), w, R * na * w, dQ, dKV, dKV[:, w:])
    _close(dQ.reshape(R, na, w), q.grad, tol=5e-5, what="dQ")
    _close(dKV.reshape(R, ne, 2 * w), kv.grad, tol=5e-5, what="dKV")
    assert torch.isfinite(O).all() and torch.isfinite(dQ).all() and torch.isfinite(dKV).all()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("B,T1,ne,na,w,codes", [(3, 4, 6, 3, 16, (MASK_OBS, MASK_OBS_WITHIN, MASK_OBS_INTERACT)),
                                                (2, 3, 32, 16, 128, (MASK_ENTITY, MASK_WITHIN, MASK_INTERACT)),
                                                (2, 5, 7, 5, 24, (MASK_ENTITY,))])
def test_pool_forward_backward(mode, B, T1, ne, na, w, codes):
    """EntityPoolingLayer core (attention.py:114-123): masked entities enter as zeros, mean over ALL entities, max with
    torch's first-maximum gradient routing -- against autograd."""
    import hip_ops
    torch.manual_seed(B * 100 + ne + w + mode)
    R = B * T1
    E = torch.randn(R * ne, 2 * w)                       # in_trans output lives in the first w columns of a [.., 2w] buffer
    if mode == 2:
        E[:, :w] -= 0.7                                   # many all-negative columns: the masked zeros win the max
    obs = (torch.rand(B, T1, ne, ne) < 0.4).to(torch.uint8)
    em = (torch.rand(B, T1, ne) < 0.3).to(torch.uint8)
    em0 = em[:, 0].contiguous()
    gb = (torch.rand(B, ne) < 0.5).to(torch.uint8)
    Er = E[:, :w].clone().view(R, ne, w).requires_grad_(True)
    outs = []
    for c in codes:
        pm = _masks(c, obs, em, em0, gb, na).reshape(R, na, ne)
        rep = Er[:, None].expand(-1, na, -1, -1).masked_fill(pm[..., None], 0.0)
        outs.append(rep.max(dim=2)[0] if mode == 2 else rep.mean(dim=2))
    ref = torch.stack(outs)                               # [nvar, R, na, w]
    dO = torch.randn_like(ref)
    ref.backward(dO)
    Ed = E.to(DEV)
    d = hip_ops.attn_desc(Ed, Ed, Ed, 2 * w, 2 * w, R, T1, ne, na, 1, w, list(codes), obs_mask=obs.to(DEV), ent_mask=em.view(R, ne).to(DEV),
                          ent_mask0=em0.to(DEV), group_bits=gb.to(DEV))
    O = torch.full((len(codes), R * na, w), float("nan"), device=DEV)
    hip_ops.pool_forward(d, mode, O, w, R * na * w)
    _close(O.view_as(ref), ref.detach(), what="pool fwd")
    dE = torch.full((R * ne, 2 * w), float("nan"), device=DEV)
    hip_ops.pool_backward(d, mode, dO.reshape(len(codes), R * na, w).to(DEV), w, R * na * w, dE)
    _close(dE[:, :w], Er.grad.reshape(R * ne, w), what="pool bwd")


# ------------------------------------------------------------------------------------------------
# persistent GRU
# ------------------------------------------------------------------------------------------------
@pytest.fixture
def set_tuning():
    """refil_set_tuning for the duration of a test (every knob back to its built-in default afterwards)"""
    from refil_amd import _lib, tuning
    touched = []

    def apply(**kw):
        for k, v in kw.items():
            assert v == -1 or v in tuning.PARITY_TESTED[k], (k, v)
            _lib.check(_lib.lib().refil_set_tuning(k.encode(), int(v)), "refil_set_tuning")
            touched.append(k)
    yield apply
    for k in touched:
        _lib.check(_lib.lib().refil_set_tuning(k.encode(), -1), "refil_set_tuning")


# pd: the recurrences' prefetch distance ("gru_pd": 2 selects OTHER template instantiations of the 4-row kernels -- a different
# register ring and vmcnt schedule -- and is what the autotuned bench line runs at cfg-T; -1 = the built-in default, 4)
@pytest.mark.parametrize("pd", [-1, 2, 4])
@pytest.mark.parametrize("GB,T1,na,H,valu", [(6, 9, 16, 64, 0), (5, 4, 3, 64, 0), (2, 1, 8, 64, 0), (3, 21, 5, 64, 0),
                                             (5, 6, 7, 32, 0), (2, 1, 8, 32, 0), (4, 7, 6, 128, 0), (3, 5, 16, 128, 0),
                                             (3, 23, 9, 32, 0), (2, 30, 16, 128, 0),
                                             (6, 9, 16, 64, 3), (3, 21, 5, 64, 3)])
def test_gru_forward_backward(GB, T1, na, H, valu, pd, monkeypatch, set_tuning):
    """valu = 3: the opt-in vector-ALU form of the recurrent products (REFIL_GRU_VALU, gru.hip), same reference.
    Reference: torch autograd through the oracle's GRUCell restatement (entity_rnn_agent.py:49-55)."""
    import hip_ops
    from oracle.refil_oracle import gru_cell
    if valu and pd != -1:
        pytest.skip("the vector-ALU form has one prefetch distance")
    monkeypatch.setenv("REFIL_GRU_VALU", str(valu))
    set_tuning(gru_pd=pd)
    torch.manual_seed(GB * 7 + T1)
    NR = GB * na
    w_ih = (torch.randn(3 * H, H) / 8).requires_grad_(True)
    w_hh = (torch.randn(3 * H, H) / 8).requires_grad_(True)
    b_ih = (torch.randn(3 * H) / 8).requires_grad_(True)
    b_hh = (torch.randn(3 * H) / 8).requires_grad_(True)
    x = torch.randn(GB, T1, na, H, requires_grad=True)
    h0 = torch.randn(GB, na, H) / 2
    h = h0.reshape(NR, H)
    hs = []
    for t in range(T1):
        h = gru_cell(x[:, t].reshape(NR, H), h, w_ih, w_hh, b_ih, b_hh)
        hs.append(h.reshape(GB, na, H))
    hs = torch.stack(hs, 1)
    dhs = torch.randn(GB, T1, na, H)
    (hs * dhs).sum().backward()

    gi = (x.detach().reshape(-1, H) @ w_ih.detach().t() + b_ih.detach()).to(DEV)
    hsx = torch.full((GB, T1 + 1, na, H), float("nan"), device=DEV)
    hsx[:, 0] = h0.to(DEV)
    saves = [torch.full((GB * T1 * na, H), float("nan"), device=DEV) for _ in range(4)]
    whh, bhh = w_hh.detach().to(DEV), b_hh.detach().to(DEV)
    d = hip_ops.gru_desc(gi, hsx, whh, bhh, NR, T1, na, H=H, saves=saves)
    hip_ops.gru_forward(d)
    _close(hsx[:, 1:], hs, what="gru hs")
    dgi = torch.full((GB * T1 * na, 3 * H), float("nan"), device=DEV)
    dgh = torch.full((GB * T1 * na, H), float("nan"), device=DEV)           # the n block of d(gh) only (refil_gru_desc.dgh)
    d = hip_ops.gru_desc(gi, hsx, whh, bhh, NR, T1, na, H=H, saves=saves, dhs=dhs.to(DEV), dgi=dgi, dgh=dgh)
    hip_ops.gru_backward(d)
    dgi_c = dgi.cpu()
    dgh_c = torch.cat([dgi_c[:, :2 * H], dgh.cpu()], dim=1)                  # d(gh) = (dgi_r, dgi_z, dgh_n)
    _close(dgi_c.sum(0), b_ih.grad, tol=1e-4, what="db_ih")
    _close(dgh_c.sum(0), b_hh.grad, tol=1e-4, what="db_hh")
    _close(dgi_c.t() @ x.detach().reshape(-1, H), w_ih.grad, tol=1e-4, what="dW_ih")
    hprev = hsx[:, :T1].reshape(-1, H).cpu()
    _close(dgh_c.t() @ hprev, w_hh.grad, tol=1e-4, what="dW_hh")
    _close((dgi_c @ w_ih.detach()).reshape(GB, T1, na, H), x.grad, tol=1e-4, what="dx")
    # inference variant (no saves) gives the same hidden states
    hsx2 = torch.zeros_like(hsx)
    hsx2[:, 0] = h0.to(DEV)
    hip_ops.gru_forward(hip_ops.gru_desc(gi, hsx2, whh, bhh, NR, T1, na, H=H))
    assert torch.equal(hsx2[:, 1:], hsx[:, 1:])


# ------------------------------------------------------------------------------------------------
# row lists (refil_gemm_desc.row_index / row_count) and row skipping (attention / GRU t_last, dead rows)
# ------------------------------------------------------------------------------------------------
def _row_list(M, frac, seed, trash):
    """sorted random subset of [0, M) as a device list padded to a multiple of 64 with `trash`, and its device count"""
    g = torch.Generator().manual_seed(seed)
    keep = torch.nonzero(torch.rand(M, generator=g) < frac).flatten().to(torch.int32)
    n = keep.numel()
    padded = torch.full(((n + 63) // 64 * 64 + 128,), trash, dtype=torch.int32)
    padded[:n] = keep
    return keep.long(), padded.to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV)


@pytest.mark.parametrize("M,N,K,batch,frac", [(4096, 128, 84, 1, 0.45), (6400, 256, 128, 2, 0.6), (2048, 64, 52, 1, 0.3),
                                              (4096, 128, 148, 1, 0.5), (4096, 512, 196, 1, 0.4),
                                              (4096, 512, 84, 1, 0.0), (4096, 128, 64, 1, 1.0)])
def test_gemm_wres_forward_row_list(M, N, K, batch, frac, wres_mode):
    """x W^T over a device-side row list: listed rows are computed, every other row of C stays untouched."""
    import hip_ops
    torch.manual_seed(M + N + K)
    x = torch.randn(M + 8, batch * K)
    W = torch.randn(batch, N, K) / math.sqrt(K)
    b = torch.randn(batch, N)
    keep, lst, cnt = _row_list(M, frac, M + N, trash=M)
    ref = torch.full((batch, M + 8, N), 7.0)
    for n in range(batch):
        ref[n, keep] = torch.relu(x[keep, n * K:(n + 1) * K] @ W[n].t() + b[n])
    y = torch.full((batch, M + 8, N), 7.0, device=DEV)
    hip_ops.gemm(x.to(DEV), W.to(DEV), y, M, N, K, batch * K, K, N, flags=GEMM_RELU, bias=b.to(DEV), batch=batch, sA=K,
                 sB=N * K, sC=(M + 8) * N, sBias=N, row_index=lst, row_count=cnt)
    _close(y[:, :M], ref[:, :M], what="wres forward (row list)")
    assert torch.equal(y[:, :M].cpu() == 7.0, ref[:, :M] == 7.0), "rows outside the list were written"
    # agent-row list on top of the agent-row map (the Q projection)
    na, ne = 4, 8
    R = M // na
    xe = torch.randn(R * ne + 8, K)
    keep2, lst2, cnt2 = _row_list(R * na, max(frac, 0.2), M + 1, trash=R * na)
    ref2 = torch.zeros(R * na + 8, N)
    ref2[keep2] = xe[:R * ne].view(R, ne, K)[:, :na].reshape(R * na, K)[keep2] @ W[0].t()
    y2 = torch.zeros(R * na + 8, N, device=DEV)
    hip_ops.gemm(xe.to(DEV), W[0].contiguous().to(DEV), y2, R * na, N, K, K, K, N, a_map=(na, ne, 0), row_index=lst2, row_count=cnt2)
    _close(y2[:R * na], ref2[:R * na], what="wres forward (row list + row map)")


@pytest.mark.parametrize("M,N,K,frac", [(4096, 128, 256, 0.45), (2048, 64, 128, 0.5), (4096, 128, 128, 0.7)])
def test_gemm_wres_dx_relu_bwd_row_list(M, N, K, frac, wres_mode):
    import hip_ops
    torch.manual_seed(M + N + K + 5)
    na, ne = 4, 8
    R = M // ne
    dy = torch.randn(M + 8, K)
    W = torch.randn(K, N) / math.sqrt(K)
    x1 = torch.randn(M + 8, N)
    dq = torch.randn(R * na + 8, K)
    Wq = torch.randn(K, N) / math.sqrt(K)
    keep, lst, cnt = _row_list(M, frac, M + K, trash=M)
    ref = torch.full((M + 8, N), 3.0)
    ref[keep] = (dy[keep] @ W) * (x1[keep] > 0)
    dx = torch.full((M + 8, N), 3.0, device=DEV)
    x1d = x1.to(DEV)
    hip_ops.gemm(dy.to(DEV), W.to(DEV), dx, M, N, K, K, N, N, flags=GEMM_B_OUTC | GEMM_RELU_BWD, aux=x1d, row_index=lst, row_count=cnt)
    _close(dx[:M], ref[:M], what="wres dx relu-bwd (row list)")
    if R * na >= 2048:                 # accumulate the query part on listed agent rows (scatter through c_map)
        ent = torch.zeros(M, dtype=torch.bool); ent[keep] = True
        ag = ent.view(R, ne)[:, :na].reshape(-1)                     # listed agents = agents whose entity row is listed
        keepa = torch.nonzero(ag).flatten()
        n = keepa.numel()
        lsta = torch.full(((n + 63) // 64 * 64 + 128,), R * na, dtype=torch.int32); lsta[:n] = keepa.to(torch.int32)
        add = torch.zeros(R, ne, N)
        add[:, :na] = ((dq[:R * na] @ Wq) * (x1[:M].view(R, ne, N)[:, :na].reshape(R * na, N) > 0)).view(R, na, N) * ag.view(R, na, 1)
        hip_ops.gemm(dq.to(DEV), Wq.to(DEV), dx, R * na, N, K, K, N, N, flags=GEMM_B_OUTC | GEMM_RELU_BWD | GEMM_ACCUM, aux=x1d,
                     c_map=(na, ne, 0), row_index=lsta.to(DEV), row_count=torch.tensor([n], dtype=torch.int32, device=DEV))
        _close(dx[:M], ref[:M] + add.view(M, N), what="wres dx accumulate (row list + c_map)")


@pytest.mark.parametrize("Rr,N,K,splits,frac", [(40000, 256, 128, 64, 0.45), (20000, 64, 52, 32, 0.6), (8192, 512, 84, 16, 0.3),
                                                (9000, 128, 128, 8, 0.0)])
def test_gemm_dw_stream_row_list(Rr, N, K, splits, frac):
    """dW = dy^T x and db = colsum(dy) over the listed rows only (everything else may hold garbage)."""
    import hip_ops
    torch.manual_seed(Rr + N + K)
    dy = torch.randn(Rr + 8, N)
    x = torch.randn(Rr + 8, K)
    keep, lst, cnt = _row_list(Rr, frac, Rr + N, trash=Rr)
    ref_w = dy[keep].double().t() @ x[keep].double()
    ref_b = dy[keep].double().sum(0)
    dead = torch.ones(Rr + 8, dtype=torch.bool); dead[keep] = False
    dyd, xd = dy.clone(), x.clone()
    dyd[dead] = float("nan"); xd[dead] = float("nan")              # rows outside the list must never be touched
    dW = torch.full((N, K), float("nan"), device=DEV)
    db = torch.full((N,), float("nan"), device=DEV)
    partial = torch.zeros(splits * (N * K + N), device=DEV)
    hip_ops.gemm(dyd.to(DEV), xd.to(DEV), dW, N, K, Rr, N, K, K, flags=GEMM_A_OUTC | GEMM_B_OUTC | GEMM_COLSUM_A, colsum=db,
                 partial=partial, splits=splits, row_index=lst, row_count=cnt)
    scale = max(ref_w.abs().max().item(), 1.0)
    assert (dW.cpu().double() - ref_w).abs().max().item() <= 5e-5 * scale
    assert (db.cpu().double() - ref_b).abs().max().item() <= 5e-5 * max(ref_b.abs().max().item(), 1.0)


def test_gemm_row_list_rejected_by_the_tiled_kernel():
    import hip_ops
    x = torch.randn(100, 16, device=DEV); W = torch.randn(8, 16, device=DEV); y = torch.zeros(100, 8, device=DEV)
    lst = torch.arange(128, dtype=torch.int32, device=DEV); cnt = torch.tensor([100], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="row lists"):
        hip_ops.gemm(x, W, y, 100, 8, 16, 16, 16, 8, row_index=lst, row_count=cnt)


def test_attention_row_skipping():
    """t_last leaves the rows of finished episodes untouched; dead K/V and Q rows enter as zeros whatever they hold, are not
    fetched, and their gradient rows (exact zeros nobody reads) are not written."""
    import hip_ops
    torch.manual_seed(5)
    B, T1, ne, na, heads, hd = 3, 6, 32, 16, 4, 32
    R, w = B * T1, heads * hd
    em = torch.zeros(B, T1, ne, dtype=torch.uint8)
    em[:, :, 10:16] = 1; em[:, :, 25:] = 1                          # padded agents / enemies
    Q, K, V = torch.randn(R * na, w), torch.randn(R * ne, 2 * w), None
    t_last = torch.tensor([5, 2, -1], dtype=torch.int32)
    kv_dead = em.reshape(R * ne).clone()
    q_dead = em[:, :, :na].reshape(R * na).clone()
    Kd, Qd = K.clone(), Q.clone()
    Kd[kv_dead.bool()] = float("nan"); Qd[q_dead.bool()] = float("nan")       # garbage in the skipped producers' rows
    Kz, Qz = K.clone(), Q.clone()
    Kz[kv_dead.bool()] = 0; Qz[q_dead.bool()] = 0
    dO = torch.randn(R * na, w)
    outs = []
    for (Qx, Kx, skip) in ((Qz, Kz, False), (Qd, Kd, True)):
        Qg, Kg = Qx.to(DEV), Kx.to(DEV)
        d = hip_ops.attn_desc(Qg, Kg, Kg[:, w:], w, 2 * w, R, T1, ne, na, heads, hd, [MASK_ENTITY], ent_mask=em.view(R, ne).to(DEV))
        if skip:
            hip_ops.attn_skip(d, t_last.to(DEV), kv_dead.to(DEV), q_dead.to(DEV))
        O = torch.full((R * na, w), 5.0, device=DEV)
        hip_ops.attn_forward(d, O, w, R * na * w)
        dQ = torch.full((R * na, w), 5.0, device=DEV); dKV = torch.full((R * ne, 2 * w), 5.0, device=DEV)
        hip_ops.attn_backward(d, dO.to(DEV), w, R * na * w, dQ, dKV, dKV[:, w:])
        outs.append((O.cpu().view(B, T1, na, w), dQ.cpu().view(B, T1, na, w), dKV.cpu().view(B, T1, ne, 2 * w)))
    (O0, dQ0, dK0), (O1, dQ1, dK1) = outs
    for b in range(B):
        tl = int(t_last[b])
        live_q = ~em[b, :tl + 1, :na].bool()
        assert torch.equal(O1[b, :tl + 1][live_q], O0[b, :tl + 1][live_q])
        assert torch.equal(dQ1[b, :tl + 1][live_q], dQ0[b, :tl + 1][live_q])
        live_k = ~em[b, :tl + 1].bool()
        assert torch.equal(dK1[b, :tl + 1][live_k], dK0[b, :tl + 1][live_k])
        assert (dK0[b, :tl + 1][~live_k] == 0).all()                   # what the dense schedule writes there ...
        assert (dK1[b, :tl + 1][~live_k] == 5.0).all()                 # ... is left unwritten when the rows are declared dead
        assert (dQ1[b, :tl + 1][~live_q] == 5.0).all()
        assert torch.isfinite(O1[b, :tl + 1]).all() and torch.isfinite(dK1[b, :tl + 1]).all()
        assert (O1[b, tl + 1:] == 5.0).all() and (dK1[b, tl + 1:] == 5.0).all() and (dQ1[b, tl + 1:] == 5.0).all()


@pytest.mark.parametrize("pd", [-1, 2])
@pytest.mark.parametrize("T1,t_last,H", [(7, (6, 3, 0), 64), (19, (18, 9, 2), 64), (13, (5, 12, 0), 32), (11, (10, 1, 6), 128)])
def test_gru_time_bounds(T1, t_last, H, pd, set_tuning):
    """t_last: the recurrence of an episode stops after its last contributing step; BPTT starts there and zero-fills.
    Both prefetch distances, all three hidden sizes, episodes ending at every position of the unrolled ring."""
    import hip_ops
    torch.manual_seed(11)
    set_tuning(gru_pd=pd)
    G, B, na = 2, 3, 16
    GB, NR = G * B, G * B * na
    t_last = torch.tensor(t_last, dtype=torch.int32)
    gi = torch.randn(GB * T1 * na, 3 * H, device=DEV)
    whh, bhh = torch.randn(3 * H, H, device=DEV) / 8, torch.randn(3 * H, device=DEV) / 8
    h0 = torch.randn(GB, na, H, device=DEV) / 2
    dhs = torch.randn(GB, T1, na, H, device=DEV)
    for gb in range(GB):
        dhs[gb, int(t_last[gb % B]) + 1:] = 0                       # no loss term reaches the skipped steps
    res = []
    for skip in (False, True):
        hsx = torch.full((GB, T1 + 1, na, H), 9.0, device=DEV); hsx[:, 0] = h0
        saves = [torch.full((GB * T1 * na, H), 9.0, device=DEV) for _ in range(4)]
        d = hip_ops.gru_desc(gi, hsx, whh, bhh, NR, T1, na, H=H, saves=saves)
        if skip:
            hip_ops.gru_skip(d, t_last.to(DEV), B)
        hip_ops.gru_forward(d)
        dgi = torch.full((GB * T1 * na, 3 * H), 9.0, device=DEV); dgh = torch.full((GB * T1 * na, H), 9.0, device=DEV)
        d = hip_ops.gru_desc(gi, hsx, whh, bhh, NR, T1, na, H=H, saves=saves, dhs=dhs, dgi=dgi, dgh=dgh)
        if skip:
            hip_ops.gru_skip(d, t_last.to(DEV), B)
        hip_ops.gru_backward(d)
        res.append((hsx.cpu(), dgi.cpu().view(GB, T1, na, 3 * H), dgh.cpu().view(GB, T1, na, H)))
    (h_full, gi_full, gh_full), (h_skip, gi_skip, gh_skip) = res
    for gb in range(GB):
        tl = int(t_last[gb % B])
        assert torch.equal(h_skip[gb, :tl + 2], h_full[gb, :tl + 2])
        assert (h_skip[gb, tl + 2:] == 9.0).all()
        assert torch.equal(gi_skip[gb, :tl + 1], gi_full[gb, :tl + 1]) and torch.equal(gh_skip[gb, :tl + 1], gh_full[gb, :tl + 1])
        assert (gi_skip[gb, tl + 1:] == 0).all() and (gh_skip[gb, tl + 1:] == 0).all()
        assert gi_full[gb, tl + 1:].abs().max().item() == 0.0 if tl + 1 < T1 else True


@pytest.fixture(params=[0, 6], ids=["dw_fp32", "dw_bf16x6"])
def dw_mode(request):
    """Both arithmetic forms of the large weight gradients (refil_set_tuning("dw_split", .) / REFIL_DW_SPLIT)."""
    from refil_amd import _lib
    _lib.check(_lib.lib().refil_set_tuning(b"dw_split", request.param), "refil_set_tuning")
    yield request.param
    _lib.check(_lib.lib().refil_set_tuning(b"dw_split", -1), "refil_set_tuning")


@pytest.mark.parametrize("Rr,N,K,batch,splits,frac,bmap", [
    (20000, 256, 128, 1, 24, 0.0, None), (20000 + 61, 256, 128, 4, 7, 0.45, None), (9000 + 3, 128, 128, 1, 16, 0.0, None),
    (9000 + 130, 128, 128, 2, 5, 0.6, None), (4096, 384, 128, 1, 2, 0.0, None), (70000, 256, 128, 1, 96, 0.5, None),
    (12000 + 7, 512, 84, 1, 12, 0.5, None), (9000, 128, 84, 1, 8, 0.0, None), (8000 + 19, 128, 100, 2, 6, 0.4, None),
    (12000 + 5, 128, 128, 1, 10, 0.5, (16, 32, 0)), (6000 + 44, 256, 128, 2, 4, 0.0, (16, 32, 0)), (5184, 128, 96, 1, 3, 0.3, (1296, 1312, 0))])
def test_gemm_dws_accuracy(Rr, N, K, batch, splits, frac, bmap):
    """gemm_dws_kernel (bf16 x 6 weight gradient, outputs 65 .. 128 columns wide): against an fp64 product, next to the fp32-instruction
    kernel on the same operands -- rms error <= 2 x, max error <= 3 x the fp32 path's and <= 2e-6 of the result's rms; column sums (bias gradient) to 2e-6 of their scale.
    Covers whole ring periods, the fp32 tail (row counts that are not multiples of 64), row lists, batches, one and two tile rows, padded
    column tiles (84 / 96 / 100 columns) and a row map on the x rows (agents' rows in entity-major storage)."""
    import hip_ops
    from refil_amd import _lib
    torch.manual_seed(Rr + N + batch)
    dy = torch.randn(Rr + 8, batch * N) * torch.exp(0.5 * torch.randn(Rr + 8, 1))
    if bmap:
        grp, gstride, off = bmap
        nphys = (Rr + 8) // grp * gstride + grp + off + 8
        xb = torch.randn(nphys, batch * K)
        rows = torch.arange(Rr + 8)
        x = xb[rows + (rows // grp) * (gstride - grp) + off]
    else:
        xb = x = torch.randn(Rr + 8, batch * K)
    if frac > 0:
        keep, lst, cnt = _row_list(Rr, frac, N + K, trash=Rr)
    else:
        keep, lst, cnt = torch.arange(Rr), None, None
    ref_w = torch.stack([dy[keep, n * N:(n + 1) * N].double().t() @ x[keep, n * K:(n + 1) * K].double() for n in range(batch)])
    ref_b = torch.stack([dy[keep, n * N:(n + 1) * N].double().sum(0) for n in range(batch)])
    errs = {}
    try:
        for mode in (0, 6):
            _lib.check(_lib.lib().refil_set_tuning(b"dw_split", mode), "refil_set_tuning")
            dW = torch.full((batch, N, K), float("nan"), device=DEV)
            db = torch.full((batch, N), float("nan"), device=DEV)
            partial = torch.zeros(batch * splits * (N * K + N), device=DEV)
            hip_ops.gemm(dy.to(DEV), xb.to(DEV), dW, N, K, Rr, batch * N, batch * K, K, flags=GEMM_A_OUTC | GEMM_B_OUTC | GEMM_COLSUM_A,
                         colsum=db, partial=partial, splits=splits, batch=batch, sA=N, sB=K, sC=N * K, sColsum=N, row_index=lst, row_count=cnt,
                         b_map=bmap or (0, 0, 0))
            d = dW.cpu().double() - ref_w
            errs[mode] = (d.pow(2).mean().sqrt().item(), d.abs().max().item())
            assert (db.cpu().double() - ref_b).abs().max().item() <= 2e-6 * max(ref_b.abs().max().item(), 1.0) * max(1.0, (len(keep) / 20000) ** 0.5), mode
    finally:
        _lib.check(_lib.lib().refil_set_tuning(b"dw_split", -1), "refil_set_tuning")
    ref_rms = ref_w.pow(2).mean().sqrt().item()
    print(f"fp32 MFMA: rms {errs[0][0]:.3e} max {errs[0][1]:.3e}; bf16 x 6: rms {errs[6][0]:.3e} max {errs[6][1]:.3e}; |ref| rms {ref_rms:.3e}")
    # (the products are fp32-accurate; what differs is the length of the accumulation chains -- a workgroup's whole row range in one
    # accumulator here, a quarter of it per wave in the fp32 kernel: measured 1.0-1.4 x its rms error)
    assert errs[6][0] <= 2.0 * errs[0][0] and errs[6][1] <= 3.0 * errs[0][1] and errs[6][0] <= 2e-6 * ref_rms, errs


@pytest.mark.parametrize("count", [0, 1, 5, 63, 64, 65, 127, 128, 200, 4 * 64 * 3, 4 * 64 * 3 + 1])
@pytest.mark.parametrize("N,bmap", [(256, None), (128, (16, 32, 0))])
def test_gemm_dws_short_lists(count, N, bmap, dw_mode):
    """Row lists far shorter than the launch was sized for (capacity 6000 rows, 3 splits): empty, shorter than one 16-row step, exactly /
    just past whole ring periods of a workgroup's range -- the empty-resource requests, the fp32 tail and the column sums at the edges."""
    import hip_ops
    torch.manual_seed(count + N)
    cap, K, splits = 6000, 128, 3
    dy = torch.randn(cap + 8, N)
    nphys = (cap + 8) // 16 * 32 + 64 if bmap else cap + 8
    xb = torch.randn(nphys, K)
    rows = torch.arange(cap + 8)
    x = xb[rows + (rows // 16) * 16] if bmap else xb
    keep = torch.sort(torch.randperm(cap)[:count]).values
    lst = torch.full(((count + 63) // 64 * 64 + 128,), cap, dtype=torch.int32)
    lst[:count] = keep.to(torch.int32)
    cnt = torch.tensor([count], dtype=torch.int32, device=DEV)
    ref_w = dy[keep].double().t() @ x[keep].double()
    ref_b = dy[keep].double().sum(0)
    dW = torch.full((N, K), float("nan"), device=DEV)
    db = torch.full((N,), float("nan"), device=DEV)
    partial = torch.zeros(splits * (N * K + N), device=DEV)
    hip_ops.gemm(dy.to(DEV), xb.to(DEV), dW, N, K, cap, N, K, K, flags=GEMM_A_OUTC | GEMM_B_OUTC | GEMM_COLSUM_A, colsum=db, partial=partial,
                 splits=splits, row_index=lst.to(DEV), row_count=cnt, b_map=bmap or (0, 0, 0))
    assert (dW.cpu().double() - ref_w).abs().max().item() <= 5e-5 * max(ref_w.abs().max().item(), 1.0)
    assert (db.cpu().double() - ref_b).abs().max().item() <= 5e-5 * max(ref_b.abs().max().item(), 1.0)


@pytest.mark.parametrize("N,K", [(256, 128), (512, 84), (128, 128), (192, 64), (64, 128), (32, 128), (22, 64), (96, 96), (300, 40),
                                 (128, 20), (64, 52)])
@pytest.mark.parametrize("batch,use_list", [(1, False), (3, False), (1, True)])
def test_gemm_dw4_tiles(N, K, batch, use_list, dw_mode):
    """gemm_dw4.hip (every wave-tile shape TI x TJ): dW = dy^T x per batch (+ db), optionally over a row list. dw_mode: the fp32
    instruction everywhere / the bf16 x 6 kernel (gemm_dws_kernel) for the 128-column outputs."""
    import hip_ops
    torch.manual_seed(N * 7 + K + batch)
    Rr, splits = 5000 + 37, 9
    dy = torch.randn(Rr + 8, batch * N)
    x = torch.randn(Rr + 8, batch * K)
    if use_list:
        keep, lst, cnt = _row_list(Rr, 0.5, N + K, trash=Rr)
    else:
        keep, lst, cnt = torch.arange(Rr), None, None
    ref_w = torch.stack([dy[keep, n * N:(n + 1) * N].double().t() @ x[keep, n * K:(n + 1) * K].double() for n in range(batch)])
    ref_b = torch.stack([dy[keep, n * N:(n + 1) * N].double().sum(0) for n in range(batch)])
    dW = torch.full((batch, N, K), float("nan"), device=DEV)
    db = torch.full((batch, N), float("nan"), device=DEV)
    partial = torch.zeros(batch * splits * (N * K + N), device=DEV)
    hip_ops.gemm(dy.to(DEV), x.to(DEV), dW, N, K, Rr, batch * N, batch * K, K, flags=GEMM_A_OUTC | GEMM_B_OUTC | GEMM_COLSUM_A,
                 colsum=db, partial=partial, splits=splits, batch=batch, sA=N, sB=K, sC=N * K, sColsum=N, row_index=lst, row_count=cnt)
    scale = max(ref_w.abs().max().item(), 1.0)
    assert (dW.cpu().double() - ref_w).abs().max().item() <= 5e-5 * scale
    assert (db.cpu().double() - ref_b).abs().max().item() <= 5e-5 * max(ref_b.abs().max().item(), 1.0)


@pytest.mark.parametrize("ne,na,heads,hd,variants", [(32, 16, 4, 32, (MASK_OBS, MASK_OBS_WITHIN, MASK_OBS_INTERACT)),
                                                     (16, 8, 4, 16, (MASK_ENTITY, MASK_WITHIN, MASK_INTERACT)), (48, 24, 4, 32, (MASK_ENTITY,))])
def test_attention_precomputed_mask_words(ne, na, heads, hd, variants):
    """refil_attn_mask_words + mask_words / row_bits in the desc == the in-kernel mask phase, bit for bit."""
    import hip_ops
    torch.manual_seed(ne + na)
    B, T1 = 2, 5
    R, w = B * T1, heads * hd
    em = (torch.rand(B, T1, ne) < 0.3).to(torch.uint8)
    obs = (torch.rand(B, T1, ne, ne) < 0.4).to(torch.uint8)
    gb = (torch.rand(B, ne) < 0.5).to(torch.uint8)
    Q, K = torch.randn(R * na, w, device=DEV), torch.randn(R * ne, 2 * w, device=DEV)
    dO = torch.randn(len(variants), R * na, w, device=DEV)
    res = []
    for pre in (False, True):
        d = hip_ops.attn_desc(Q, K, K[:, w:], w, 2 * w, R, T1, ne, na, heads, hd, list(variants), obs_mask=obs.to(DEV),
                              ent_mask=em.view(R, ne).to(DEV), ent_mask0=em[:, 0].contiguous().to(DEV), group_bits=gb.to(DEV))
        if pre:
            hip_ops.attn_mask_words(d, na)
        O = torch.zeros(len(variants), R * na, w, device=DEV)
        hip_ops.attn_forward(d, O, w, R * na * w)
        dQ = torch.zeros(R * na, w, device=DEV); dKV = torch.zeros(R * ne, 2 * w, device=DEV)
        hip_ops.attn_backward(d, dO, w, R * na * w, dQ, dKV, dKV[:, w:])
        res.append((O.cpu(), dQ.cpu(), dKV.cpu()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,T1,ne,na,heads,hd,pre", [(24, 40, 32, 16, 4, 32, True), (24, 40, 32, 16, 4, 32, False), (40, 30, 16, 8, 4, 16, True),
                                                       (12, 64, 48, 24, 4, 32, True), (300, 3, 16, 8, 2, 16, True)])
def test_attention_persistent_walk(B, T1, ne, na, heads, hd, pre):
    """More live rows than resident workgroups: every workgroup walks several rows and its waves pipeline their jobs
    across them; ragged episode ends (t_last), dead K/V and Q rows, three variants, against the fp32 torch reference."""
    import hip_ops
    torch.manual_seed(B + T1)
    R, w = B * T1, heads * hd
    variants = [MASK_OBS, MASK_OBS_WITHIN, MASK_OBS_INTERACT]
    q = torch.randn(R, na, w, requires_grad=True)
    kv = torch.randn(R, ne, 2 * w, requires_grad=True)
    obs = (torch.rand(B, T1, ne, ne) < 0.4).to(torch.uint8)
    em = torch.zeros(B, T1, ne, dtype=torch.uint8)
    em[:, :, na - 2:na] = 1; em[:, :, ne - 5:] = 1                   # padded agents / enemies: dead Q and K/V rows
    em[1::3, T1 // 2:, 0] = 1                                        # an agent dying mid-episode
    obs = obs | em[:, :, :, None] | em[:, :, None, :]                # (dead entities are never observed)
    em0 = em[:, 0].contiguous()
    gb = (torch.rand(B, ne) < 0.5).to(torch.uint8)
    t_last = torch.randint(-1, T1, (B,), dtype=torch.int32)
    t_last[0] = T1 - 1; t_last[1] = -1
    live = (torch.arange(T1)[None, :] <= t_last[:, None]).reshape(R)
    masks = [_masks(c, obs, em, em0, gb, na).reshape(R, na, ne) for c in variants]
    qz = q * (1 - em[:, :, :na].reshape(R, na, 1).float())            # dead rows enter as zeros
    kvz = kv * (1 - em.reshape(R, ne, 1).float())
    outs = _attn_ref(qz, kvz[:, :, :w], kvz[:, :, w:], masks, heads)
    dO = torch.randn(3, R, na, w)
    sum((o * dO[i])[live].sum() for i, o in enumerate(outs)).backward()

    qd, kvd = q.detach().reshape(R * na, w).clone(), kv.detach().reshape(R * ne, 2 * w).clone()
    qd[em[:, :, :na].reshape(R * na).bool()] = float("nan")           # garbage where the producers skipped
    kvd[em.reshape(R * ne).bool()] = float("nan")
    qd, kvd = qd.to(DEV), kvd.to(DEV)
    d = hip_ops.attn_desc(qd, kvd, kvd[:, w:], w, 2 * w, R, T1, ne, na, heads, hd, variants, obs_mask=obs.to(DEV),
                          ent_mask=em.reshape(R, ne).to(DEV), ent_mask0=em0.to(DEV), group_bits=gb.to(DEV))
    hip_ops.attn_skip(d, t_last.to(DEV), em.reshape(R * ne).to(DEV), em[:, :, :na].reshape(R * na).contiguous().to(DEV))
    if pre:
        hip_ops.attn_mask_words(d, na)
    O = torch.full((3, R * na, w), 7.0, device=DEV)
    hip_ops.attn_forward(d, O, w, R * na * w)
    dQ = torch.full((R * na, w), 7.0, device=DEV)
    dKV = torch.full((R * ne, 2 * w), 7.0, device=DEV)
    hip_ops.attn_backward(d, dO.reshape(3, R * na, w).to(DEV), w, R * na * w, dQ, dKV, dKV[:, w:])
    lq = (live[:, None] & ~em[:, :, :na].reshape(R, na).bool())
    lk = (live[:, None] & ~em.reshape(R, ne).bool())
    O, dQ, dKV = O.cpu().reshape(3, R, na, w), dQ.cpu().reshape(R, na, w), dKV.cpu().reshape(R, ne, 2 * w)
    for i in range(3):
        _close(O[i][lq], outs[i].detach()[lq], what=f"walk fwd variant {i}")
    _close(dQ[lq], q.grad[lq], tol=5e-5, what="walk dQ")
    _close(dKV[lk], kv.grad[lk], tol=5e-5, what="walk dKV")
    assert (O[0][~live] == 7.0).all() and (dQ[~lq] == 7.0).all() and (dKV[~lk] == 7.0).all()     # untouched


# ------------------------------------------------------------------------------------------------
# in_trans + attention core in one launch (attention_qkv.hip)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T1,ne,na,heads,hd,nvar,store", [
    (24, 40, 32, 16, 4, 32, 3, True), (24, 40, 32, 16, 4, 32, 1, False), (40, 30, 16, 8, 4, 16, 3, True), (5, 7, 20, 7, 4, 32, 2, True),
    (3, 4, 12, 5, 2, 32, 1, True), (6, 5, 32, 16, 8, 16, 3, False), (4, 6, 9, 9, 4, 16, 1, True), (300, 3, 16, 8, 4, 32, 1, False),
    (2, 3, 27, 16, 2, 32, 3, True),
    # more than 32 entities / 16 agents (three key tiles, two agent tiles, 64-bit mask words; BASELINE configs[4] is 48 / 24):
    (12, 20, 48, 24, 4, 32, 3, True), (4, 6, 40, 12, 4, 32, 2, True), (5, 5, 32, 24, 4, 32, 3, True), (3, 4, 48, 32, 4, 16, 1, False),
    (6, 5, 30, 20, 4, 16, 3, True), (4, 5, 36, 9, 4, 16, 2, True), (40, 3, 48, 24, 4, 32, 1, False)])
def test_attention_qkv_forward(B, T1, ne, na, heads, hd, nvar, store):
    """refil_attn_qkv_forward (layer input + in_trans.weight -> attention output, projections optionally stored) against fp32 torch:
    ragged episode ends, dead K/V / Q rows holding NaN (never read), up to three mask variants, a strided layer input."""
    import hip_ops
    torch.manual_seed(B * 1000 + ne * 10 + hd + nvar)
    R, w = B * T1, heads * hd
    variants = [MASK_OBS, MASK_OBS_WITHIN, MASK_OBS_INTERACT][:nvar]
    x = torch.randn(R, ne, w)
    W = torch.randn(3 * w, w) / math.sqrt(w)
    obs = (torch.rand(B, T1, ne, ne) < 0.4).to(torch.uint8)
    em = torch.zeros(B, T1, ne, dtype=torch.uint8)
    if na > 2:
        em[:, :, na - 2:na] = 1
    if ne > na + 3:
        em[:, :, ne - 3:] = 1                                          # padded agents / enemies: dead Q and K/V rows
    em[1::3, T1 // 2:, 0] = 1                                        # an agent dying mid-episode
    obs = obs | em[:, :, :, None] | em[:, :, None, :]
    em0 = em[:, 0].contiguous()
    gb = (torch.rand(B, ne) < 0.5).to(torch.uint8)
    t_last = torch.randint(-1, T1, (B,), dtype=torch.int32)
    t_last[0] = T1 - 1
    if B > 1:
        t_last[1] = -1
    live = (torch.arange(T1)[None, :] <= t_last[:, None]).reshape(R)
    masks = [_masks(c, obs, em, em0, gb, na).reshape(R, na, ne) for c in variants]
    xz = x * (1 - em.reshape(R, ne, 1).float())                        # dead rows enter as zeros
    qr = (xz[:, :na] @ W[:w].t()) * (1 - em[:, :, :na].reshape(R, na, 1).float())
    kr, vr = xz @ W[w:2 * w].t(), xz @ W[2 * w:].t()
    outs = _attn_ref(qr, kr, vr, masks, heads)

    ldx = 2 * w + 8
    xd = torch.full((R * ne, ldx), float("nan"))
    xd[:, w:2 * w] = x.reshape(R * ne, w)
    xd[em.reshape(R * ne).bool()] = float("nan")                       # garbage where the producer skipped
    xd = xd.to(DEV)
    Wd = W.to(DEV)
    dummy = torch.zeros(4, device=DEV)
    d = hip_ops.attn_desc(dummy, dummy, dummy, w, 2 * w, R, T1, ne, na, heads, hd, variants, obs_mask=obs.to(DEV),
                          ent_mask=em.reshape(R, ne).to(DEV), ent_mask0=em0.to(DEV), group_bits=gb.to(DEV))
    hip_ops.attn_skip(d, t_last.to(DEV), em.reshape(R * ne).to(DEV), em[:, :, :na].reshape(R * na).contiguous().to(DEV))
    hip_ops.attn_mask_words(d, na)
    O = torch.full((nvar, R * na, w), 7.0, device=DEV)
    qo = torch.full((R * na, w), 7.0, device=DEV) if store else None
    kvo = torch.full((R * ne, 2 * w), 7.0, device=DEV) if store else None
    hip_ops.attn_qkv_forward(d, xd[:, w:], ldx, Wd, O, w, R * na * w, q_out=qo, k_out=kvo, v_out=kvo[:, w:] if store else None)
    lq = (live[:, None] & ~em[:, :, :na].reshape(R, na).bool())
    lk = (live[:, None] & ~em.reshape(R, ne).bool())
    O = O.cpu().reshape(nvar, R, na, w)
    for i in range(nvar):
        _close(O[i][lq], outs[i][lq], what=f"qkv fwd variant {i}")
    assert (O[0][~live] == 7.0).all()                                  # rows of finished episodes untouched
    if store:
        qo, kvo = qo.cpu().reshape(R, na, w), kvo.cpu().reshape(R, ne, 2 * w)
        _close(qo[lq], qr[lq], tol=3e-6, what="stored Q")
        _close(kvo[lk][:, :w], kr[lk], tol=3e-6, what="stored K")
        _close(kvo[lk][:, w:], vr[lk], tol=3e-6, what="stored V")
        assert (qo[~lq] == 7.0).all() and (kvo[~lk] == 7.0).all()      # dead rows are not written


@pytest.mark.parametrize("ne,na,heads,hd", [(12, 6, 4, 16), (32, 16, 4, 32), (48, 24, 4, 32)])
def test_attention_qkv_query_alive_but_dead_as_key(ne, na, heads, hd):
    """refil_attn_qkv_forward with an agent that row_bits marks alive as a QUERY and dead as a KEY (and that every mask excludes as a
    key): its x row is fetched all the same -- Q^T is projected from the registers K^T / V come from -- and its output row equals the
    reference's. (The learner's row lists never produce such a row; the separate refil_attn_forward takes Q from a buffer.)"""
    import hip_ops
    torch.manual_seed(ne + hd)
    B, T1, nvar = 3, 4, 2
    R, w = B * T1, heads * hd
    variants = [MASK_OBS, MASK_OBS_WITHIN]
    x = torch.randn(R, ne, w)
    W = torch.randn(3 * w, w) / math.sqrt(w)
    em = torch.zeros(B, T1, ne, dtype=torch.uint8)
    em[:, :, ne - 2:] = 1
    obs = (torch.rand(B, T1, ne, ne) < 0.3).to(torch.uint8)
    obs = obs | em[:, :, :, None] | em[:, :, None, :]
    odd = 1                                                      # the agent nobody observes (not even itself): dead as a key, alive as a query
    obs[:, :, :, odd] = 1
    em0 = em[:, 0].contiguous()
    gb = (torch.rand(B, ne) < 0.5).to(torch.uint8)
    kv_dead = em.clone()
    kv_dead[:, :, odd] = 1
    masks = [_masks(c, obs, em, em0, gb, na).reshape(R, na, ne) for c in variants]
    xz = x * (1 - em.reshape(R, ne, 1).float())
    qr = xz[:, :na] @ W[:w].t()
    kr, vr = xz @ W[w:2 * w].t(), xz @ W[2 * w:].t()
    outs = _attn_ref(qr, kr, vr, masks, heads)
    xd = x.reshape(R * ne, w).clone()
    xd[em.reshape(R * ne).bool()] = float("nan")
    xd = xd.to(DEV)
    dummy = torch.zeros(4, device=DEV)
    d = hip_ops.attn_desc(dummy, dummy, dummy, w, 2 * w, R, T1, ne, na, heads, hd, variants, obs_mask=obs.to(DEV),
                          ent_mask=em.reshape(R, ne).to(DEV), ent_mask0=em0.to(DEV), group_bits=gb.to(DEV))
    hip_ops.attn_skip(d, None, kv_dead.reshape(R * ne).to(DEV), em[:, :, :na].reshape(R * na).contiguous().to(DEV))
    hip_ops.attn_mask_words(d, na)
    O = torch.full((nvar, R * na, w), 7.0, device=DEV)
    hip_ops.attn_qkv_forward(d, xd, w, W.to(DEV), O, w, R * na * w)
    O = O.cpu().reshape(nvar, R, na, w)
    lq = ~em[:, :, :na].reshape(R, na).bool()
    for i in range(nvar):
        _close(O[i][lq], outs[i][lq], what=f"variant {i}")
        assert torch.isfinite(O[i]).all()


def _qkv_fuzz_cases(n, seed=606):
    import random
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        heads, hd = rnd.choice([(4, 32), (4, 32), (4, 16), (8, 16), (2, 32)])       # w = 128 or 64
        wide_ok = (hd, heads * hd) in ((32, 128), (16, 64))
        ne = rnd.randint(1, 48 if wide_ok else 32)
        na = rnd.randint(1, min(ne, 32 if wide_ok else 16))
        out.append((rnd.randint(1, 7), rnd.randint(1, 9), ne, na, heads, hd, rnd.randint(1, 3), rnd.random() < 0.5, rnd.choice([0.0, 0.15, 0.5, 0.85]),
                    9000 + i))
    return out


@pytest.mark.parametrize("B,T1,ne,na,heads,hd,nvar,store,dead,seed", _qkv_fuzz_cases(int(__import__("os").environ.get("REFIL_FUZZ_QKV_N", "12")),
                                                                                          int(__import__("os").environ.get("REFIL_FUZZ_QKV_SEED", "606"))))
def test_attention_qkv_random_shapes(B, T1, ne, na, heads, hd, nvar, store, dead, seed):
    """Fuzz of the fused in_trans + attention launch: any entity / agent count its instantiations take (one to three key tiles, one or two
    agent tiles), RANDOM dead entities at every density (so the key compaction sees every fill of its tiles, including rows with no live
    entity and rows where only agents / only non-agents live), ragged episode ends, 1-3 mask variants, stores on / off, against fp32 torch.
    Half the cases use the entity / within / interact masks, which leave keys VISIBLE that are dead at this step (the imagined groups are
    drawn on the first step's entity mask, entity_rnn_agent.py:94-101): a dead K / V row is a zero row that still counts in the softmax,
    also when the compaction leaves it outside the computed tiles (found by this fuzz, seed 606 case 7: the first version dropped them)."""
    import hip_ops
    torch.manual_seed(seed)
    R, w = B * T1, heads * hd
    groups = [[MASK_OBS, MASK_OBS_WITHIN, MASK_OBS_INTERACT], [MASK_ENTITY, MASK_WITHIN, MASK_INTERACT]]
    variants = groups[seed % 2][:nvar]
    x = torch.randn(R, ne, w)
    W = torch.randn(3 * w, w) / math.sqrt(w)
    em = (torch.rand(B, T1, ne) < dead).to(torch.uint8)
    if B > 1:
        em[1] = 1                                                      # an episode with every entity inactive
    obs = (torch.rand(B, T1, ne, ne) < 0.4).to(torch.uint8)
    obs = obs | em[:, :, :, None] | em[:, :, None, :]
    em0 = em[:, 0].contiguous()
    gb = (torch.rand(B, ne) < 0.5).to(torch.uint8)
    t_last = torch.randint(-1, T1, (B,), dtype=torch.int32)
    t_last[0] = T1 - 1
    live = (torch.arange(T1)[None, :] <= t_last[:, None]).reshape(R)
    masks = [_masks(c, obs, em, em0, gb, na).reshape(R, na, ne) for c in variants]
    xz = x * (1 - em.reshape(R, ne, 1).float())
    qr = (xz[:, :na] @ W[:w].t()) * (1 - em[:, :, :na].reshape(R, na, 1).float())
    kr, vr = xz @ W[w:2 * w].t(), xz @ W[2 * w:].t()
    outs = _attn_ref(qr, kr, vr, masks, heads)
    xd = x.reshape(R * ne, w).clone()
    xd[em.reshape(R * ne).bool()] = float("nan")
    xd, Wd = xd.to(DEV), W.to(DEV)
    dummy = torch.zeros(4, device=DEV)
    d = hip_ops.attn_desc(dummy, dummy, dummy, w, 2 * w, R, T1, ne, na, heads, hd, variants, obs_mask=obs.to(DEV),
                          ent_mask=em.reshape(R, ne).to(DEV), ent_mask0=em0.to(DEV), group_bits=gb.to(DEV))
    hip_ops.attn_skip(d, t_last.to(DEV), em.reshape(R * ne).to(DEV), em[:, :, :na].reshape(R * na).contiguous().to(DEV))
    hip_ops.attn_mask_words(d, na)
    O = torch.full((nvar, R * na, w), 7.0, device=DEV)
    qo = torch.full((R * na, w), 7.0, device=DEV) if store else None
    kvo = torch.full((R * ne, 2 * w), 7.0, device=DEV) if store else None
    hip_ops.attn_qkv_forward(d, xd, w, Wd, O, w, R * na * w, q_out=qo, k_out=kvo, v_out=kvo[:, w:] if store else None)
    lq = (live[:, None] & ~em[:, :, :na].reshape(R, na).bool())
    lk = (live[:, None] & ~em.reshape(R, ne).bool())
    O = O.cpu().reshape(nvar, R, na, w)
    for i in range(nvar):
        if lq.any():
            _close(O[i][lq], outs[i][lq], what=f"qkv fuzz variant {i}")
        dq = live[:, None] & em[:, :, :na].reshape(R, na).bool()
        assert (O[i][dq] == 0).all(), "rows of inactive agents of a live step read as exact zeros"
    assert (O[0][~live] == 7.0).all()
    if store:
        qo, kvo = qo.cpu().reshape(R, na, w), kvo.cpu().reshape(R, ne, 2 * w)
        if lq.any():
            _close(qo[lq], qr[lq], tol=3e-6, what="stored Q")
        if lk.any():
            _close(kvo[lk][:, :w], kr[lk], tol=3e-6, what="stored K")
            _close(kvo[lk][:, w:], vr[lk], tol=3e-6, what="stored V")
        assert (qo[~lq] == 7.0).all() and (kvo[~lk] == 7.0).all()
