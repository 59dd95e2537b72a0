"""GPU parity tests, per kernel: the HIP path (through the C-ABI) against the CPU oracle /
the golden vectors generated from the reference.  Run with `-m gpu` on an MI355X."""
import os

import numpy as np
import pytest
import torch

import self_forcing_amd as sfa
from self_forcing_amd import ops
from oracle import wan_oracle as wo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


@pytest.fixture(scope="module")
def opsgold():
    return np.load(os.path.join(GOLD, "ops.npz"))


# ---------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (72, 64, 64), (200, 256, 128), (257, 384, 1536), (1560, 1536, 512),
                                   (4680, 512, 1024)])
@pytest.mark.parametrize("epi", ["bias", "gelu", "resid", "gate_resid"])
def test_gemm(M, N, K, epi):
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a, w = bf((M, K), g), bf((N, K), g, 1.0 / K ** 0.5)
    bias = bf((N,), g, 0.5)
    resid = bf((M, N), g)
    groups = 3 if M % 3 == 0 else 1
    gate_mod, e0 = bf((N,), g, 0.5), bf((groups, 6, N), g, 0.5)
    y = a.float() @ w.float().t() + bias.float()
    if epi == "gelu":
        ref = torch.nn.functional.gelu(y, approximate="tanh")
    elif epi == "resid":
        ref = resid.float() + y
    elif epi == "gate_resid":
        gate = (gate_mod.float()[None] + e0[:, 2].float()).to(torch.bfloat16).float()   # [groups, N]
        ref = resid.float() + y * gate.repeat_interleave(M // groups, dim=0)
    else:
        ref = y
    kw = {}
    if epi in ("resid", "gate_resid"):
        kw["resid"] = resid.to(DEV)
    if epi == "gate_resid":
        kw.update(gate_mod=gate_mod.to(DEV), gate_e0=e0.to(DEV)[:, 2], rows_per_group=M // groups)
    out = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), epilogue=epi, **kw)
    # one bf16 rounding of an fp32-accumulated result: <= 2^-9 relative per element
    assert rel(out, ref) < 4e-3
    assert (out.float().cpu() - ref).abs().max() <= 2e-2 * ref.abs().max()


@pytest.mark.parametrize("M,N,K,epi", [(4680, 1536, 1536, "gate_resid"), (4100, 2048, 128, "gelu"), (300, 192, 192, "bias"),
                                       (5000, 1664, 640, "resid")])
def test_gemm_large_ragged(M, N, K, epi):
    """Full-width problems: ragged M, N not a multiple of the tile, K of 2..24 steps, grouped tile
    order with a partial last row group."""
    g = torch.Generator().manual_seed(M + N + K)
    a, w, bias, resid = bf((M, K), g), bf((N, K), g, 1.0 / K ** 0.5), bf((N,), g, 0.5), bf((M, N), g)
    groups = 3 if M % 3 == 0 else 1
    gate_mod, e0 = bf((N,), g, 0.5), bf((groups, 6, N), g, 0.5)
    y = a.float() @ w.float().t() + bias.float()
    kw = {}
    if epi == "gelu":
        ref = torch.nn.functional.gelu(y, approximate="tanh")
    elif epi == "resid":
        ref, kw = resid.float() + y, {"resid": resid.to(DEV)}
    elif epi == "gate_resid":
        gate = (gate_mod.float()[None] + e0[:, 5].float()).to(torch.bfloat16).float()
        ref = resid.float() + y * gate.repeat_interleave(M // groups, dim=0)
        kw = dict(resid=resid.to(DEV), gate_mod=gate_mod.to(DEV), gate_e0=e0.to(DEV)[:, 5], rows_per_group=M // groups)
    else:
        ref = y
    out = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), epilogue=epi, **kw)
    assert rel(out, ref) < 4e-3


def test_gemm_inplace_residual_and_strides():
    g = torch.Generator().manual_seed(3)
    M, N, K = 300, 256, 192
    a_full = bf((M, K + 64), g).to(DEV)
    a = a_full[:, :K]                      # row stride K+64
    w, x = bf((N, K), g, 0.1).to(DEV), bf((M, N), g).to(DEV)
    ref = x.float() + a.float() @ w.float().t()
    ops.gemm(a, w, None, epilogue="resid", resid=x, out=x)   # out aliases resid
    assert rel(x, ref) < 4e-3


def test_gemm_rejects_bad_shapes():
    a = torch.zeros(8, 100, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(16, 100, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(sfa._lib.SfHipError):
        ops.gemm(a, w)          # K not a multiple of 64
    with pytest.raises(ValueError):
        ops.gemm(a.float(), w)  # wrong dtype


@pytest.mark.parametrize("epi", ["bias", "gelu", "gate_resid"])
def test_gemm_large_tile_structure(epi):
    """Wide outputs whose 256 x 256 tiles fill the chip dispatch the 8-wave ping-pong kernel (ffn.0's shape, 3 k-tiles
    here: its prologue / tail paths); ragged M (4680 = 18 x 256 + 72) exercises its row clamping."""
    M, N, K = 4680, 8960, 192
    g = torch.Generator().manual_seed(77)
    a, w, bias = bf((M, K), g), bf((N, K), g, 1.0 / K ** 0.5), bf((N,), g, 0.5)
    y = a.float() @ w.float().t() + bias.float()
    kw = {}
    if epi == "gelu":
        ref = torch.nn.functional.gelu(y, approximate="tanh")
    elif epi == "gate_resid":
        resid, gate_mod, e0 = bf((M, N), g), bf((N,), g, 0.5), bf((3, 6, N), g, 0.5)
        gate = (gate_mod.float()[None] + e0[:, 2].float()).to(torch.bfloat16).float()
        ref = resid.float() + y * gate.repeat_interleave(M // 3, dim=0)
        kw = dict(resid=resid.to(DEV), gate_mod=gate_mod.to(DEV), gate_e0=e0[:, 2].to(DEV), rows_per_group=M // 3)
    else:
        ref = y
    out = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), epilogue=epi, **kw)
    assert rel(out, ref) < 4e-3


@pytest.mark.parametrize("structure", ["t128", "pp256", "pp224", "pp192", "pp128"])
@pytest.mark.parametrize("M,N,K,epi", [(4680, 1536, 1536, "gate_resid"), (4680, 4608, 1536, "bias"), (1560, 8960, 1536, "gelu"),
                                       (4680, 1536, 8960, "gate_resid"), (10800, 5120, 5120, "resid"), (1100, 1288, 128, "gelu"),
                                       (257, 264, 192, "resid"), (3000, 1024, 64, "bias"), (9360, 8960, 512, "gelu"), (9360, 4608, 1536, "bias")])
def test_gemm_every_structure(structure, M, N, K, epi):
    """Every tiling on the rollout's own GEMM shapes (qkv, the N = 1536 projections, ffn.0, ffn.2 with its 140 k-tiles,
    a 14B / 720p projection) and on ragged ones: rows not a multiple of 16, N not a multiple of the 256 / 128 tile,
    2 and 3 k-tiles (the ping-pong structure's prologue and tail), K = 64 (falls back to the 128 x 128 kernel).
    All structures must also agree with each other to the last bit: same MFMA, same k order, same epilogue."""
    g = torch.Generator().manual_seed(M + N + K)
    a, w, bias, resid = bf((M, K), g), bf((N, K), g, 1.0 / K ** 0.5), bf((N,), g, 0.5), bf((M, N), g)
    groups = 3 if M % 3 == 0 else 1
    gate_mod, e0 = bf((N,), g, 0.5), bf((groups, 6, N), g, 0.5)
    y = a.float() @ w.float().t() + bias.float()
    kw = {}
    if epi == "gelu":
        ref = torch.nn.functional.gelu(y, approximate="tanh")
    elif epi == "resid":
        ref, kw = resid.float() + y, {"resid": resid.to(DEV)}
    elif epi == "gate_resid":
        gate = (gate_mod.float()[None] + e0[:, 5].float()).to(torch.bfloat16).float()
        ref = resid.float() + y * gate.repeat_interleave(M // groups, dim=0)
        kw = dict(resid=resid.to(DEV), gate_mod=gate_mod.to(DEV), gate_e0=e0.to(DEV)[:, 5], rows_per_group=M // groups)
    else:
        ref = y
    ad, wd, bd = a.to(DEV), w.to(DEV), bias.to(DEV)
    out = ops.gemm(ad, wd, bd, epilogue=epi, structure=structure, **kw)
    assert rel(out, ref) < 4e-3
    assert torch.equal(out, ops.gemm(ad, wd, bd, epilogue=epi, structure="t128", **kw))
    assert torch.equal(out, ops.gemm(ad, wd, bd, epilogue=epi, **kw))          # the automatic choice


def test_gemm_pingpong_is_race_free_under_repetition():
    """The ping-pong structure keeps LDS-DMA requests in flight across barriers (counted vmcnt): an early fragment
    read or a late buffer reuse would show as rare wrong tiles that come and go with timing.  200 launches of two
    shapes (many k-tiles; few k-tiles + many workgroup rounds), beside a second stream that keeps the memory system
    busy, every result compared bit for bit with the first."""
    g = torch.Generator().manual_seed(5)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.bfloat16, device=DEV)
    for (M, N, K, st) in [(4680, 1536, 8960, "pp128"), (4680, 8960, 1536, "pp256"), (4680, 4608, 1536, "pp128"), (2048, 2048, 2048, "pp256"),
                          (9360, 8960, 1536, "pp224"), (9360, 1536, 8960, "pp224"), (4680, 4608, 1536, "pp192"), (9360, 4608, 1536, "pp224")]:
        a, w = bf((M, K), g).to(DEV), bf((N, K), g, 1.0 / K ** 0.5).to(DEV)
        first = ops.gemm(a, w, None, structure=st)
        assert rel(first, a.float().cpu() @ w.float().cpu().t()) < 4e-3
        bad = 0
        for i in range(50):
            if i % 5 == 0:
                with torch.cuda.stream(side):
                    junk.add_(1)
            bad += int(not torch.equal(ops.gemm(a, w, None, structure=st), first))
        torch.cuda.synchronize()
        assert bad == 0, (M, N, K, st, bad)


# ----------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 1, 32, 64), (1, 2, 100, 200), (2, 3, 130, 24), (1, 4, 24, 512),
                                        (1, 2, 1560, 4680), (1, 1, 200, 1561)])
def test_attention(B, H, Lq, Lk):
    g = torch.Generator().manual_seed(Lq + Lk)
    q, k, v = bf((B, Lq, H, 128), g), bf((B, Lk, H, 128), g), bf((B, Lk, H, 128), g)
    ref = wo.sdpa(q.float(), k.float(), v.float())
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    # P is rounded to bf16 before P.V (as flash-attn does) and the output once more
    assert rel(out, ref) < 6e-3


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 12, 4200, 300), (2, 8, 3100, 130), (1, 12, 4680, 1561), (1, 16, 3073, 64)])
def test_attention_w8_structure(B, H, Lq, Lk):
    """Shapes that fill the chip dispatch the 8-wave staggered kernel (256 query rows / workgroup)."""
    g = torch.Generator().manual_seed(Lq + Lk)
    q, k, v = bf((B, Lq, H, 128), g), bf((B, Lk, H, 128), g), bf((B, Lk, H, 128), g)
    ref = wo.sdpa(q.float(), k.float(), v.float())
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    assert rel(out, ref) < 6e-3


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 12, 4200, 1100), (2, 8, 3100, 1300), (1, 12, 4680, 4680), (1, 16, 3073, 2049)])
def test_attention_r64_structure(B, H, Lq, Lk):
    """Long key sequences that fill the chip dispatch the hand-scheduled 64-rows-per-wave kernel."""
    g = torch.Generator().manual_seed(Lq + Lk)
    q, k, v = bf((B, Lq, H, 128), g), bf((B, Lk, H, 128), g), bf((B, Lk, H, 128), g)
    ref = wo.sdpa(q.float(), k.float(), v.float())
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    assert rel(out, ref) < 6e-3


def test_attention_r64_lazy_rescale_and_stale_lds():
    """The 64-row kernel moves its softmax reference only when a row's maximum exceeds it by 2^8: feed it
    scores that (a) stay inside the threshold, (b) jump far beyond it late in the sequence, (c) decay, and
    rows whose large early maximum makes every later probability underflow; NaN left in LDS by a previous
    launch must not leak into rows past Lk."""
    g = torch.Generator().manual_seed(21)
    B, H, Lq, Lk = 1, 2, 200, 1000
    q, k, v = bf((B, Lq, H, 128), g), bf((B, Lk, H, 128), g, 0.2), bf((B, Lk, H, 128), g)
    k[0, 900, 0] = (q[0, 5, 0].float() * 4).to(torch.bfloat16)      # +hundreds of log2 units at tile 14
    k[0, 2, 1] = (q[0, 9, 1].float() * 4).to(torch.bfloat16)        # dominant key in the first tile
    k[0, 500, 0] = (q[0, 70, 0].float() * 0.6).to(torch.bfloat16)   # moderate bump (inside / near the threshold)
    ref = wo.sdpa(q.float(), k.float(), v.float())
    nan = torch.full((1, 64, 2, 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.attention(nan, nan, nan, structure="r64")                    # leaves NaN bit patterns in LDS
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), structure="r64")
    assert torch.isfinite(out.float()).all()
    assert rel(out, ref) < 6e-3
    assert (out.float().cpu() - ref).abs().max() < 4e-2


@pytest.mark.parametrize("force", ["w8", "w4", "r64"])
def test_attention_both_structures_small_and_spiky(force):
    """All kernels on the same ragged inputs incl. a late max spike (rescale branch) and Lk = 1."""
    g = torch.Generator().manual_seed(5)
    for (B, H, Lq, Lk) in [(1, 2, 300, 448), (2, 1, 33, 65), (1, 3, 257, 1), (1, 1, 512, 129)]:
        q, k, v = bf((B, Lq, H, 128), g), bf((B, Lk, H, 128), g, 0.3), bf((B, Lk, H, 128), g)
        if Lk > 400:
            k[0, 400, 0] = (q[0, 5, 0].float() * 3).to(torch.bfloat16)
        ref = wo.sdpa(q.float(), k.float(), v.float())
        out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), structure=force)
        assert rel(out, ref) < 6e-3, (B, H, Lq, Lk)


def _row_subset(Lq, step):
    """Strided query rows + the whole ragged last 64-row block + the first block."""
    rows = sorted(set(range(0, Lq, step)) | set(range(max(0, Lq - 70), Lq)) | set(range(0, 66)))
    return torch.tensor(rows)


@pytest.mark.parametrize("H,Lq,Lk,kind", [
    (12, 4680, 32760, "random"),      # the benchmark's last chunk: 512 key tiles, K ring of 4 / V ring of 3 wrap > 120 times
    (12, 4680, 32760, "drift"),       # scores creep upwards along the sequence + late spikes: lazy rescale fires over and over
    (12, 4680, 65520, "random"),      # the long-context configuration (42 latent frames)
    (12, 4680, 65520, "drift"),
    (12, 4680, 18720, "drift"),       # the mean cache length of a rollout (roofline leg)
    (40, 10800, 10800, "random"),     # Wan-14B / 720p: 40 heads, 3 x 3600 tokens, first chunk
    (40, 10800, 21600, "drift"),
])
def test_attention_r64_long_sequences_vs_fp32(H, Lq, Lk, kind):
    """The hand-scheduled 64-row kernel at the lengths where the benchmark spends its FLOPs (Lk 14040 ... 32760), at the
    long-context length 65520 and at the 14B / 720p shape, against fp32 softmax(QK^T)V on the CPU for a strided subset
    of query rows of EVERY head that includes the first block and the ragged last one (4680 = 18 x 256 + 72,
    10800 = 42 x 256 + 48; Lk 32760 / 65520 / 10800 are not multiples of the 64-key tile either).  Same bounds as
    the short-sequence tests: relative Frobenius 6e-3, and a max-abs bound scaled to the output's rms."""
    g = torch.Generator().manual_seed(H + Lq + Lk)
    q, k, v = bf((1, Lq, H, 128), g), bf((1, Lk, H, 128), g, 0.5), bf((1, Lk, H, 128), g)
    if kind == "drift":
        # key norms grow 4x along the sequence (row maxima keep rising: the reference point of the lazy rescale is moved
        # again and again, hundreds of tiles apart), plus keys aligned with single queries far along the sequence
        ramp = torch.linspace(0.5, 2.0, Lk).view(1, Lk, 1, 1)
        k = (k.float() * ramp).to(torch.bfloat16)
        for j, pos in enumerate(range(Lk // 3, Lk - 5, Lk // 9)):
            k[0, pos, j % H] = (q[0, (977 * j + 13) % Lq, j % H].float() * 4).to(torch.bfloat16)
            k[0, pos + 3, (j + 1) % H] = (q[0, Lq - 1 - j, (j + 1) % H].float() * 4).to(torch.bfloat16)   # rows of the ragged tail
    rows = _row_subset(Lq, 61)
    ref = wo.sdpa(q[:, rows].float(), k.float(), v.float())
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), structure="r64")
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    got = out[:, rows.to(DEV)].float().cpu()
    assert rel(got, ref) < 6e-3, rel(got, ref)
    # element-wise: bf16 rounding of the output (2^-9 relative: rows dominated by one aligned key hold O(1) values) plus
    # a floor scaled to the output's rms (long sequences average thousands of V rows: rms ~ 0.05 ... 0.1)
    rms = ref.pow(2).mean().sqrt().item()
    assert ((got - ref).abs() <= 1.2e-2 * ref.abs() + 0.05 * rms).all(), ((got - ref).abs().max().item(), rms)
    # per head as well: one bad head among 40 would hide in the overall norm
    for h in range(H):
        assert rel(got[:, :, h], ref[:, :, h]) < 8e-3, h
    # the automatic dispatch picks this structure for these shapes and gives the same bits
    assert torch.equal(out, ops.attention(q.to(DEV), k.to(DEV), v.to(DEV)))


def test_attention_online_softmax_rescale_branch():
    """Force the running max to jump late: one key far along the sequence dominates one query
    (cdna guide rule 26: a rescale branch needs an input that takes it)."""
    g = torch.Generator().manual_seed(11)
    B, H, Lq, Lk = 1, 1, 64, 448
    q, k, v = bf((B, Lq, H, 128), g), bf((B, Lk, H, 128), g, 0.2), bf((B, Lk, H, 128), g)
    k[0, 400, 0] = (q[0, 5, 0].float() * 3).to(torch.bfloat16)     # spike in the 7th tile
    k[0, 3, 0] = (q[0, 9, 0].float() * 2).to(torch.bfloat16)       # and one in the first
    ref = wo.sdpa(q.float(), k.float(), v.float())
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    assert rel(out, ref) < 6e-3
    assert (out.float().cpu() - ref).abs().max() < 3e-2


def test_attention_strided_cache_window():
    """K/V as a window [start:end) of a larger cache, Q/K/V token-major with all heads interleaved
    (the layout of the KV cache, pipeline/causal_inference.py:288-293)."""
    g = torch.Generator().manual_seed(12)
    B, H, S = 2, 3, 300
    kc, vc = bf((B, S, H, 128), g).to(DEV), bf((B, S, H, 128), g).to(DEV)
    q = bf((B, 70, H, 128), g).to(DEV)
    out = ops.attention(q, kc[:, 37:250], vc[:, 37:250])
    ref = wo.sdpa(q.float().cpu(), kc[:, 37:250].float().cpu(), vc[:, 37:250].float().cpu())
    assert rel(out, ref) < 6e-3


# ------------------------------------------------------------------------- norms / rope / cache
@pytest.mark.parametrize("C", [512, 1536, 5120])
def test_layernorm_modulate_and_affine(C):
    g = torch.Generator().manual_seed(C)
    M, G = 96, 3
    x = bf((M, C), g, 2.0)
    mod, e0 = bf((6, C), g, 0.3), bf((G, 6, C), g, 0.3)
    e = (mod.float()[None] + e0.float()).to(torch.bfloat16).float()           # [G, 6, C]
    ln = wo.layer_norm(x.float(), 1e-6)
    ref = ln.view(G, M // G, C) * (1 + e[:, 1:2]) + e[:, 0:1]
    e0d = e0.to(DEV)
    out = ops.layernorm_modulate(x.to(DEV), mod[0].to(DEV), mod[1].to(DEV), e0d[:, 0], e0d[:, 1], M // G)
    assert rel(out, ref.reshape(M, C)) < 3e-3
    w, b = bf((C,), g), bf((C,), g)
    out2 = ops.layernorm_affine(x.to(DEV), w.to(DEV), b.to(DEV))
    assert rel(out2, wo.layer_norm(x.float(), 1e-6, w.float(), b.float())) < 3e-3


def test_rmsnorm_golden(opsgold):
    x, w = T(opsgold["rms_x"]).bfloat16(), T(opsgold["rms_w"]).bfloat16()
    out = ops.rmsnorm(x.to(DEV), w.to(DEV))
    # same rounding points as the reference -> equal up to 1 bf16 ulp on a few elements
    ref = T(opsgold["rms_out_bf16"])
    assert rel(out, ref) < 2e-3
    assert (out.float().cpu() != ref).float().mean() < 0.02


def test_qkv_norm_rope_cache_vs_oracle():
    g = torch.Generator().manual_seed(21)
    B, f, h, w, H = 2, 2, 4, 6, 4
    C, L, S = H * 128, f * h * w, 100
    qkv = bf((B * L, 3 * C), g)
    wq, wk = (1 + 0.1 * torch.randn(C, generator=g)).to(torch.bfloat16), (1 + 0.1 * torch.randn(C, generator=g)).to(torch.bfloat16)
    kc = torch.zeros(B, S, H, 128, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    cos, sin = sfa.model.rope_tables(128)
    start_frame, write_start = 5, 31
    q = ops.qkv_norm_rope_cache(qkv.to(DEV), wq.to(DEV), wk.to(DEV), kc, vc, cos.to(DEV), sin.to(DEV), (f, h, w),
                                write_start, start_frame)
    rc, rs, split = wo.rope_tables(128)
    x = qkv.view(B, L, 3, C)
    qr = wo.causal_rope_apply(wo.rms_norm(x[:, :, 0], wq, 1e-6).view(B, L, H, 128), (f, h, w), rc, rs, split, start_frame)
    kr = wo.causal_rope_apply(wo.rms_norm(x[:, :, 1], wk, 1e-6).view(B, L, H, 128), (f, h, w), rc, rs, split, start_frame)
    # bf16 oracle mode has the same rounding points; allow 1-ulp flips from fp32-vs-fp64 rotation
    assert rel(q.view(B, L, H, 128), qr.float()) < 2e-3
    assert (q.view(B, L, H, 128).cpu() != qr).float().mean() < 0.02
    assert rel(kc[:, write_start:write_start + L], kr.float()) < 2e-3
    assert torch.equal(vc[:, write_start:write_start + L].cpu(), x[:, :, 2].reshape(B, L, H, 128))
    assert kc[:, :write_start].abs().sum() == 0 and kc[:, write_start + L:].abs().sum() == 0


def test_qkv_norm_rope_cache_overflow_raises():
    C = 512
    qkv = torch.zeros(24, 3 * C, dtype=torch.bfloat16, device=DEV)
    w = torch.ones(C, dtype=torch.bfloat16, device=DEV)
    kc = torch.zeros(1, 30, 4, 128, dtype=torch.bfloat16, device=DEV)
    cos, sin = sfa.model.rope_tables(128)
    with pytest.raises(sfa._lib.SfHipError, match="overflow"):
        ops.qkv_norm_rope_cache(qkv, w, w, kc, kc.clone(), cos.to(DEV), sin.to(DEV), (1, 4, 6), 10, 0)


def test_kv_evict():
    g = torch.Generator().manual_seed(4)
    B, S, H = 2, 40, 4
    cache = bf((B, S, H, 128), g).to(DEV)
    ref = cache.clone()
    sink, evict, keep = 8, 6, 20
    ref[:, sink:sink + keep] = cache[:, sink + evict:sink + evict + keep].clone()
    scratch = torch.empty(B * keep * H * 128 * 2, dtype=torch.uint8, device=DEV)
    ops.kv_evict(cache, sink, evict, keep, scratch)
    assert torch.equal(cache, ref)


# ----------------------------------------------------------------------- small kernels, golden
def test_sinusoid_golden(opsgold):
    t = T(opsgold["sinus_t"], torch.float64).float()
    out = ops.sinusoid_embedding(t.to(DEV), 256)
    ref = T(opsgold["sinus_out"], torch.float64).to(torch.bfloat16)
    # device fp64 cos/sin/pow vs the host's: equal except rare last-bit ties
    assert (out.cpu() != ref).float().mean() < 0.005
    assert (out.float().cpu() - ref.float()).abs().max() < 1e-2
    ti = torch.tensor([0, 250, 1000], dtype=torch.int64)
    outi = ops.sinusoid_embedding(ti.to(DEV), 256)
    refi = wo.sinusoidal_embedding_1d(256, ti).to(torch.bfloat16)
    assert (outi.cpu() != refi).float().mean() < 0.005


def test_small_linear():
    g = torch.Generator().manual_seed(8)
    for M, N, K, ai, ao in [(3, 1536, 256, None, "silu"), (1, 512, 512, None, None), (21, 9216 // 4, 1536, "silu", None),
                            (7, 130, 64, "silu", "gelu")]:
        x, w, b = bf((M, K), g), bf((N, K), g, 0.05), bf((N,), g, 0.1)
        xin = torch.nn.functional.silu(x.float()).to(torch.bfloat16).float() if ai == "silu" else x.float()
        y = xin @ w.float().t() + b.float()
        if ao == "silu":
            y = torch.nn.functional.silu(y)
        elif ao == "gelu":
            y = torch.nn.functional.gelu(y, approximate="tanh")
        out = ops.small_linear(x.to(DEV), w.to(DEV), b.to(DEV), ai, ao)
        assert rel(out, y) < 4e-3, (M, N, K)


def test_patchify_gemm_matches_conv_golden():
    mods = np.load(os.path.join(GOLD, "modules_reduced.npz"))
    sd = sfa.synth_state_dict(sfa.WAN_REDUCED, seed=0)
    x = T(mods["pe_x"]).bfloat16()                       # [1, 16, 2, 8, 12] (model layout: C before F)
    cols = ops.patchify(x.permute(0, 2, 1, 3, 4).contiguous().to(DEV))
    tok = ops.gemm(cols, sd["patch_embedding.weight"].flatten(1).to(DEV), sd["patch_embedding.bias"].to(DEV))
    assert rel(tok, T(mods["pe_out_f32"])[0]) < 4e-3


def test_unpatchify_x0_golden(opsgold):
    mods = np.load(os.path.join(GOLD, "modules_reduced.npz"))
    sched = wo.FlowMatchTables(5.0)
    hx = T(mods["unp_x"]).bfloat16()                     # [2*24, 64]
    g = torch.Generator().manual_seed(2)
    xt = bf((1, 2, 16, 8, 12), g)
    ts = torch.tensor([[937.5, 625.0]], dtype=torch.float32)
    flow, x0 = ops.unpatchify_x0(hx.to(DEV), xt.to(DEV), ts.to(DEV), sched.sigmas.to(DEV), sched.timesteps.to(DEV))
    ref_flow = T(mods["unp_out_bf16"]).permute(1, 0, 2, 3)[None]          # [16,2,8,12] -> [1,2,16,8,12]
    assert torch.equal(flow.float().cpu(), ref_flow)
    ref_x0 = wo.flow_to_x0(sched, ref_flow[0].bfloat16(), xt[0], ts[0])
    assert torch.equal(x0[0].cpu(), ref_x0)              # fp64 math, one rounding: bit exact
    # one modulation group for both frames: a single sigma is broadcast
    flow1, x01 = ops.unpatchify_x0(hx.to(DEV), xt.to(DEV), ts[:, :1].to(DEV), sched.sigmas.to(DEV), sched.timesteps.to(DEV))
    ref1 = wo.flow_to_x0(sched, ref_flow[0].bfloat16(), xt[0], ts[0, :1].repeat(2))
    assert torch.equal(x01[0].cpu(), ref1)


def test_add_noise_golden(opsgold):
    sched = wo.FlowMatchTables(5.0)
    x0, eps, t = T(opsgold["an_x0"]).bfloat16(), T(opsgold["an_eps"]).bfloat16(), T(opsgold["an_t"])
    out = ops.add_noise(x0.to(DEV), eps.to(DEV), t.to(DEV), sched.sigmas.to(DEV), sched.timesteps.to(DEV))
    assert torch.equal(out.float().cpu(), T(opsgold["an_out_bf16"]))
    ti = torch.from_numpy(opsgold["x0_ti"])
    outi = ops.add_noise(x0.to(DEV), eps.to(DEV), ti.to(DEV), sched.sigmas.to(DEV), sched.timesteps.to(DEV))
    assert torch.equal(outi.float().cpu(), T(opsgold["an_out_int_bf16"]))


# ------------------------------------------------------------------------- torch custom ops (row b')
def test_torch_ops_bit_identical_to_the_ctypes_path():
    """torch.ops.sf_hip.* against direct calls into the C-ABI with raw pointers: same bits (the custom-op layer adds
    validation, allocation and the stream lookup, no arithmetic)."""
    import ctypes as C
    from self_forcing_amd import _lib
    g = torch.Generator().manual_seed(77)
    q, k, v = bf((1, 300, 2, 128), g).to(DEV), bf((1, 1000, 2, 128), g).to(DEV), bf((1, 1000, 2, 128), g).to(DEV)
    stream = torch.cuda.current_stream().cuda_stream
    direct = torch.empty_like(q)
    _lib.check(_lib.lib().sf_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), direct.data_ptr(), 1, 2, 300, 1000, 256, 300 * 256,
                                       256, 1000 * 256, 256, 300 * 256, stream))
    assert torch.equal(torch.ops.sf_hip.attention(q, k, v), direct)
    assert torch.equal(ops.attention(q, k, v), direct)

    a, w, b = bf((700, 512), g).to(DEV), bf((384, 512), g, 0.05).to(DEV), bf((384,), g).to(DEV)
    ga = _lib.GemmArgs()
    d2 = torch.empty(700, 384, dtype=torch.bfloat16, device=DEV)
    ga.a, ga.w, ga.bias, ga.out = a.data_ptr(), w.data_ptr(), b.data_ptr(), d2.data_ptr()
    ga.M, ga.N, ga.K, ga.lda, ga.ldw, ga.ldo, ga.epilogue, ga.rows_per_group = 700, 384, 512, 512, 512, 384, _lib.EPI_BIAS_GELU, 1
    _lib.check(_lib.lib().sf_gemm_bf16(ga, stream))
    assert torch.equal(torch.ops.sf_hip.gemm(a, w, b, _lib.EPI_BIAS_GELU, None, None, None, 1, 0), d2)
    out = torch.zeros_like(d2)
    torch.ops.sf_hip.gemm_out(out, a, w, b, _lib.EPI_BIAS_GELU, None, None, None, 1, 0)
    assert torch.equal(out, d2)

    xs = [bf((3, 16, 8, 12), g).to(DEV) for _ in range(3)]
    cf = [0.25, -1.5, 3.0]
    d3 = torch.empty_like(xs[0])
    ptrs = (C.c_void_p * 3)(*[t.data_ptr() for t in xs])
    _lib.check(_lib.lib().sf_lincomb_bf16(d3.data_ptr(), ptrs, (C.c_float * 3)(*cf), 3, xs[0].numel(), stream))
    assert torch.equal(torch.ops.sf_hip.lincomb(xs, cf), d3)
    torch.cuda.synchronize()


def test_torch_ops_trace_and_compile_without_graph_surprises():
    """The reference's demo wraps the generator in torch.compile (demo.py:340).  Through the custom ops a function over
    the HIP kernels traces (fake tensors, declared mutation) and the compiled function returns the eager bits.
    backend='aot_eager': functionalisation + the ops' fake kernels, no code generation (there is no Triton here)."""
    g = torch.Generator().manual_seed(78)
    q, k, v = bf((1, 200, 2, 128), g).to(DEV), bf((1, 600, 2, 128), g).to(DEV), bf((1, 600, 2, 128), g).to(DEV)
    w, b = bf((256, 256), g, 0.05).to(DEV), bf((256,), g).to(DEV)

    def block(q, k, v, w, b):
        o = torch.ops.sf_hip.attention(q, k, v)
        y = torch.ops.sf_hip.gemm(o.reshape(200, 256), w, b, 0, None, None, None, 1, 0)
        acc = torch.zeros_like(y)
        torch.ops.sf_hip.lincomb_out(acc, [y, y], [0.5, 0.25])
        return acc

    eager = block(q, k, v, w, b)
    compiled = torch.compile(block, backend="aot_eager", fullgraph=True)(q, k, v, w, b)
    torch.cuda.synchronize()
    assert torch.equal(eager, compiled)
