"""GPU parity of the dense part of the path (pre-processing, Swin-T, FPN, RPN convs, semantic head) against
the oracle and the committed reference goldens, through the C ABI (libnuhtc_hip.so)."""
import numpy as np
import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu

# fp32 tolerance of this tier: |err| <= ATOL + RTOL*|ref| per element (accumulation order differs between the
# MFMA k-ordered fma chain and the CPU's blocked sums); decisions downstream are compared by IoU, not by value.
RTOL, ATOL = 2e-4, 2e-4


def _engine(g, max_batch=None):
    from nuhtc_amd.engine import Engine
    sd = G.seeded_sd(g)
    tiles = g['tiles']
    return Engine(sd, device=0, max_batch=max_batch or len(tiles), tile=tiles.shape[1:3]), sd


def _cmp(name, got, ref, errs, rtol=RTOL, atol=ATOL):
    got = got.detach().cpu().float().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().float().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    if got.shape != ref.shape:
        errs.append(f'{name}: shape {got.shape} vs {ref.shape}')
        return
    err = np.abs(got - ref)
    bad = err > atol + rtol * np.abs(ref)
    msg = f'{name}: max_abs_err={err.max():.3e} ref_absmax={np.abs(ref).max():.3e} bad={int(bad.sum())}/{bad.size}'
    print(msg)
    if bad.any() or not np.isfinite(got).all():
        errs.append(msg)


def test_gemm_op_shapes(hip_device):
    from nuhtc_amd.engine import Engine
    from nuhtc_amd import weights
    g = G.load('small_b2')
    eng, _ = _engine(g)
    errs = []
    gen = torch.Generator().manual_seed(0)
    for (M, N, K) in [(128, 96, 96), (200, 288, 96), (1, 32, 32), (49, 64, 64), (333, 128, 192), (130, 384, 96), (257, 96, 384),
                      (64, 256, 3136), (500, 2304, 768), (77, 768, 3072)]:
        A = torch.randn(M, K, generator=gen)
        W = torch.randn(N, K, generator=gen) / K ** 0.5
        b = torch.randn(N, generator=gen)
        for act in (0, 1, 2):
            ref = A.double() @ W.double().T + b.double()
            ref = ref if act == 0 else (torch.relu(ref) if act == 1 else torch.nn.functional.gelu(ref))
            for pipe in ('fp32', 'split'):
                out = eng.op_gemm(A.cuda(), W.cuda(), b.cuda(), act, pipe=pipe)
                _cmp(f'gemm[{pipe}] M{M} N{N} K{K} act{act}', out, ref.float(), errs, 1e-5, 2e-5)
    # A = I with an asymmetric B catches transposed fragment/C layouts
    I = torch.eye(96)
    Wb = torch.arange(96 * 96, dtype=torch.float32).reshape(96, 96) / 97.0
    for pipe in ('fp32', 'split'):
        _cmp(f'gemm[{pipe}] identity', eng.op_gemm(I.cuda(), Wb.cuda(), None, 0, pipe=pipe), Wb.T.contiguous(), errs, 0, 1e-6)
    assert not errs, '\n'.join(errs)


def test_split_bf16_pipe_is_fp32_arithmetic(hip_device):
    """The default matrix pipe (exact three-way bf16 split of both operands, six v_mfma_f32_32x32x16_bf16 per 16-deep step) against
    an fp64 reference, beside the fp32 MFMA kernel on the same operands: its error, normalised by sum_k |a_k b_k|, must stay at
    the level of an fp32 accumulation (a few 1e-7) and within 1.25x of the fp32 MFMA chain's -- normal operands and operands
    spread over ~5 decades, K = 96 .. 3136 (the path's depths).  A bf16 (or 2-term) product would be 100-1000x off."""
    g = G.load('small_b2')
    eng, _ = _engine(g)
    gen = torch.Generator().manual_seed(3)
    rows = []
    for wide in (False, True):
        # the last shapes reach the other instantiations: 256-row block tiles (>= 512 of them), 128- and 64-column tiles at full grids
        for (M, N, K) in [(256, 96, 96), (256, 384, 96), (256, 96, 384), (128, 192, 768), (128, 256, 3136), (256, 64, 576),
                          (65536, 192, 96), (16384, 256, 128), (32768, 64, 192)]:
            A = torch.randn(M, K, generator=gen)
            W = torch.randn(N, K, generator=gen)
            if wide:
                A = A * torch.exp(2.0 * torch.randn(M, K, generator=gen))
                W = W * torch.exp(2.0 * torch.randn(N, K, generator=gen))
            ref = A.double() @ W.double().T
            mag = A.double().abs() @ W.double().abs().T
            e = {}
            for pipe in ('fp32', 'split'):
                out = eng.op_gemm(A.cuda(), W.cuda(), None, 0, pipe=pipe).cpu().double()
                err = ((out - ref).abs() / mag)
                e[pipe] = (float(err.max()), float((err ** 2).mean().sqrt()))
            rows.append((wide, M, N, K, e))
            print(f"{'wide  ' if wide else 'normal'} M{M} N{N} K{K}: fp32 mfma max {e['fp32'][0]:.2e} rms {e['fp32'][1]:.2e} | split max {e['split'][0]:.2e} rms {e['split'][1]:.2e}")
            assert e['split'][1] <= 1.25 * e['fp32'][1] + 1e-9, (wide, M, N, K, e)
            assert e['split'][0] <= (3e-6 if wide else 4e-7), (wide, M, N, K, e)


def test_ln_gemm_kernel_vs_fp64(hip_device):
    """LayerNorm in the A path of the linear behind it (round 5: gemm.hip A_LN + swin.hip ln_stats_kernel; the QKV and fc1 linears of Swin
    stages 2-4, mmdet swin.py:358,365) against an fp64 reference of LN-then-linear, beside the two-kernel form it replaces (fp32
    LayerNorm by torch, then the split GEMM): every shape class of the path (K = C = 192 / 384 / 768, N = 3C and 4C, identity and
    gathered rows, GELU), rows with a large common offset (|mean| = 50 sigma: the subtraction happens BEFORE the product, so nothing
    cancels), and a row count that is not a multiple of the tile."""
    g = G.load('small_b2')
    eng, _ = _engine(g)
    gen = torch.Generator().manual_seed(11)
    for (T, M, N, K, act, offset) in [(512, None, 576, 192, 0, 0.0), (1000, 777, 768, 192, 2, 0.0), (256, None, 1152, 384, 0, 50.0),
                                      (300, 300, 1536, 384, 2, 0.0), (130, None, 2304, 768, 0, 3.0), (256, 200, 3072, 768, 2, 50.0),
                                      (66000, None, 576, 192, 0, 0.0),
                                      # N = 64: the FPN laterals behind the stages' output norms (swin.py:756-762 -> fpn.py:152), under- and well-filled grids
                                      (1000, None, 64, 96, 0, 0.0), (513, 400, 64, 192, 0, 3.0), (300, None, 64, 768, 0, 0.0), (70000, None, 64, 96, 0, 0.0)]:
        x = torch.randn(T, K, generator=gen) * (1.0 + torch.rand(T, 1, generator=gen)) + offset * torch.randn(T, 1, generator=gen).sign()
        w = torch.randn(N, K, generator=gen) / K ** 0.5
        b = torch.randn(N, generator=gen) * 0.1
        lg = 1.0 + 0.2 * torch.randn(K, generator=gen)
        lb = 0.1 * torch.randn(K, generator=gen)
        rows = torch.randperm(T, generator=gen)[:M].to(torch.int32) if M is not None else None
        xs = x[rows.long()] if rows is not None else x
        xd = xs.double()
        y = (xd - xd.mean(1, keepdim=True)) / (xd.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt() * lg.double() + lb.double()
        ref = y @ w.double().T + b.double()
        if act == 2:
            ref = torch.nn.functional.gelu(ref)
        mag = y.abs() @ w.double().abs().T + b.double().abs()
        out = eng.op_ln_gemm(x.cuda(), w, b, lg, lb, rows.cuda() if rows is not None else None, act).cpu().double()
        two = eng.op_gemm(torch.nn.functional.layer_norm(xs, (K,), lg, lb, 1e-5).cuda(), w.cuda(), b.cuda(), act, pipe='split').cpu().double()
        e1, e2 = ((out - ref).abs() / mag), ((two - ref).abs() / mag)
        print(f'T{T} M{M} N{N} K{K} act{act} offset{offset}: A_LN max {float(e1.max()):.2e} rms {float((e1 ** 2).mean().sqrt()):.2e} | LN kernel + GEMM max {float(e2.max()):.2e} rms {float((e2 ** 2).mean().sqrt()):.2e}')
        assert torch.isfinite(out).all()
        assert float((e1 ** 2).mean().sqrt()) <= 1.5 * float((e2 ** 2).mean().sqrt()) + 2e-8, (T, M, N, K)
        assert float(e1.max()) <= (2e-5 if offset >= 50 else 2e-6), (T, M, N, K, float(e1.max()))


def test_patch_merging_in_one_launch_vs_fp64(hip_device):
    """PatchMerging (mmdet transformer.py:363-385: nn.Unfold(2, stride 2) -> LayerNorm(4C) -> Linear(4C, 2C, bias=False)) as ONE launch: the A rows
    of the reduction linear are gathered from the token tensor as two runs of 2C floats, the norm rides in the A path and its statistics are
    merged from per-token partials over 96 channels (gemm.hip A_LN with seg_k; 4 / 8 / 16 partials per merged row).  Against an fp64
    restatement by torch's own Unfold, beside the two-kernel form (fp32 LayerNorm by torch, then the split GEMM): the three merges of the
    path, a non-square grid, a row count that is not a tile multiple, tokens with a large common offset."""
    g = G.load('small_b2')
    eng, _ = _engine(g)
    gen = torch.Generator().manual_seed(31)
    for (B, H, W, C, offset) in [(2, 64, 64, 96, 0.0), (3, 32, 32, 192, 0.0), (5, 16, 16, 384, 0.0), (1, 10, 6, 96, 0.0), (2, 12, 20, 192, 30.0), (1, 6, 10, 384, 3.0)]:
        x = torch.randn(B * H * W, C, generator=gen) * (1.0 + torch.rand(B * H * W, 1, generator=gen)) + offset * torch.randn(B * H * W, 1, generator=gen).sign()
        w = torch.randn(2 * C, 4 * C, generator=gen) / (4 * C) ** 0.5
        lg = 1.0 + 0.2 * torch.randn(4 * C, generator=gen)
        lb = 0.1 * torch.randn(4 * C, generator=gen)
        u = torch.nn.functional.unfold(x.view(B, H, W, C).permute(0, 3, 1, 2), kernel_size=2, stride=2).transpose(1, 2).reshape(-1, 4 * C)   # channel-major: k = c*4 + kh*2 + kw
        ud = u.double()
        y = (ud - ud.mean(1, keepdim=True)) / (ud.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt() * lg.double() + lb.double()
        ref = y @ w.double().T
        mag = y.abs() @ w.double().abs().T + 1e-30
        out = eng.op_merge_ln_gemm(x.cuda(), B, H, W, w, lg, lb).cpu().double()
        two = eng.op_gemm(torch.nn.functional.layer_norm(u, (4 * C,), lg, lb, 1e-5).cuda(), w.cuda(), None, 0, pipe='split').cpu().double()
        e1, e2 = ((out - ref).abs() / mag), ((two - ref).abs() / mag)
        print(f'B{B} {H}x{W} C{C} offset{offset}: one launch max {float(e1.max()):.2e} rms {float((e1 ** 2).mean().sqrt()):.2e} | LN kernel + GEMM max {float(e2.max()):.2e} rms {float((e2 ** 2).mean().sqrt()):.2e}')
        assert out.shape == ref.shape and torch.isfinite(out).all()
        assert float((e1 ** 2).mean().sqrt()) <= 1.5 * float((e2 ** 2).mean().sqrt()) + 2e-8, (B, H, W, C)
        assert float(e1.max()) <= (2e-5 if offset >= 30 else 2e-6), (B, H, W, C, float(e1.max()))


def test_epilogue_ln_statistics_chain_vs_fp64(hip_device):
    """The engine's chain for the norms of Swin stages 2-4 (round 5): the GEMM that PRODUCES a token tensor (proj + residual with its row
    scatter, fc2 + residual, patch merging) leaves, per row and 96 columns, {mean, sum of squared deviations} of what it stores
    (GemmParams.stats_out); the linear behind the LayerNorm merges those partials in its A path (A_LN) -- no pass over the tensor computes
    statistics.  Against fp64 LayerNorm-then-linear of the tensor the producer actually stored, beside the two-kernel form: C = 192 / 384 /
    768 (2 / 4 / 8 partials per row), with and without residual and row map, a row count that is not a tile multiple, rows with a large
    common offset."""
    g = G.load('small_b2')
    eng, _ = _engine(g)
    gen = torch.Generator().manual_seed(23)
    for (M, Kp, K, N, act, use_res, use_map, offset) in [(512, 192, 192, 768, 2, True, True, 0.0), (1000, 768, 192, 576, 0, True, False, 0.0),
                                                          (300, 384, 384, 1536, 2, True, True, 30.0), (130, 1536, 384, 1152, 0, False, False, 0.0),
                                                          (256, 768, 768, 3072, 2, True, True, 0.0), (200, 1536, 768, 2304, 0, False, False, 30.0),
                                                          (33000, 192, 192, 576, 0, True, True, 0.0)]:
        a = torch.randn(M, Kp, generator=gen)
        wp = torch.randn(K, Kp, generator=gen) / Kp ** 0.5
        bp = torch.randn(K, generator=gen) * 0.1
        res = (torch.randn(M, K, generator=gen) * 2.0 + offset * torch.randn(M, 1, generator=gen).sign()) if use_res else None
        rmap = torch.randperm(M, generator=gen).to(torch.int32) if use_map else None
        w = torch.randn(N, K, generator=gen) / K ** 0.5
        b = torch.randn(N, generator=gen) * 0.1
        lg = 1.0 + 0.2 * torch.randn(K, generator=gen)
        lb = 0.1 * torch.randn(K, generator=gen)
        y, c = eng.op_gemm_ln_gemm(a.cuda(), wp, bp, w, b, lg, lb, res.cuda() if res is not None else None, rmap.cuda() if rmap is not None else None, act)
        y, c = y.cpu(), c.cpu().double()
        # the producer itself: y[rmap[m]] = a wp^T + bp + res[rmap[m]]
        yref = a.double() @ wp.double().T + bp.double()
        dst = rmap.long() if rmap is not None else torch.arange(M)
        if res is not None:
            yref = yref + res.double()[dst]
        full = torch.zeros(M, K, dtype=torch.float64)
        full[dst] = yref
        assert float((y.double() - full).abs().max()) <= 2e-5 * max(1.0, float(full.abs().max())), (M, K)
        yd = y.double()
        yn = (yd - yd.mean(1, keepdim=True)) / (yd.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt() * lg.double() + lb.double()
        ref = yn @ w.double().T + b.double()
        if act == 2:
            ref = torch.nn.functional.gelu(ref)
        mag = yn.abs() @ w.double().abs().T + b.double().abs()
        two = eng.op_gemm(torch.nn.functional.layer_norm(y, (K,), lg, lb, 1e-5).cuda(), w.cuda(), b.cuda(), act, pipe='split').cpu().double()
        e1, e2 = ((c - ref).abs() / mag), ((two - ref).abs() / mag)
        print(f'M{M} Kp{Kp} K{K} N{N} act{act} res{use_res} map{use_map} offset{offset}: chain max {float(e1.max()):.2e} rms {float((e1 ** 2).mean().sqrt()):.2e} | LN kernel + GEMM max {float(e2.max()):.2e} rms {float((e2 ** 2).mean().sqrt()):.2e}')
        assert torch.isfinite(c).all()
        assert float((e1 ** 2).mean().sqrt()) <= 1.5 * float((e2 ** 2).mean().sqrt()) + 3e-8, (M, K, N)
        assert float(e1.max()) <= (2e-5 if offset else 2e-6), (M, K, N, float(e1.max()))


@pytest.mark.parametrize('case', ['small_b2', 'small_wsi_b3', 'full_b1', 'five_b2'])
def test_dense_stages_vs_oracle_and_golden(hip_device, case):
    from oracle import model as O
    g = G.load(case)
    eng, sd = _engine(g)
    eng.enable_token_dump()
    tiles = g['tiles']
    B = len(tiles)
    mode = int(g['channel_mode'])
    eng.infer_async(eng.to_device(tiles), mode)
    eng.check()
    errs = []
    img = O.preprocess(tiles, mode)
    _cmp('img', eng.buffer('img')[:B].permute(0, 3, 1, 2), img, errs, 0, 1e-6)
    with torch.no_grad():
        c, toks = O.backbone(sd, img, return_tokens=True)
        x = O.fpn(sd, c)
        rcls, rreg = O.rpn_convs(sd, x)
        sp, sf = O.semantic_head(sd, x)
    for s in range(4):
        for b in range(O.DEPTHS[s]):
            t = eng.buffer(f'tok_s{s}b{b}')[:B]
            _cmp(f'tok_s{s}b{b}', t, toks[f's{s}b{b}'], errs)
            G.check_sub(g, f's{s}b{b}', t, RTOL * 5, ATOL * 5)
    for i in range(4):
        ci = eng.buffer(f'c{i}')[:B].permute(0, 3, 1, 2)
        _cmp(f'c{i}', ci, c[i], errs)
        xi = eng.buffer(f'x{i}')[:B].permute(0, 3, 1, 2)
        _cmp(f'x{i}', xi, x[i], errs)
        G.check_sub(g, f'x{i}', xi.contiguous(), RTOL * 5, ATOL * 5)
        r = eng.buffer(f'rpn{i}')[:B].permute(0, 3, 1, 2)
        _cmp(f'rpn_cls{i}', r[:, 0:3], rcls[i], errs)
        _cmp(f'rpn_reg{i}', r[:, 3:15], rreg[i], errs)
    _cmp('sem_pred', eng.buffer('sem_pred')[:B][:, None], sp, errs, 5e-4, 5e-4)
    _cmp('sem_feat', eng.buffer('sem_feat')[:B].permute(0, 3, 1, 2), sf, errs, 5e-4, 5e-4)
    G.check_sub(g, 'sem_feat', eng.buffer('sem_feat')[:B].permute(0, 3, 1, 2).contiguous(), 2e-3, 2e-3)
    assert not errs, '\n'.join(errs)


def _split_vs_fp32(eng, A, W, label, max_rel, rms_ratio=1.25):
    """Both pipes on the same operands against fp64; errors relative to sum_k |a_k b_k| (per output), printed class by class."""
    ref = A.double() @ W.double().T
    mag = A.double().abs() @ W.double().abs().T
    e = {}
    for pipe in ('fp32', 'split'):
        out = eng.op_gemm(A.cuda(), W.cuda(), None, 0, pipe=pipe).cpu().double()
        assert torch.isfinite(out).all(), (label, pipe)
        err = (out - ref).abs() / mag
        e[pipe] = (float(err.max()), float((err ** 2).mean().sqrt()))
    print(f"{label}: fp32 mfma max {e['fp32'][0]:.2e} rms {e['fp32'][1]:.2e} | split max {e['split'][0]:.2e} rms {e['split'][1]:.2e}")
    assert e['split'][0] <= max_rel, (label, e)
    assert e['split'][1] <= rms_ratio * e['fp32'][1] + 1e-9, (label, e)
    return e


def test_split_pipe_hard_operand_classes(hip_device):
    """The classes a 'this is fp32 arithmetic' claim has to survive: catastrophic cancellation, operands whose
    first round-to-nearest split crosses a binade, third planes that are subnormal, the top of the exponent range, and Inf / NaN
    propagation -- split pipe beside the fp32 MFMA chain, both against fp64, class by class."""
    g = G.load('small_b2')
    eng, _ = _engine(g)
    gen = torch.Generator().manual_seed(11)
    M, N, K = 256, 96, 384
    # (1) cancellation: pairs (a, a) x (w, -w (1 + d)), d ~ 1e-6: sum a*b ~ 1e-6 of sum |a*b|.  The error is measured against
    # sum |a*b| (the only bound any fp32 summation has); what must hold is that the split pipe does not lose the small net sum's
    # leading digits more than the fp32 chain does
    a = torch.randn(M, K // 2, generator=gen)
    w = torch.randn(N, K // 2, generator=gen)
    d = 1e-6 * torch.randn(N, K // 2, generator=gen)
    A = torch.stack([a, a], 2).reshape(M, K)
    W = torch.stack([w, -w * (1 + d)], 2).reshape(N, K)
    _split_vs_fp32(eng, A, W, 'cancellation (net sum ~1e-6 of sum|ab|)', 4e-7)
    # (2) first split rounds up across a binade: values just below a power of two (a1 = 2^k, negative residual planes)
    k2 = torch.randint(-6, 7, (M, K), generator=gen).float()
    A = torch.exp2(k2) * (2.0 - torch.rand(M, K, generator=gen) * 2e-3) * (torch.randint(0, 2, (M, K), generator=gen) * 2 - 1)
    W = torch.exp2(torch.randint(-6, 7, (N, K), generator=gen).float()) * (2.0 - torch.rand(N, K, generator=gen) * 2e-3)
    _split_vs_fp32(eng, A, W, 'just below powers of two (split rounds up a binade)', 4e-7)
    # (3) subnormal third planes: |a| ~ 2^-110 puts a3 ~ 2^-126 .. 2^-134 below the smallest normal bf16 / fp32; the products
    # (~2^-110) and the results are normal numbers
    A = torch.randn(M, K, generator=gen) * 2.0 ** -110
    W = torch.randn(N, K, generator=gen)
    _split_vs_fp32(eng, A, W, 'third plane of A subnormal (|a| ~ 2^-110)', 4e-7)
    A = torch.randn(M, K, generator=gen)
    W = torch.randn(N, K, generator=gen) * 2.0 ** -110
    _split_vs_fp32(eng, A, W, 'third plane of W subnormal (|w| ~ 2^-110)', 4e-7)
    # (4) top of the range: |a| up to the largest bf16 (0x7f7f0000 = 3.3895e38; above it the first split rounds to Inf, a documented
    # domain limit of the pipe), weights small enough that the sums stay finite
    A = (torch.rand(M, K, generator=gen) * 0.5 + 0.5) * 3.38e38 * (torch.randint(0, 2, (M, K), generator=gen) * 2 - 1)
    W = torch.randn(N, K, generator=gen) * 1e-4
    _split_vs_fp32(eng, A, W, 'top binade (|a| up to 3.38e38)', 4e-7)
    # (5) Inf / NaN: a non-finite operand makes exactly the outputs it touches non-finite on both pipes (an Inf may come out as NaN on
    # the split pipe: Inf - bf16(Inf) is NaN in the residual planes), every other output stays finite and accurate
    A = torch.randn(M, K, generator=gen)
    W = torch.randn(N, K, generator=gen)
    A[3, 5] = float('inf'); A[100, 17] = float('-inf'); A[200, 300] = float('nan')
    W[7, 9] = float('nan'); W[50, 383] = float('inf')
    ref = torch.nan_to_num(A, nan=0.0, posinf=0.0, neginf=0.0).double() @ torch.nan_to_num(W, nan=0.0, posinf=0.0, neginf=0.0).double().T
    bad = torch.zeros(M, N, dtype=torch.bool)
    bad[[3, 100, 200], :] = True
    bad[:, [7, 50]] = True
    for pipe in ('fp32', 'split'):
        out = eng.op_gemm(A.cuda(), W.cuda(), None, 0, pipe=pipe).cpu()
        assert torch.equal(~torch.isfinite(out), bad), pipe
        err = ((out.double() - ref).abs() / (A.double().abs().nan_to_num(0, 0, 0) @ W.double().abs().nan_to_num(0, 0, 0).T))[~bad]
        assert float(err.max()) <= 4e-7, (pipe, float(err.max()))


def test_fused_mlp_kernel_vs_fp64(hip_device):
    """csrc/mlp.hip (LN2 + fc1 + GELU + fc2 + residual in one kernel, hidden tensor in registers) against an fp64 evaluation of
    mmdet swin.py:365-367 and beside the three separate launches' arithmetic (fp32 torch): its error must be at the fp32 level.
    Token counts that are not a multiple of the 256-token block, large activations, and in-place use."""
    g = G.load('small_b2')
    eng, _ = _engine(g)
    gen = torch.Generator().manual_seed(5)
    C = 96
    for T, scale in [(256, 1.0), (1000, 1.0), (31, 3.0), (4096 + 17, 10.0)]:
        x = torch.randn(T, C, generator=gen) * scale + 0.3
        ln_g = 1 + 0.1 * torch.randn(C, generator=gen)
        ln_b = 0.1 * torch.randn(C, generator=gen)
        w1 = torch.randn(4 * C, C, generator=gen) / C ** 0.5
        b1 = 0.1 * torch.randn(4 * C, generator=gen)
        w2 = torch.randn(C, 4 * C, generator=gen) / (4 * C) ** 0.5
        b2 = 0.1 * torch.randn(C, generator=gen)

        def ref(dt):
            xd = x.to(dt)
            xn = torch.nn.functional.layer_norm(xd, (C,), ln_g.to(dt), ln_b.to(dt), 1e-5)
            h = torch.nn.functional.gelu(xn @ w1.to(dt).T + b1.to(dt))
            return xd + h @ w2.to(dt).T + b2.to(dt)
        r64, r32 = ref(torch.float64), ref(torch.float32)
        d = {k: v.cuda() for k, v in dict(x=x, g=ln_g, b=ln_b, b1=b1, b2=b2).items()}      # kept alive across the raw-pointer call below
        out = eng.op_swin_mlp(d['x'], d['g'], d['b'], w1, d['b1'], w2, d['b2']).cpu()
        e_hip = float((out.double() - r64).abs().max())
        e_f32 = float((r32.double() - r64).abs().max())
        mag = float(r64.abs().max())
        print(f'fused mlp T{T} scale {scale}: max abs err {e_hip:.2e} (torch fp32 chain {e_f32:.2e}), |out| max {mag:.2f}')
        assert torch.isfinite(out).all()
        assert e_hip <= max(4.0 * e_f32, 2e-6 * mag), (T, e_hip, e_f32)
        # in place (the engine's use): out aliases x
        xd = d['x'].clone()
        w1h, w2h = w1.numpy(), w2.numpy()
        eng._check(eng.lib.nuhtc_op_swin_mlp(eng.h, xd.data_ptr(), d['g'].data_ptr(), d['b'].data_ptr(), w1h.ctypes.data, d['b1'].data_ptr(),
                                             w2h.ctypes.data, d['b2'].data_ptr(), xd.data_ptr(), T, C, eng._stream()))
        assert torch.equal(xd.cpu(), out)


def test_fused_proj_mlp_kernel_vs_fp64(hip_device):
    """csrc/mlp.hip with the attention projection in front (round 4: x' = x + Wp att + bp, then LN2 + fc1 + GELU + fc2 + residual on x',
    one kernel: the second half of a Swin block from the attention output on, mmdet swin.py:360-367) against an fp64 evaluation and beside
    the fp32 torch chain; ragged token counts, large activations, and the engine's in-place use (out aliases x)."""
    g = G.load('small_b2')
    eng, _ = _engine(g)
    gen = torch.Generator().manual_seed(7)
    C = 96
    for T, scale in [(256, 1.0), (1000, 1.0), (31, 3.0), (4096 + 17, 10.0)]:
        x = torch.randn(T, C, generator=gen) * scale + 0.3
        att = torch.randn(T, C, generator=gen) * scale
        wp = torch.randn(C, C, generator=gen) / C ** 0.5
        bp = 0.1 * torch.randn(C, generator=gen)
        ln_g = 1 + 0.1 * torch.randn(C, generator=gen)
        ln_b = 0.1 * torch.randn(C, generator=gen)
        w1 = torch.randn(4 * C, C, generator=gen) / C ** 0.5
        b1 = 0.1 * torch.randn(4 * C, generator=gen)
        w2 = torch.randn(C, 4 * C, generator=gen) / (4 * C) ** 0.5
        b2 = 0.1 * torch.randn(C, generator=gen)

        def ref(dt):
            xd = x.to(dt) + att.to(dt) @ wp.to(dt).T + bp.to(dt)
            xn = torch.nn.functional.layer_norm(xd, (C,), ln_g.to(dt), ln_b.to(dt), 1e-5)
            h = torch.nn.functional.gelu(xn @ w1.to(dt).T + b1.to(dt))
            return xd + h @ w2.to(dt).T + b2.to(dt)
        r64, r32 = ref(torch.float64), ref(torch.float32)
        d = {k: v.cuda() for k, v in dict(x=x, att=att, bp=bp, g=ln_g, b=ln_b, b1=b1, b2=b2).items()}
        out = eng.op_swin_proj_mlp(d['x'], d['att'], wp, d['bp'], d['g'], d['b'], w1, d['b1'], w2, d['b2']).cpu()
        e_hip = float((out.double() - r64).abs().max())
        e_f32 = float((r32.double() - r64).abs().max())
        mag = float(r64.abs().max())
        print(f'fused proj + mlp T{T} scale {scale}: max abs err {e_hip:.2e} (torch fp32 chain {e_f32:.2e}), |out| max {mag:.2f}')
        assert torch.isfinite(out).all()
        assert e_hip <= max(4.0 * e_f32, 2e-6 * mag), (T, e_hip, e_f32)
        assert torch.equal(d['x'].cpu(), x)                      # the input is left alone when out is another buffer
        xd = d['x'].clone()                                      # in place (the engine's use)
        h = lambda w: np.ascontiguousarray(w.numpy(), dtype=np.float32)
        wph, w1h, w2h = h(wp), h(w1), h(w2)
        eng._check(eng.lib.nuhtc_op_swin_proj_mlp(eng.h, xd.data_ptr(), d['att'].data_ptr(), wph.ctypes.data, d['bp'].data_ptr(), d['g'].data_ptr(),
                                                  d['b'].data_ptr(), w1h.ctypes.data, d['b1'].data_ptr(), w2h.ctypes.data, d['b2'].data_ptr(),
                                                  xd.data_ptr(), T, C, eng._stream()))
        assert torch.equal(xd.cpu(), out)
