"""MI355X: seeded shape fuzzing of the pipelined kernels (LDS-DMA rings with counted waits, raw LDS reads, asm prefetches):
ragged M / K / N tails, tiny problems, tile-boundary sizes.  fp32 against torch (float64 accumulation), 2e-4 relative."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from mobilenet_yolo_pytorch_amd import ops as o
    return o


def _close(got, ref, rel=2e-4, what=""):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, "%s: err %.3e vs scale %.3e" % (what, err, scale)


ACT = {0: lambda z: z, 1: lambda z: torch.clamp(z, 0, 6), 2: lambda z: F.leaky_relu(z, 0.1), 3: torch.relu}


def test_pointwise_gemm_shapes(ops):
    r = np.random.RandomState(0)
    ks = [4, 8, 12, 16, 20, 24, 32, 36, 48, 64, 100, 160, 516]           # K % 4 == 0: DMA kernel, incl. ragged K % 16
    ns = [1, 3, 4, 10, 16, 31, 32, 33, 75, 96, 100, 128, 144, 160, 161, 320, 512]
    ms = [1, 5, 63, 64, 127, 128, 129, 255, 300, 1000, 4097]
    for trial in range(40):
        M, K, N = int(r.choice(ms)), int(r.choice(ks)), int(r.choice(ns))
        act = int(r.randint(0, 4))
        g = torch.Generator().manual_seed(trial)
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) * K ** -0.5
        sc, sh = 1 + 0.2 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
        a = ACT[act](x * sc + sh)
        xs = x.view(1, 1, M, K).cuda()
        use_bias = N % 4 != 0 or trial % 3 == 0
        b = torch.randn(N, generator=g) if use_bias else None
        add = torch.randn(M, N, generator=g) if (trial % 2 and use_bias) else None   # (statistics are of conv outputs: never with an addend)
        got, st = ops.pw_fwd((xs, sc.cuda(), sh.cuda(), act), w.cuda(), bias=b.cuda() if use_bias else None,
                             addend=add.view(1, 1, M, N).cuda() if add is not None else None, want_stats=not use_bias)
        ref = a.double() @ w.double().t() + (b.double() if use_bias else 0) + (add.double() if add is not None else 0)
        _close(got.view(M, N), ref, what="pw_fwd M%d K%d N%d act%d" % (M, K, N, act))
        if not use_bias:
            _close(st[:, 0].double().sum(0), ref.sum(0), 3e-4, "stats sum M%d K%d N%d" % (M, K, N))
            _close(st[:, 1].double().sum(0), (ref ** 2).sum(0), 3e-4, "stats sumsq M%d K%d N%d" % (M, K, N))
        dy = torch.randn(M, N, generator=g)
        dw, db = ops.pw_wgrad((xs, sc.cuda(), sh.cuda(), act), dy.view(1, 1, M, N).cuda(), want_dbias=True)
        _close(dw, dy.double().t() @ a.double(), what="pw_wgrad M%d K%d N%d" % (M, K, N))
        _close(db, dy.double().sum(0), what="dbias")


def test_fused_units_shapes(ops):
    r = np.random.RandomState(1)
    import ctypes
    from mobilenet_yolo_pytorch_amd import _lib
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    for trial in range(12):                                              # depthwise 3x3 s1 unit backward
        N, H, W = int(r.choice([1, 2, 5])), int(r.choice([1, 2, 7, 16, 17, 33])), int(r.choice([1, 3, 8, 19, 40]))
        C = int(r.choice([4, 8, 36, 96, 132, 516]))
        act, xact = int(r.randint(0, 4)), int(r.randint(0, 4))
        g = torch.Generator().manual_seed(100 + trial)
        x = torch.randn(N, C, H, W, generator=g)
        w = torch.randn(C, 1, 3, 3, generator=g) * 0.4
        xs, xh = 1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
        gamma, beta = 1 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
        a_in = ACT[xact](x * xs.view(1, -1, 1, 1) + xh.view(1, -1, 1, 1)).detach().requires_grad_(True)
        wr, gr, br = w.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        y_raw = F.conv2d(a_in, wr, None, 1, 1, 1, C)
        if N * H * W < 2:
            continue                                                      # BatchNorm needs more than one value per channel
        out = ACT[act](F.batch_norm(y_raw, None, None, gr, br, True, 0.1, 1e-5))
        gg = torch.randn(*out.shape, generator=g)
        out.backward(gg)
        nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()        # noqa: E731
        yd, gd, xd = nh(y_raw.detach()), nh(gg), nh(x)
        M = N * H * W
        y32 = yd.view(M, C)
        st = torch.stack((y32.sum(0), (y32 ** 2).sum(0))).view(1, 2, C).contiguous()
        scale, shift, mean, invstd = ops.bn_finalize(st, M, gamma.cuda(), beta.cuda())
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        parts = _lib.query("mny_bn_bwd_parts", M, C)
        red = torch.empty(parts, 2, C, device="cuda")
        gam_d = gamma.cuda()
        _lib.call("mny_bn_bwd_reduce", p(gd), p(yd), p(scale), p(shift), act, p(mean), p(invstd), p(red), M, C, stream)
        dgamma, dbeta, coef = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(3, C, device="cuda")
        _lib.call("mny_bn_bwd_finalize", p(red), parts, M, p(gam_d), p(mean), p(invstd), p(dgamma), p(dbeta), p(coef), C, stream)
        dx, dw = ops.dw_bnbwd(gd, yd, scale, shift, act, coef, (xd, xs.cuda(), xh.cuda(), xact), w.cuda().contiguous())
        what = "dw_bnbwd N%d H%d W%d C%d act%d/%d" % (N, H, W, C, act, xact)
        _close(dx.permute(0, 3, 1, 2), a_in.grad, 1e-3, what + " dX")    # train-mode BN on few samples amplifies rounding
        _close(dw, wr.grad, 1e-3, what + " dW")


def test_round2_stencil_kernels_shapes(ops):
    """Seeded shape fuzz of the round-2 stencil kernels: odd sizes, one-pixel rows / columns, channel counts that split into
    several chunks.  (a) 3x3 / 5x5 depthwise forward + weight gradient + stride-1 data gradient against torch;
    (b) the fused stride-2 unit backward against the three launches it replaces; (c) the producer BN sums of the fused stride-1
    backward against mny_bn_bwd_reduce."""
    import ctypes
    from mobilenet_yolo_pytorch_amd import _lib
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()               # noqa: E731
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731
    r = np.random.RandomState(7)
    acts = dict(ACT)
    acts[4] = lambda z: z * F.relu6(z + 3) / 6
    for trial in range(24):
        N, H, W = int(r.choice([1, 2, 3])), int(r.choice([1, 2, 5, 8, 17, 33, 40])), int(r.choice([1, 3, 4, 9, 22, 41]))
        C = int(r.choice([4, 16, 40, 132, 260, 672]))
        K, S = int(r.choice([3, 5])), int(r.choice([1, 2]))
        act = int(r.choice([0, 1, 2, 3, 4]))
        g = torch.Generator().manual_seed(300 + trial)
        x = torch.randn(N, C, H, W, generator=g)
        w = torch.randn(C, 1, K, K, generator=g) * 0.3
        sc, sh = 1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
        a = acts[act](x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        yref = F.conv2d(a, wr, None, S, K // 2, 1, C)
        dy = torch.randn(*yref.shape, generator=g)
        yref.backward(dy)
        what = "N%d H%d W%d C%d K%d s%d act%d" % (N, H, W, C, K, S, act)
        view = (nh(x), sc.cuda(), sh.cuda(), act)
        got, st = ops.dw_fwd(view, w.cuda().contiguous(), S)
        _close(got.permute(0, 3, 1, 2), yref, 2e-4, "dw_fwd " + what)
        _close(st[:, 0].double().sum(0), yref.detach().double().sum((0, 2, 3)), 5e-4, "dw stats " + what)
        dwg = ops.dw_bwd_weight(view, nh(dy), K, S)
        _close(dwg, wr.grad, 5e-4, "dw_bwd_weight " + what)
        dxg = ops.dw_bwd_data(nh(dy), w.cuda().contiguous(), (H, W), S)
        _close(dxg.permute(0, 3, 1, 2), a.grad, 2e-4, "dw_bwd_data " + what)
        if K != 3:
            continue
        # fused unit backward (stride 1 with producer sums / stride 2) against the unfused chain on the same inputs
        Ho, Wo = yref.shape[2], yref.shape[3]
        yraw, gg = nh(torch.randn(N, C, Ho, Wo, generator=g)), nh(torch.randn(N, C, Ho, Wo, generator=g))
        mk = lambda a0, b0: (a0 + b0 * torch.randn(C, generator=g)).cuda()   # noqa: E731
        scale, shift, coef = mk(1.0, 0.2), mk(0.0, 0.3), torch.stack((mk(1.0, 0.2), mk(0.0, 0.05), mk(0.0, 0.05))).contiguous()
        uact = int(r.choice([0, 1, 2]))
        dyb = torch.empty_like(gg)
        _lib.call("mny_bn_bwd_apply", p(gg), p(yraw), p(scale), p(shift), uact, p(coef), p(dyb), N * Ho * Wo, C, stream())
        dx_ref = ops.dw_bwd_data(dyb, w.cuda().contiguous(), (H, W), S)
        dw_ref = ops.dw_bwd_weight(view, dyb, 3, S)
        if S == 2:
            dx, dwf = ops.dw_bnbwd_s2(gg, yraw, scale, shift, uact, coef, view, w.cuda().contiguous())
        else:
            xm, xi = mk(0.0, 0.2), mk(1.0, 0.1).abs()
            dx, dwf, red = ops.dw_bnbwd(gg, yraw, scale, shift, uact, coef, view, w.cuda().contiguous(), in_stats=(xm, xi))
            parts = _lib.query("mny_bn_bwd_parts", N * H * W, C)
            ref = torch.empty(parts, 2, C, device="cuda")
            _lib.call("mny_bn_bwd_reduce", p(dx), p(view[0]), p(view[1]), p(view[2]), act, p(xm), p(xi), p(ref), N * H * W, C, stream())
            for k in range(2):
                _close(red.double().sum(0)[k], ref.double().sum(0)[k], 1e-4, "producer sums %d " % k + what)
        _close(dx, dx_ref, 3e-4, "fused dX " + what)
        _close(dwf, dw_ref, 5e-4, "fused dW " + what)


def test_short_reduction_kernel_shapes():
    """pwthin.hip through the entry points that route to it: random (M, K in {8,16,24,32}, N % 4 == 0 up to 256) around the tile
    boundaries (R*rpb rows per tile, grid = resident workgroups), with the view / bias / addend / statistics / BN-backward forms drawn
    at random; fp32 exact-path check 2e-4, bf16 one rounding of the output."""
    import ctypes
    from mobilenet_yolo_pytorch_amd import _lib
    r = np.random.RandomState(7)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None            # noqa: E731
    for trial in range(48):
        K = int(r.choice([8, 16, 24, 32]))
        N = 4 * int(r.randint(4, 65))
        nq = N // 4
        rpb = 256 // nq
        tile = (8 if 128 // rpb >= 8 else 2) * rpb
        M = int(r.choice([1, tile - 1, tile, tile + 1, 3 * tile + rpb - 1, 512 * tile, 512 * tile + 5, 769 * tile + 1, int(r.randint(1, 300000))]))
        bf = bool(r.randint(0, 2)) and K in (8, 16, 24)
        dtype, sfx = (torch.bfloat16, "_bf16") if bf else (torch.float32, "")
        rel = 2 ** -7 if bf else 2e-4
        act = int(r.randint(0, 4))
        g = torch.Generator().manual_seed(100 + trial)
        x = torch.randn(M, K, generator=g).cuda().to(dtype)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().to(dtype)
        sc, sh = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
        xd, wd = x.double().cpu(), w.double().cpu()
        form = int(r.randint(0, 4))
        y = torch.empty(M, N, device="cuda", dtype=dtype)
        if form == 0:                                    # view + statistics
            a = ACT[act](xd * sc.double().cpu() + sh.double().cpu())
            if bf:
                a = a.to(torch.bfloat16).double()        # the bf16 operand
            parts = _lib.query("mny_pw_stat_parts" + sfx, M, K, N)
            st = torch.full((parts, 2, N), float("nan"), device="cuda")
            _lib.call("mny_pw_fwd" + sfx, p(x), p(sc), p(sh), act, p(w), None, None, p(y), p(st), M, K, N, stream)
            _close(y, a @ wd.t(), rel, "trial %d view+stats M=%d K=%d N=%d" % (trial, M, K, N))
            yd = y.double().cpu()
            _close(st[:, 0].double().sum(0), yd.sum(0), 1e-5 * max(1.0, yd.abs().sum(0).max().item() / (yd.sum(0).abs().max().item() + 1e-9)), "stats sum")
            _close(st[:, 1].double().sum(0), (yd ** 2).sum(0), 1e-5, "stats sumsq")
        elif form == 1:                                  # bias + addend in place
            b = torch.randn(N, generator=g).cuda()
            add = torch.randn(M, N, generator=g).cuda().to(dtype)
            buf = add.clone()
            _lib.call("mny_pw_fwd" + sfx, p(x), None, None, 0, p(w), p(b), p(buf), p(buf), None, M, K, N, stream)
            _close(buf, xd @ wd.t() + b.double().cpu() + add.double().cpu(), rel, "trial %d bias+addend M=%d K=%d N=%d" % (trial, M, K, N))
        else:                                            # data gradient + BN-backward sums, with / without an addend
            yraw = (torch.randn(M, N, generator=g) * 2).cuda().to(dtype)
            c = [(1 + 0.3 * torch.randn(N, generator=g)).cuda(), (0.5 * torch.randn(N, generator=g)).cuda(),
                 (0.2 * torch.randn(N, generator=g)).cuda(), (1 + 0.2 * torch.randn(N, generator=g).abs()).cuda()]
            add = torch.randn(M, N, generator=g).cuda().to(dtype) if form == 3 else None
            parts = _lib.query("mny_pw_dgrad_bnred_parts" + sfx, M, K, N)
            red = torch.full((parts, 2, N), float("nan"), device="cuda")
            if add is not None:
                _lib.call("mny_pw_dgrad_bnred_add" + sfx, p(x), p(w), p(add), p(y), p(yraw), p(c[0]), p(c[1]), act, p(c[2]), p(c[3]), p(red), M, K, N, stream)
            else:
                _lib.call("mny_pw_dgrad_bnred" + sfx, p(x), p(w), p(y), p(yraw), p(c[0]), p(c[1]), act, p(c[2]), p(c[3]), p(red), M, K, N, stream)
            _close(y, xd @ wd.t() + (add.double().cpu() if add is not None else 0), rel, "trial %d dgrad form %d M=%d K=%d N=%d" % (trial, form, M, K, N))
            z = yraw.double().cpu() * c[0].double().cpu() + c[1].double().cpu()
            d = {0: torch.ones_like(z), 1: ((z > 0) & (z < 6)).double(), 2: torch.where(z > 0, 1.0, 0.1).double(), 3: (z > 0).double()}[act]
            dz = y.double().cpu() * d
            xhat = (yraw.double().cpu() - c[2].double().cpu()) * c[3].double().cpu()
            kink = ((z.abs() < 1e-5) | ((z - 6).abs() < 1e-5)).double() * y.double().cpu().abs()
            r1, r2 = red[:, 0].double().sum(0).cpu(), red[:, 1].double().sum(0).cpu()
            assert ((r1 - dz.sum(0)).abs() <= 2e-5 * dz.abs().sum(0).max().item() + 1e-5 + kink.sum(0)).all(), (trial, M, K, N)
            assert ((r2 - (dz * xhat).sum(0)).abs() <= 2e-5 * (dz * xhat).abs().sum(0).max().item() + 1e-5 + (kink * xhat.abs()).sum(0)).all(), (trial, M, K, N)


def test_barrier_free_matrix_core_kernel_shapes():
    """pwwide.hip (forward with view + statistics / plain, data gradient + BN-backward sums with / without an addend) and pwwgs.hip (weight
    gradient, direct and deferred combine) through the entry points that route to them: random K in 52..96 (K % 4 == 0), N a multiple of
    64 / 96 / 128, row counts around the 32-row tile, the per-wave tile stride and the 16-row chunk (remainder rows take the LDS-DMA kernel
    and one more partial row; M % 16 != 0 takes the LDS-DMA weight gradient).  fp32-accurate path: 2e-5 of the result's scale."""
    import ctypes
    from mobilenet_yolo_pytorch_amd import _lib
    r = np.random.RandomState(11)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None            # noqa: E731
    for trial in range(24):
        K = 4 * int(r.randint(13, 25))                                                # 52..96
        N = int(r.choice([64, 96, 128, 192, 256, 288, 384, 512, 576]))
        if N < K:
            N = 192
        M = int(r.choice([8192, 8192 + 31, 8224, 16384, 16400, 20480 + 1, 4 * 128 * 32 * 2 + 32, int(r.randint(8192, 70000)), 16 * int(r.randint(1024, 4000))]))
        act = int(r.randint(0, 4))
        g = torch.Generator().manual_seed(500 + trial)
        x = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        sc, sh = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
        xd, wd = x.double().cpu(), w.double().cpu()
        what = "trial %d M=%d K=%d N=%d act=%d" % (trial, M, K, N, act)
        form = trial % 4
        y = torch.empty(M, N, device="cuda")
        if form == 0:                                    # view + statistics
            a = ACT[act](xd * sc.double().cpu() + sh.double().cpu())
            parts = _lib.query("mny_pw_stat_parts", M, K, N)
            st = torch.full((parts, 2, N), float("nan"), device="cuda")
            _lib.call("mny_pw_fwd", p(x), p(sc), p(sh), act, p(w), None, None, p(y), p(st), M, K, N, stream)
            _close(y, a @ wd.t(), 2e-5, "view+stats " + what)
            yd = y.double().cpu()
            assert not torch.isnan(st).any(), what
            _close(st[:, 0].double().sum(0), yd.sum(0), 1e-5 * max(1.0, yd.abs().sum(0).max().item() / (yd.sum(0).abs().max().item() + 1e-9)), "stats sum " + what)
            _close(st[:, 1].double().sum(0), (yd ** 2).sum(0), 1e-5, "stats sumsq " + what)
        elif form == 1:                                  # weight gradient of a conv K -> N and of its mirror image N -> K
            dy = torch.randn(M, N, generator=g).cuda()
            a = ACT[act](xd * sc.double().cpu() + sh.double().cpu())
            for (xin, s1, s2, ac, dyin, kk, nn, ref) in ((x, sc, sh, act, dy, K, N, dy.double().cpu().t() @ a), (dy, None, None, 0, x, N, K, xd.t() @ dy.double().cpu())):
                ws = torch.zeros(_lib.query("mny_pw_wgrad_ws_floats", M, kk, nn), device="cuda")
                dw = torch.empty(nn, kk, device="cuda")
                _lib.call("mny_pw_wgrad", p(xin), p(s1), p(s2), ac, p(dyin), p(dw), None, p(ws), M, kk, nn, stream)
                _close(dw, ref, 2e-5, "wgrad %d->%d " % (kk, nn) + what)
                splits = _lib.query("mny_pw_wgrad_splits", M, kk, nn)
                ws.zero_()
                _lib.call("mny_pw_wgrad", p(xin), p(s1), p(s2), ac, p(dyin), None, None, p(ws), M, kk, nn, stream)
                _close(ws[:splits * nn * kk].view(splits, nn, kk).double().sum(0), ref, 2e-5, "wgrad partial rows %d->%d " % (kk, nn) + what)
        else:                                            # data gradient + BN-backward sums, with / without an addend (in place)
            yraw = (torch.randn(M, N, generator=g) * 2).cuda()
            c = [(1 + 0.3 * torch.randn(N, generator=g)).cuda(), (0.5 * torch.randn(N, generator=g)).cuda(),
                 (0.2 * torch.randn(N, generator=g)).cuda(), (1 + 0.2 * torch.randn(N, generator=g).abs()).cuda()]
            add = torch.randn(M, N, generator=g).cuda() if form == 3 else None
            parts = _lib.query("mny_pw_dgrad_bnred_parts", M, K, N)
            red = torch.full((parts, 2, N), float("nan"), device="cuda")
            if add is not None:
                assert _lib.query("mny_pw_dgrad_bnred_add_supported", M, K, N, act) == 1
                y.copy_(add)
                _lib.call("mny_pw_dgrad_bnred_add", p(x), p(w), p(y), p(y), p(yraw), p(c[0]), p(c[1]), act, p(c[2]), p(c[3]), p(red), M, K, N, stream)
            else:
                _lib.call("mny_pw_dgrad_bnred", p(x), p(w), p(y), p(yraw), p(c[0]), p(c[1]), act, p(c[2]), p(c[3]), p(red), M, K, N, stream)
            _close(y, xd @ wd.t() + (add.double().cpu() if add is not None else 0), 2e-5, "dgrad form %d " % form + what)
            assert not torch.isnan(red).any(), what
            z = yraw.double().cpu() * c[0].double().cpu() + c[1].double().cpu()
            d = {0: torch.ones_like(z), 1: ((z > 0) & (z < 6)).double(), 2: torch.where(z > 0, 1.0, 0.1).double(), 3: (z > 0).double()}[act]
            dz = y.double().cpu() * d
            xhat = (yraw.double().cpu() - c[2].double().cpu()) * c[3].double().cpu()
            kink = ((z.abs() < 1e-5) | ((z - 6).abs() < 1e-5)).double() * y.double().cpu().abs()
            r1, r2 = red[:, 0].double().sum(0).cpu(), red[:, 1].double().sum(0).cpu()
            assert ((r1 - dz.sum(0)).abs() <= 2e-5 * dz.abs().sum(0).max().item() + 1e-5 + kink.sum(0)).all(), what
            assert ((r2 - (dz * xhat).sum(0)).abs() <= 2e-5 * (dz * xhat).abs().sum(0).max().item() + 1e-5 + (kink * xhat.abs()).sum(0)).all(), what
