# determinism stress of the forward form of the wide-output kernel and of the stream weight-gradient kernel: repeated runs must be bit-identical
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mobilenet_yolo_pytorch_amd import _lib
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = ((96, 576, 20480), (64, 384, 20480), (96, 512, 40960), (80, 192, 20480), (576, 96, 20480), (160, 960, 30976))
if len(sys.argv) > 1 and sys.argv[1] == "more":      # LDS-DMA and short-reduction kernels
    shapes = ((512, 512, 30976), (1280, 512, 30976), (384, 64, 123904), (24, 144, 495616), (16, 96, 495616), (144, 24, 495616), (320, 1280, 30976), (960, 160, 30976))
for K, N, M in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    sc, sh = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
    dy = torch.randn(M, N, generator=g).cuda()
    parts = _lib.query("mny_pw_stat_parts", M, K, N)
    wsn = _lib.query("mny_pw_wgrad_ws_floats", M, K, N)
    outs = []
    for it in range(30):
        junk = torch.randn(1 << 22, device="cuda")
        st = torch.full((parts, 2, N), float("nan"), device="cuda")
        y = torch.empty(M, N, device="cuda")
        _lib.call("mny_pw_fwd", p(x), p(sc), p(sh), 1, p(w), None, None, p(y), p(st), M, K, N, stream)
        ws = torch.zeros(wsn, device="cuda"); dw = torch.empty(N, K, device="cuda")
        _lib.call("mny_pw_wgrad", p(x), p(sc), p(sh), 1, p(dy), p(dw), None, p(ws), M, K, N, stream)
        torch.cuda.synchronize()
        outs.append((y.clone(), st.clone(), dw.clone()))
        del junk
    bad = [sum(1 for o in outs[1:] if not torch.equal(o[i], outs[0][i])) for i in range(3)]
    print("K=%d N=%d M=%d: runs differing in y %d, stats %d, dW %d (of 29)" % (K, N, M, bad[0], bad[1], bad[2]))
