import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from shifu_amd._lib import lib
L = lib()
import ctypes
raw = ctypes.CDLL(os.environ["SHIFU_AMD_LIB"])
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 24576
for K, N, act in ((512, 256, 1), (256, 128, 1)):
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05, torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    for _ in range(3):
        L.shf_mlp_linear_forward(p(x), p(w), p(b), p(y), M, K, N, act, st)
    torch.cuda.synchronize()
    nb = (M // 128) * ((N + 127) // 128)
    buf = (C.c_longlong * (4 * nb))()
    raw.shf_mlp_probe_read(buf, 4 * nb)
    a = np.array(buf[:]).reshape(nb, 4)
    for xcd in range(2):
        sel = a[xcd::8]
        o = np.argsort(sel[:, 0])
        print("  xcd", xcd, "starts (k ticks, sorted):", ((sel[o, 0] - sel[o, 0][0]) // 1000).tolist())
        print("  xcd", xcd, "ends:", ((sel[o, 2] - sel[o, 0][0]) // 1000).tolist())
    t0 = a[:, 0].min()
    print(K, N, "blocks", nb, "start spread", (a[:, 0].max() - t0), "loop mean", (a[:, 1] - a[:, 0]).mean(), "epi mean", (a[:, 2] - a[:, 1]).mean(),
          "total span", a[:, 2].max() - t0, "loop min/max", (a[:, 1] - a[:, 0]).min(), (a[:, 1] - a[:, 0]).max(), "epi min/max", (a[:, 2] - a[:, 1]).min(), (a[:, 2] - a[:, 1]).max())
