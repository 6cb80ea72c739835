"""In-kernel clock of k_mlp_panel (build: tools/mlp_probe.py build clock -DSHF_MLP_PROBE_CLOCK): per block start, end of
the prologue (first K chunk in LDS), end of the reduction loop, end of the epilogue -- shader clock cycles; then of
k_mlp_chain: start, end of layer 0, end of layer 1, end of the last layer."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from shifu_amd._lib import lib
L = lib()
raw = C.CDLL(os.environ["SHIFU_AMD_LIB"])
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 24576
for K, N, act in ((512, 256, 1), (256, 128, 1), (128, 12, 0)):
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05, torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    nb = C.c_int64(); L.shf_mlp_pack_bytes(K, N, C.byref(nb))
    pack = torch.empty(nb.value, device="cuda", dtype=torch.uint8)
    L.shf_mlp_pack_weights(p(w), p(pack), K, N, st)
    for _ in range(3):
        L.shf_mlp_panel_forward(p(x), p(pack), p(b), p(y), M, K, N, act, st)
    torch.cuda.synchronize()
    for bm in (64, 32):
        nblk = M // bm
        buf = (C.c_longlong * (4 * nblk))()
        raw.shf_mlp_probe_read(buf, 4 * nblk)
        a = np.array(buf[:]).reshape(nblk, 4)
        if a[-1, 0] == 0: continue
        t0 = a[:, 0].min()
        d = lambda i, j: (a[:, j] - a[:, i])
        print(K, N, "blocks", nblk, "start spread", a[:, 0].max() - t0, "span", a[:, 3].max() - t0,
              "| prologue mean/max", d(0, 1).mean().round(), d(0, 1).max(), "| loop", d(1, 2).mean().round(), d(1, 2).max(),
              "| epilogue", d(2, 3).mean().round(), d(2, 3).max())
        o = np.argsort(a[:, 0])
        print("   starts (ticks):", (a[o, 0] - t0)[::max(1, nblk // 24)].tolist())
        break

# k_mlp_chain: the A1 network 259 -> 512 -> 256 -> 128 -> 12 in one launch
from shifu_amd.rl import mfma_linear as ML
for M in (4096, 24576):
    dims = (259, 512, 256, 128, 12)
    net = ML.MfmaMLP(*[m for i in range(4) for m in (ML.MfmaLinear(dims[i], dims[i + 1], elu=i < 3), torch.nn.Identity())][:-1]).cuda()
    ls = [m for m in net if isinstance(m, ML.MfmaLinear)]
    x = torch.randn(M, 259, device="cuda")
    ML.refresh_packs(net)
    ys = [torch.empty(M, d, device="cuda") for d in dims[1:]]
    for _ in range(3):
        ML._chain_call(x, ls, [m._pack for m in ls], ys)
    torch.cuda.synchronize()
    nblk = M // 32
    buf = (C.c_longlong * (4 * nblk))()
    raw.shf_mlp_probe_read(buf, 4 * nblk)
    a = np.array(buf[:]).reshape(nblk, 4)
    d = lambda i, j: (a[:, j] - a[:, i])
    print(M, "blocks", nblk, "layer0 mean/max", d(0,1).mean().round(), d(0,1).max(), "| layer1", d(1,2).mean().round(), d(1,2).max(), "| layers 2+3", d(2,3).mean().round(), d(2,3).max(), "| total", d(0,3).mean().round())
