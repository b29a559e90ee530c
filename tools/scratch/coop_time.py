import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from curve25519_amd import synth
dev = torch.device('cuda', 0)
n = 1
sk_np, pk_np = synth.x25519_inputs(64)
sk, pk = torch.from_numpy(sk_np).to(dev), torch.from_numpy(pk_np).to(dev)
out = torch.empty((64, 32), dtype=torch.uint8, device=dev)
vp, sz = C.c_void_p, C.c_size_t
for path in sys.argv[1:]:
    L = C.CDLL(os.path.abspath(path)); L.curve25519_dh_CreateSharedKey_dev.argtypes = [vp, vp, vp, sz, vp]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: L.curve25519_dh_CreateSharedKey_dev(out.data_ptr(), pk.data_ptr(), sk.data_ptr(), n, st)
    for _ in range(5): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    print(f"{os.path.basename(path):30s} {sorted(ts)[len(ts)//2]:8.1f} us")
