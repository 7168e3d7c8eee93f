# A/B of one C-ABI entry point between two builds of libksmi.so: put the other tree (git archive <rev> | tar -x, make) under old_snapshot/ and run from the repo root
import ctypes as C, torch
new = C.CDLL("kurosiwo_amd/libksmi.so"); old = C.CDLL("old_snapshot/kurosiwo_amd/libksmi.so")
torch.manual_seed(0)
for dt, tdt in ((1, torch.bfloat16), (0, torch.float32)):
  for (npix, Cc, rows) in ((8192, 32, 32), (2 * 64 * 64, 64, 32), (1000, 128, 3), (50176 * 2, 256, 392), (77, 8, 5), (4096, 512, 16)):
    x = torch.randn(npix, Cc, device="cuda").to(tdt)
    res = []
    for lib in (old, new):
        part = torch.zeros(rows, Cc, device="cuda")
        f = lib.ksmi_channel_sum; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        rc = f(x.data_ptr(), part.data_ptr(), rows, npix, Cc, dt, None); torch.cuda.synchronize()
        res.append((rc, part.clone()))
    print(dt, npix, Cc, rows, res[0][0], res[1][0], "equal", bool(torch.equal(res[0][1], res[1][1])), float((res[1][1].sum(0) - x.float().sum(0)).abs().max()))
