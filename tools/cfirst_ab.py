# A/B of one C-ABI entry point between two builds of libksmi.so: put the other tree (git archive <rev> | tar -x, make) under old_snapshot/ and run from the repo root
import ctypes as C, torch, sys
new = C.CDLL("kurosiwo_amd/libksmi.so"); old = C.CDLL("old_snapshot/kurosiwo_amd/libksmi.so")
torch.manual_seed(0)
for dt, tdt in ((1, torch.bfloat16), (0, torch.float32)):
  for (B, Cin, H, W, Cout) in ((2, 2, 64, 64, 32), (1, 3, 40, 24, 32), (2, 2, 224, 224, 32), (1, 2, 32, 32, 16), (1, 4, 32, 48, 64)):
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.3; b = torch.randn(Cout, device="cuda")
    res = []
    for lib in (old, new):
        rows = lib.ksmi_conv_first_stats_rows(B, H, W)
        out = torch.zeros(B, H, W, Cout, device="cuda", dtype=tdt); st = torch.zeros(rows, 2, Cout, device="cuda")
        f = lib.ksmi_conv_first_forward_raw
        f.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]
        rc = f(x.data_ptr(), None, Cin, w.data_ptr(), b.data_ptr(), out.data_ptr(), st.data_ptr(), B, Cin, H, W, Cout, None, None, None, dt, None)
        torch.cuda.synchronize()
        res.append((rc, out.float().clone(), st.sum(0).clone(), st.clone()))
    ref = torch.nn.functional.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1)
    print(dt, (B, Cin, H, W, Cout), "rc", res[0][0], res[1][0], "out maxdiff new-old", float((res[0][1] - res[1][1]).abs().max()),
          "vs ref old/new", float((res[0][1] - ref).abs().max()), float((res[1][1] - ref).abs().max()),
          "stats sum diff", float((res[0][2] - res[1][2]).abs().max()), "rel", float((res[0][2] - res[1][2]).abs().max() / res[0][2].abs().max()),
          "ref stats", float((res[1][2][0] - ref.sum((0, 1, 2))).abs().max()), float((res[1][2][1] - (ref * ref).sum((0, 1, 2))).abs().max() / (ref*ref).sum((0,1,2)).max()))
