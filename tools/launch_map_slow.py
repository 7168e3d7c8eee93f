import json,sys
for m in ("unet","siam-conc","bit-cd"):
    rows=json.load(open(f"gpurun_out/lm_{m}.json"))
    print("==",m,len(rows))
    for r in rows:
        ks=" ".join(r["kernels"])
        if "igemm_fwd_kernel" in ks or "igemm_wgrad_kernel" in ks or "im2col" in ks or "maxpool3s2_bwd" in ks or ("igemm2_fwd" in ks and r["ms"]>0.06):
            print(r["i"], r["kind"], r["tag"], "|", ks[:90], "|", r["ms"], "ms", r["bytes"]//1000000,"MB")
