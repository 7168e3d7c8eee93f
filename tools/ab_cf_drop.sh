#!/bin/bash
# round 6: Mlp.drop fused into the depth-wise / gelu' passes (KSMI_CF_FUSE_DROP): parity tests, bit-equality of the two routes, same-box A/B
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "stored_bf16" 2>&1 | grep -E "assert|Error|passed|failed|^E" | head -20
python -m pytest tests/test_gpu_cformer.py tests/test_gpu_changeformer.py -x -q -m gpu 2>&1 | tail -3
python - <<'PY'
import os, torch
from kurosiwo_amd.changeformer import ChangeFormerV6
from kurosiwo_amd.trainer import CDTrainStep
out = []
g = torch.Generator().manual_seed(3)
data = [(torch.randn(2, 2, 224, 224, generator=g), torch.randn(2, 2, 224, 224, generator=g), torch.randint(0, 3, (2, 224, 224), generator=g)) for _ in range(3)]
for fuse in ("0", "1"):
    os.environ["KSMI_CF_FUSE_DROP"] = fuse
    torch.manual_seed(5)
    m = ChangeFormerV6(input_nc=2, output_nc=3, decoder_softmax=True, embed_dim=64, precision="bf16").cuda().train()
    m.manual_seed(11, 0)
    st = CDTrainStep(m, 2, 224, 224, "ce+dice", (1.0, 1.0, 1.0), lr=1e-3)
    names = [n for _, _, n, _ in st.plan.fwd.calls + st.plan.bwd.calls]
    losses = [st.step(a.cuda(), b.cuda(), y.cuda()).clone() for a, b, y in data]
    torch.cuda.synchronize()
    out.append((losses, m.flat_params.clone(), m.flat_grads.clone(), len(names), names.count("ksmi_dropout_apply")))
print("launches / dropout launches:", out[0][3], out[0][4], "->", out[1][3], out[1][4])
print("bit-equal losses", all(torch.equal(a, b) for a, b in zip(out[0][0], out[1][0])), "grads", torch.equal(out[0][2], out[1][2]), "params", torch.equal(out[0][1], out[1][1]))
PY
for f in 0 1 0 1; do KSMI_CF_FUSE_DROP=$f python bench.py --model changeformer --channels 4 --steps 20 --warmup 5 --no-cpu-baseline --no-solo 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('FUSE_DROP=$f', d['value'], d['ms_per_step'])"; done
