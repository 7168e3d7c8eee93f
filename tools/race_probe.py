"""Repeat the single-stream vs side-stream comparison of a token train step: python tools/race_probe.py mae|floodvit|changeformer [repeats] [delay] [nowait]
delay: every step first parks the side stream behind a ~10 ms spin kernel and every side-stream launch behind a ~150 us one, so the
main stream runs as far ahead of each weight gradient as its waits allow -- a missing wait then shows up as a different trajectory
instead of depending on launch timing."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

fam, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6
delay = len(sys.argv) > 3
nowait = len(sys.argv) > 4            # negative control: drop the tagged waits -> the comparison must fail


def stepper(st):
    def go(*a):
        ss = st._streams()
        if delay and ss is not None and not hasattr(ss, "_slow"):
            # every launch handed to the side stream first spins ~150 us there (and the whole stream ~10 ms at the start of a step)
            fork = ss.fork_side

            def slow_fork():
                ptr = fork()
                with torch.cuda.stream(ss.side):
                    torch.cuda._sleep(350_000)
                return ptr
            ss.fork_side, ss._slow = slow_fork, True
            if nowait:
                ss.wait_side = lambda tag: None
        if delay and ss is not None:
            with torch.cuda.stream(ss.side):
                torch.cuda._sleep(25_000_000)
        return st.step(*a).clone()
    return go


def run(overlap, data):
    os.environ["KSMI_OVERLAP_WGRAD"] = overlap
    torch.manual_seed(5)
    if fam == "mae":
        from test_gpu_mae import build
        from kurosiwo_amd.trainer import MAETrainStep
        hp = dict(image_size=224, patch_size=16, dim=1024, depth=4, heads=16, mlp_dim=2048, channels=2, decoder_dim=512, decoder_depth=3, decoder_heads=16)
        model, _ = build(hp, "bf16")
        st = MAETrainStep(model, 32, lr=1e-4)
        go = stepper(st)
        losses = [go(x.cuda(), idx.cuda()) for x, idx, _ in data]
    elif fam == "changeformer":
        from kurosiwo_amd.changeformer import ChangeFormerV6
        from kurosiwo_amd.trainer import CDTrainStep
        model = ChangeFormerV6(input_nc=2, output_nc=3, decoder_softmax=True, embed_dim=256, precision="bf16").cuda().train()
        st = CDTrainStep(model, 4, 224, 224, "ce+dice", (1.0, 2.0, 3.0), lr=1e-3)
        go = stepper(st)
        losses = [go(x[:4].cuda(), x[4:8].cuda(), y[:4].cuda()) for x, _, y in data]
    else:
        from kurosiwo_amd.floodvit import FinetunerSegmentation, ViT
        from kurosiwo_amd.trainer import SegTrainStep
        enc = ViT(image_size=224, patch_size=16, num_classes=1000, dim=1024, depth=6, heads=16, mlp_dim=2048, channels=2)
        model = FinetunerSegmentation(enc, {"decoder": True, "num_classes": 3}, precision="bf16").cuda().train()
        st = SegTrainStep(model, 16, "cross_entropy", (1.0, 2.0, 3.0), lr=1e-3)
        go = stepper(st)
        losses = [go(x[:16].cuda(), y[:16].cuda()) for x, _, y in data]
    torch.cuda.synchronize()
    return torch.stack(losses), model.flat_params.clone(), model.flat_grads.clone(), model


g = torch.Generator().manual_seed(9)
data = [(torch.randn(32, 2, 224, 224, generator=g), torch.rand(32, 196, generator=g).argsort(dim=-1), torch.randint(0, 3, (32, 224, 224), generator=g)) for _ in range(4)]
base = run("0", data)
for r in range(reps):
    for ov in ("0", "1"):
        got = run(ov, data)
        same = [bool(torch.equal(a, b)) for a, b in zip(base[:3], got[:3])]
        msg = ""
        if not all(same):
            m = got[3]
            bad = [k for k, off in m._poff.items() if not torch.equal(base[2][off:off + m._p(k).numel()], got[2][off:off + m._p(k).numel()])]
            msg = f" differing gradients: {bad[:12]} ({len(bad)})"
        print(f"rep {r} overlap {ov}: losses/params/grads equal = {same}{msg}", flush=True)
