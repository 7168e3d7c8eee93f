"""HIP graph replay of the fused train step (kurosiwo_amd/trainer.py: capture_graph): the static launch list, the device-side optimizer
step counter and the device-side random-stream state make a captured step replayable; it must reproduce the eager trajectory bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches(n, B, C, S, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(B, C, S, S, generator=g), torch.randn(B, C, S, S, generator=g), torch.randint(0, 3, (B, S, S), generator=g)) for _ in range(n)]


@pytest.mark.parametrize("family", ["snunet", "fcsiam"])
def test_graph_replay_equals_eager_steps(family):
    from kurosiwo_amd.trainer import CDTrainStep
    B, S = 2, 64
    data = _batches(4, B, 2, S, 7)

    def make():
        torch.manual_seed(3)
        if family == "snunet":
            from kurosiwo_amd.snunet import SNUNet_ECAM
            m = SNUNet_ECAM(2, 3, base_channel=32, precision="bf16")
        else:                                   # Dropout2d on: the random-stream step advances inside the graph
            from kurosiwo_amd.fcsiam import SiamUnet_conc
            m = SiamUnet_conc(2, 3, precision="bf16")
            m.manual_seed(11, 0)
        m = m.cuda().train()
        return m, CDTrainStep(m, B, S, S, "ce+dice", (1.0, 1.0, 1.0), lr=1e-3)
    m1, s1 = make()
    losses1 = []
    for xA, xB, y in data:
        losses1.append(s1.step(xA.cuda(), xB.cuda(), y.cuda()).clone())
    m2, s2 = make()
    losses2 = []
    for i, (xA, xB, y) in enumerate(data):
        s2.set_batch(xA.cuda(), xB.cuda(), y.cuda())
        if i == 0:
            s2.capture_graph()                  # = one eager step (warm-up) + the capture
        else:
            s2.run()                            # graph replay
        losses2.append(s2.loss_out.clone())
    assert s2._graph is not None
    for a, b in zip(losses1, losses2):
        assert torch.equal(a, b), (a.tolist(), b.tolist())
    assert torch.equal(m1.flat_params, m2.flat_params)
    if family == "fcsiam":
        assert m2.rng_state().cpu().tolist() == [11, 4]
    # a learning-rate change re-captures instead of replaying the stale rate
    s2.optimizer.param_groups[0]["lr"] = 5e-4
    s1.optimizer.param_groups[0]["lr"] = 5e-4
    xA, xB, y = data[0]
    l1 = s1.step(xA.cuda(), xB.cuda(), y.cuda()).clone()
    l2 = s2.step(xA.cuda(), xB.cuda(), y.cuda()).clone()
    assert torch.equal(l1, l2) and torch.equal(m1.flat_params, m2.flat_params)


def test_main_entry_with_hip_graph(tmp_path, monkeypatch):
    """main.py --hip_graph: the trainer replays the captured step; same final mIoU as the eager run on the same synthetic set."""
    import os
    import shutil
    import main as entry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for flag in ([], ["--hip_graph"]):
        wd = tmp_path / ("g" if flag else "e")
        wd.mkdir()
        shutil.copytree(os.path.join(root, "configs"), wd / "configs")
        monkeypatch.chdir(wd)
        monkeypatch.setenv("KSMI_SYNTHETIC_TILES", "8,4,4")
        res.append(entry.main(["--method", "snunet", "--inputs", "pre_event_1", "post_event", "--batch_size", "4"] + flag))
    assert res[0] == res[1], res


def _lane_case(family, B, S, overlap, graph):
    from kurosiwo_amd.trainer import CDTrainStep, SegTrainStep
    torch.manual_seed(5)
    kw = dict(lr=1e-3, overlap_wgrad=overlap, overlap_lanes=overlap, graph=graph)
    if family == "snunet":
        from kurosiwo_amd.snunet import SNUNet_ECAM
        m = SNUNet_ECAM(2, 3, base_channel=32, precision="bf16").cuda().train()
        return m, CDTrainStep(m, B, S, S, "ce+dice", (1.0, 2.0, 3.0), **kw)
    if family == "bitcd":
        from kurosiwo_amd.bitcd import define_G
        m = define_G({"net_G": "base_resnet18"}, 2, precision="bf16").cuda().train()
        return m, CDTrainStep(m, B, S, S, "ce+dice", (1.0, 2.0, 3.0), **kw)
    if family == "floodvit":            # token plan: tagged side-stream gradients behind explicit waits (plan_base.side_tokens)
        from kurosiwo_amd.floodvit import FinetunerSegmentation, ViT
        enc = ViT(image_size=S, patch_size=16, num_classes=1000, dim=1024, depth=6, heads=16, mlp_dim=2048, channels=2)
        m = FinetunerSegmentation(enc, {"decoder": True, "num_classes": 3}, precision="bf16").cuda().train()
        return m, SegTrainStep(m, B, "cross_entropy", (1.0, 2.0, 3.0), **kw)
    if family == "changeformer":        # MiT encoder: tagged side-stream gradients of every block's linears (changeformer_plan._encoder_stage_bwd)
        from kurosiwo_amd.changeformer import ChangeFormerV6
        m = ChangeFormerV6(input_nc=2, output_nc=3, decoder_softmax=True, embed_dim=256, precision="bf16").cuda().train()
        return m, CDTrainStep(m, B, S, S, "ce+dice", (1.0, 2.0, 3.0), **kw)
    from kurosiwo_amd.unet import Unet
    m = Unet("resnet18", encoder_weights=None, in_channels=2, classes=3, precision="bf16").cuda().train()
    return m, SegTrainStep(m, B, "cross_entropy", (1.0, 2.0, 3.0), image_size=(S, S), **kw)


@pytest.mark.parametrize("family,graph", [("snunet", False), ("snunet", True), ("bitcd", False), ("unet", False), ("unet", True),
                                          ("floodvit", False), ("floodvit", True), ("changeformer", False), ("changeformer", True)])
def test_side_lane_equals_single_stream(family, graph):
    """trainer.py overlap_wgrad / overlap_lanes: the weight-gradient launches run on a side stream and SNUNet's deeper decoder blocks on a
    second compute lane (snunet_plan.StepStreams); every kernel is deterministic, so a missing dependency edge would show up as a
    different trajectory -- it must equal the single-stream one bit for bit, eagerly and as a captured graph (fork / join become graph
    edges).  224 x 224 tiles: launches long enough to really overlap."""
    B, S = (16, 224) if family == "floodvit" else (4, 224)       # the ViT at the benchmark's token count (3152 rows)
    data = _batches(5, B, 2, S, 21)
    out = []
    for overlap in (False, True):
        m, st = _lane_case(family, B, S, overlap, graph and overlap)
        assert st.overlap_wgrad == overlap
        losses = []
        for xA, xB, y in data:
            args = (xA.cuda(), y.cuda()) if family in ("unet", "floodvit") else (xA.cuda(), xB.cuda(), y.cuda())
            losses.append(st.step(*args).clone())
        torch.cuda.synchronize()
        if overlap:
            assert st._ss is not None and (st._graph is not None) == graph
            assert any(meta.get("side") for _, _, _, meta in st.plan.bwd.calls)
            if family == "floodvit":
                tags = [meta["side_tag"] for _, _, _, meta in st.plan.bwd.calls if meta.get("side_tag")]
                waits = [args[0] for fn, args, name, _ in st.plan.bwd.calls if name == "@wait_side"]
                assert len(tags) == 4 * 6 and set(waits) - {None} <= set(tags) and len([w for w in waits if w]) == 2 * 6 + 2 * 5
            if family == "changeformer":
                tags = [meta["side_tag"] for _, _, _, meta in st.plan.bwd.calls if meta.get("side_tag")]
                assert len([t for t in tags if t.endswith((".fc1", ".q", ".kv"))]) == 3 * 13, tags     # depths 3 + 3 + 4 + 3
            if family == "snunet":
                assert st._ss.lanes and any(meta["lane"] == 1 for _, _, _, meta in st.plan.fwd.calls + st.plan.bwd.calls)
        out.append((losses, m.flat_params.clone(), m.flat_grads.clone()))
    for a, b in zip(out[0][0], out[1][0]):
        assert torch.equal(a, b), (a.tolist(), b.tolist())
    assert torch.equal(out[0][2], out[1][2])
    assert torch.equal(out[0][1], out[1][1])


def test_lanes_without_the_side_stream_equal_single_stream():
    """overlap_lanes on, overlap_wgrad off (configs["overlap_wgrad"] = False / KSMI_OVERLAP_WGRAD=0): the weight gradients of the two compute
    lanes then run concurrently on their lanes' streams, so their partial-slab scratch must be per lane (round-2 advisor finding: one
    shared `wgrad` scratch was only safe while every weight gradient serialised on the side stream).  Bitwise-equal trajectory."""
    from kurosiwo_amd.snunet import SNUNet_ECAM
    from kurosiwo_amd.trainer import CDTrainStep
    B, S = 4, 224
    data = _batches(4, B, 2, S, 33)
    out = []
    for lanes in (False, True):
        torch.manual_seed(5)
        m = SNUNet_ECAM(2, 3, base_channel=32, precision="bf16").cuda().train()
        st = CDTrainStep(m, B, S, S, "ce+dice", (1.0, 2.0, 3.0), lr=1e-3, overlap_wgrad=False, overlap_lanes=lanes)
        losses = [st.step(xA.cuda(), xB.cuda(), y.cuda()).clone() for xA, xB, y in data]
        torch.cuda.synchronize()
        if lanes:
            assert st._ss is not None and st._ss.lanes and not st._ss.use_side
            names = set(st.plan._need)
            assert "wgrad" in names and "wgrad@1" in names          # one partial-slab scratch per lane
        out.append((losses, m.flat_params.clone(), m.flat_grads.clone()))
    for a, b in zip(out[0][0], out[1][0]):
        assert torch.equal(a, b), (a.tolist(), b.tolist())
    assert torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("family", ["snunet", "floodvit", "changeformer", "unet"])
def test_compiled_launch_list_equals_the_python_walk(family, monkeypatch):
    """snunet_plan.LaunchList: the compiled list (ONE ksmi_run_list call per segment, csrc/runlist.hip: typed call thunks, event ring,
    tagged side-stream events) issues the same launches on the same streams behind the same dependency edges as the Python walk
    (KSMI_RUN_LIST=0): multi-stream trajectories equal bit for bit, losses, gradients and parameters."""
    from kurosiwo_amd import snunet_plan as sp
    B, S = (16, 224) if family == "floodvit" else (4, 224)
    data = _batches(4, B, 2, S, 47)
    out = []
    for fast in (False, True):
        monkeypatch.setattr(sp.LaunchList, "fast", fast)
        m, st = _lane_case(family, B, S, True, False)
        losses = []
        for xA, xB, y in data:
            args = (xA.cuda(), y.cuda()) if family in ("unet", "floodvit") else (xA.cuda(), xB.cuda(), y.cuda())
            losses.append(st.step(*args).clone())
        torch.cuda.synchronize()
        assert (st.plan.bwd._compiled is not None) == fast and (st._ss._runner is not None) == fast
        out.append((losses, m.flat_params.clone(), m.flat_grads.clone()))
    for a, b in zip(out[0][0], out[1][0]):
        assert torch.equal(a, b), (a.tolist(), b.tolist())
    assert torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][1], out[1][1])
