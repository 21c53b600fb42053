"""Generate the golden fixtures in tests/golden/ from the UNMODIFIED reference (run in the build
container, where /root/reference exists):

    python tests/golden/make_golden.py

For every fixture the script (1) runs the reference function on seeded inputs, (2) checks the
`oracle/` restatement against it right here (bit-exact for box/index work, 1e-5 for the network)
and (3) stores inputs + reference outputs as small .npz files. The reference cannot travel to the
GPU box, the fixtures do; tests/test_oracle_golden.py re-checks the oracle against them everywhere.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle.refimport import load_reference  # noqa: E402
load_reference()
from oracle import boxes_np as bx  # noqa: E402
from oracle.retina_torch import OracleRetinaUNet  # noqa: E402
from oracle.detweights import fill_state  # noqa: E402
from nndetection_amd.plans import get_plan, MODEL_CFG_V001  # noqa: E402

import nndet.core.boxes as rb  # noqa: E402
from nndet.core.boxes.nms import nms_cpu  # noqa: E402
from nndet.core.boxes.ops import box_center_dist  # noqa: E402
from nndet.core.boxes.coder import BoxCoderND  # noqa: E402
from nndet.core.boxes.anchors import get_anchor_generator  # noqa: E402


def rand_boxes(rng, n, extent=(160, 160, 96), smin=2, smax=26):
    c = rng.uniform(0, 1, (n, 3)) * np.asarray(extent)
    s = rng.uniform(smin, smax, (n, 3))
    lo, hi = c - s / 2, c + s / 2
    return np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1).astype(np.float32)


def eq(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype.kind == "f":
        ok = np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
        if not ok:  # allow NaN==NaN
            ok = np.array_equal(a, b, equal_nan=True)
        assert ok, f"{what}: oracle != reference (max abs diff {np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))})"
    else:
        assert np.array_equal(a, b), f"{what}: oracle != reference"
    print(f"  [bit-exact] {what} {a.shape}")


def golden_boxes():
    rng = np.random.default_rng(0)
    g = {}
    # ---- pairwise IoU / GIoU / centre distance
    b1, b2 = rand_boxes(rng, 37), rand_boxes(rng, 501)
    b2[5] = b1[3]                      # an identical pair
    b2[6] = [10, 10, 10, 20, 5, 9]     # zero-volume box
    t1, t2 = torch.from_numpy(b1), torch.from_numpy(b2)
    g["iou_b1"], g["iou_b2"] = b1, b2
    g["iou"] = rb.box_iou(t1, t2).numpy()
    g["iou_eps"] = rb.box_iou(t1, t2, eps=1e-6).numpy()
    g["giou"] = rb.generalized_box_iou(t1, t2).numpy()
    g["giou_eps"] = rb.generalized_box_iou(t1, t2, eps=1e-7).numpy()
    g["cdist"] = box_center_dist(t1, t2)[0].numpy()
    eq(bx.box_iou(b1, b2), g["iou"], "box_iou")
    eq(bx.box_iou(b1, b2, 1e-6), g["iou_eps"], "box_iou eps")
    eq(bx.generalized_box_iou(b1, b2), g["giou"], "generalized_box_iou")
    eq(bx.generalized_box_iou(b1, b2, 1e-7), g["giou_eps"], "generalized_box_iou eps")
    # torch-CPU's vectorised sqrt (MKL/SLEEF build in this container) is not correctly rounded: 0.5 % of
    # the values are 1 ulp off IEEE sqrt. CUDA sqrtf and HIP sqrtf are correctly rounded, numpy too; the
    # squared distance (before sqrt) is bit-exact. So the distance fixture is pinned to <= 1 ulp.
    d_or = bx.box_center_dist(b1, b2)
    ulp = np.abs(d_or.view(np.int32).astype(np.int64) - g["cdist"].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, ulp.max()
    print(f"  [<=1 ulp] box_center_dist {d_or.shape}: {int((ulp > 0).sum())} of {ulp.size} values differ by 1 ulp (torch-CPU sqrt)")
    g["cdist_sq"] = (torch.from_numpy(bx.box_center(b1))[:, None] - torch.from_numpy(bx.box_center(b2))[None]).pow(2).sum(-1).numpy()
    eq((d_or.astype(np.float64) ** 2 * 0 + ((lambda d: (d[..., 0] + d[..., 1]) + d[..., 2])((bx.box_center(b1)[:, None, :] - bx.box_center(b2)[None, :, :]) ** 2))).astype(np.float32), g["cdist_sq"], "squared centre distance")
    # ---- anchors (AnchorGenerator3DS through its forward())
    W = [(4, 8, 16), (8, 16, 32), (16, 32, 64)]
    gen = get_anchor_generator(3, s_param=True)(width=W, height=W, depth=W, stride=1)
    img = torch.zeros(2, 1, 48, 40, 24)
    fms = [torch.zeros(2, 8, 12, 10, 6), torch.zeros(2, 8, 6, 5, 3), torch.zeros(2, 8, 3, 3, 3)]
    anc = gen(img, fms)
    g["anchors"] = anc[0].numpy()
    g["anchors_per_level"] = np.asarray(gen.get_num_acnhors_per_level())
    a_or, npl = bx.anchors_for_image((48, 40, 24), [(12, 10, 6), (6, 5, 3), (3, 3, 3)], W, W, W)
    eq(a_or, g["anchors"], "anchors")
    assert npl == list(g["anchors_per_level"])
    # ---- 2D NMS (round 4): the reference's CPU path nms_cpu on [N, 4] boxes (box_iou_union_2d), nndet/core/boxes/nms.py:31-53
    from nndet.core.boxes.nms import nms_cpu
    c2 = rng.uniform(0, 60, (600, 2)); s2 = rng.uniform(4, 20, (600, 2))
    b2 = np.stack([c2[:, 0] - s2[:, 0] / 2, c2[:, 1] - s2[:, 1] / 2, c2[:, 0] + s2[:, 0] / 2, c2[:, 1] + s2[:, 1] / 2], 1).astype(np.float32)
    sc2 = ((rng.permutation(600) + 1) / 601).astype(np.float32)
    g["nms2d_boxes"], g["nms2d_scores"] = b2, sc2
    for thr in (0.1, 0.5):
        k2 = nms_cpu(torch.from_numpy(b2), torch.from_numpy(sc2), thr).numpy()
        g[f"nms2d_keep_{thr}"] = k2
        eq(bx.nms2d(b2, sc2, thr), k2, f"nms2d thr {thr}")
    # ---- ATSS (tie-free GT placement: centres off the anchor-centre lattice)
    gt = np.asarray([[5.3, 7.1, 17.9, 22.2, 3.7, 12.4],
                     [20.6, 11.3, 41.1, 30.9, 6.2, 21.7],
                     [30.2, 2.9, 36.8, 9.7, 14.1, 19.9],
                     [1.1, 30.4, 9.8, 38.6, 1.3, 6.6]], np.float32)
    matcher = rb.ATSSMatcher(num_candidates=4, similarity_fn=rb.box_iou, center_in_gt=False)
    mq, matches = matcher(torch.from_numpy(gt), anc[0], num_anchors_per_level=npl, num_anchors_per_loc=27)
    g["atss_gt"], g["atss_matches"] = gt, matches.numpy()
    iou_or, m_or = bx.atss_match(gt, a_or, npl, 27, 4)
    eq(m_or, g["atss_matches"], "atss matches")
    eq(iou_or, mq.numpy(), "atss match_quality_matrix")
    print("    positives per gt:", [(g["atss_matches"] == i).sum() for i in range(4)])
    # center_in_gt=True (the reference's default, atss.py:101-107; V001 sets it False): anchor centres must lie inside the GT box
    # (its own GT set: three more boxes whose above-threshold candidates partly / all have their centre outside)
    gt_c = np.concatenate([gt, np.asarray([[9.3, 10.2, 13.6, 14.9, 5.1, 8.4], [22.2, 14.1, 28.9, 30.3, 8.2, 13.7], [25.2, 17.1, 26.9, 30.3, 9.2, 10.7]], np.float32)], 0)
    matcher_c = rb.ATSSMatcher(num_candidates=4, similarity_fn=rb.box_iou, center_in_gt=True)
    _, matches_c = matcher_c(torch.from_numpy(gt_c), anc[0], num_anchors_per_level=npl, num_anchors_per_loc=27)
    _, matches_n = matcher(torch.from_numpy(gt_c), anc[0], num_anchors_per_level=npl, num_anchors_per_loc=27)
    g["atss_gt_center_in_gt"], g["atss_matches_center_in_gt"] = gt_c, matches_c.numpy()
    _, mc_or = bx.atss_match(gt_c, a_or, npl, 27, 4, center_in_gt=True)
    eq(mc_or, g["atss_matches_center_in_gt"], "atss matches (center_in_gt)")
    assert 0 < (g["atss_matches_center_in_gt"] >= 0).sum() < (matches_n.numpy() >= 0).sum(), "the centre test must remove some positives"
    print("    positives per gt with / without center_in_gt:", [((g["atss_matches_center_in_gt"] == i).sum(), (matches_n.numpy() == i).sum())
                                                                 for i in range(len(gt_c))])
    # ---- IoUMatcher (round 4; the skeleton module's matcher, nndet/core/boxes/matcher/iou.py:20-107)
    for low, high, lq in ((0.3, 0.5, False), (0.4, 0.6, True)):
        im = rb.IoUMatcher(low_threshold=low, high_threshold=high, allow_low_quality_matches=lq, similarity_fn=rb.box_iou)
        _, mi = im(torch.from_numpy(gt_c), anc[0], num_anchors_per_level=npl, num_anchors_per_loc=27)
        g[f"ioumatch_{low}_{high}_{int(lq)}"] = mi.numpy()
        eq(bx.iou_match(gt_c, a_or, low, high, lq)[1], mi.numpy(), f"IoUMatcher low {low} high {high} low-quality {lq}")
        print(f"    IoUMatcher({low}, {high}, {lq}): matched {(mi >= 0).sum().item()}, between {(mi == -2).sum().item()}")
    mq0, m0 = matcher(torch.zeros(0, 6), anc[0], num_anchors_per_level=npl, num_anchors_per_loc=27)
    assert mq0.numel() == 0 and (m0 == -1).all()
    lab_or, mb_or = bx.assign_targets(m_or, gt, np.asarray([0, 0, 0, 0], np.float32), a_or.shape[0])
    # reference assign (retina.py:258-288) restated through torch ops on the reference outputs
    mt = matches.clamp(min=0)
    lab_ref = torch.zeros(4)[mt] + 1
    lab_ref[matches == -1] = 0
    eq(lab_or, lab_ref.numpy(), "assigned labels")
    eq(mb_or, torch.from_numpy(gt)[mt].numpy(), "matched gt boxes")
    # ---- NMS (distinct scores), reference CPU path nms_cpu == CUDA semantics for non-degenerate boxes
    for n, thr in ((300, 0.6), (1500, 0.1), (1500, 0.6)):
        b = rand_boxes(rng, n, extent=(60, 60, 40), smin=4, smax=24)
        s = (rng.permutation(n).astype(np.float32) + 1) / np.float32(n + 1)
        keep = nms_cpu(torch.from_numpy(b), torch.from_numpy(s), thr).numpy()
        g[f"nms_boxes_{n}_{thr}"], g[f"nms_scores_{n}_{thr}"], g[f"nms_keep_{n}_{thr}"] = b, s, keep
        eq(bx.nms(b, s, thr), keep, f"nms n={n} thr={thr}")
    b = rand_boxes(rng, 800, extent=(60, 60, 40), smin=4, smax=24)
    s = (rng.permutation(800).astype(np.float32) + 1) / np.float32(801)
    cls = rng.integers(0, 3, 800)
    keep = rb.batched_nms(torch.from_numpy(b), torch.from_numpy(s), torch.from_numpy(cls), 0.5).numpy()
    g["bnms_boxes"], g["bnms_scores"], g["bnms_cls"], g["bnms_keep"] = b, s, cls, keep
    eq(bx.batched_nms(b, s, cls, 0.5), keep, "batched_nms")
    # ---- decode / clip / small boxes
    coder = BoxCoderND(weights=(1.,) * 6)
    rel = (rng.standard_normal((a_or.shape[0], 6)) * 0.5).astype(np.float32)
    rel[7, 2] = 9.0  # exercises the exp clamp
    dec = coder.decode_single(torch.from_numpy(rel), torch.from_numpy(a_or)).numpy()
    g["dec_rel"], g["dec_boxes"] = rel, dec
    d_or = bx.decode_single(rel, a_or)
    err = np.abs(d_or - dec).max()
    assert err < 1e-4, err
    print(f"  [<=1e-4] decode_single max abs err {err:.2e} (exp is library-dependent)")
    clipped = rb.clip_boxes_to_image_(torch.from_numpy(dec.copy()), (48, 40, 24)).numpy()
    g["clip_boxes"] = clipped
    eq(bx.clip_boxes_to_image(dec, (48, 40, 24)), clipped, "clip_boxes_to_image_3d_")
    eq(bx.remove_small_boxes(clipped, 0.01), rb.remove_small_boxes(torch.from_numpy(clipped), 0.01).numpy(), "remove_small_boxes")
    np.savez_compressed(os.path.join(OUT, "boxes_golden.npz"), **g)


# ----------------------------------------------------------------------------------------------
def build_reference_net(plan):
    """RetinaUNetModule.from_config_plan's constructor calls (retinaunet/base.py:388-466) for V001."""
    from nndet.arch.conv import Generator, ConvInstanceRelu, ConvGroupRelu
    from nndet.arch.blocks.basic import StackedConvBlock2
    from nndet.arch.encoder.modular import Encoder
    from nndet.arch.decoder.base import UFPNModular
    from nndet.arch.heads.classifier import BCECLassifier
    from nndet.arch.heads.regressor import GIoURegressor
    from nndet.arch.heads.comb import DetectionHeadHNMNative
    from nndet.arch.heads.segmenter import DiCESegmenterFgBg
    from nndet.core.boxes.sampler import HardNegativeSamplerBatched
    from nndet.core.retina import BaseRetinaNet
    pa, pan, mc = plan["arch"], plan["anchors"], MODEL_CFG_V001
    coder = BoxCoderND(weights=(1.,) * 6)
    anchors = get_anchor_generator(3, s_param=True)(**pan)
    conv, hconv = Generator(ConvInstanceRelu, 3), Generator(ConvGroupRelu, 3)
    enc = Encoder(conv=conv, conv_kernels=pa["conv_kernels"], strides=pa["strides"], block_cls=StackedConvBlock2,
                  in_channels=pa["in_channels"], start_channels=pa["start_channels"], stage_kwargs=None,
                  max_channels=pa["max_channels"])
    dec = UFPNModular(conv=conv, conv_kernels=pa["conv_kernels"], strides=enc.get_strides(),
                      in_channels=enc.get_channels(), decoder_levels=pa["decoder_levels"],
                      fixed_out_channels=pa["fpn_channels"], **mc["decoder_kwargs"])
    matcher = rb.ATSSMatcher(similarity_fn=rb.box_iou, **mc["matcher_kwargs"])
    A = anchors.num_anchors_per_location()[0]
    cls = BCECLassifier(conv=hconv, in_channels=pa["fpn_channels"], internal_channels=pa["head_channels"],
                        num_classes=pa["classifier_classes"], anchors_per_pos=A,
                        num_levels=len(pa["decoder_levels"]), **mc["head_classifier_kwargs"])
    reg = GIoURegressor(conv=hconv, in_channels=pa["fpn_channels"], internal_channels=pa["head_channels"],
                        anchors_per_pos=A, num_levels=len(pa["decoder_levels"]), **mc["head_regressor_kwargs"])
    head = DetectionHeadHNMNative(classifier=cls, regressor=reg, coder=coder,
                                  sampler=HardNegativeSamplerBatched(**mc["head_sampler_kwargs"]), log_num_anchors=None)
    seg = DiCESegmenterFgBg(conv, seg_classes=pa["seg_classes"], in_channels=dec.get_channels(),
                            decoder_levels=pa["decoder_levels"], **mc["segmenter_kwargs"])
    return BaseRetinaNet(dim=3, encoder=enc, decoder=dec, head=head, anchor_generator=anchors, matcher=matcher,
                         num_classes=pa["classifier_classes"], decoder_levels=pa["decoder_levels"], segmenter=seg,
                         detections_per_img=100, score_thresh=0, topk_candidates=10000,
                         remove_small_boxes=0.01, nms_thresh=0.6)


import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("_nndet_amd_tests_gpu_util", os.path.join(ROOT, "tests", "gpu_util.py"))
_gu = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_gu)      # (the reference ships a `tests` package of its own)
det_randperm, synth_inputs = _gu.det_randperm, _gu.synth_inputs                  # shared with the parity tests; no reference imports


def golden_net(name, lite=False, batch=None, tag=None):
    """lite: store only scalars / small vectors; the inputs are regenerated from the seed by tests.gpu_util.synth_inputs.
    batch: override the plan's batch size (luna160 is generated with ONE patch: the reference needs ~10 GB per patch on CPU)."""
    plan = get_plan(name)
    if batch is not None:
        plan["batch_size"] = batch
    ref = build_reference_net(plan)
    ora = OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001)
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys()), "state-dict keys differ"
    for (k1, v1), (k2, v2) in zip(ref.state_dict().items(), ora.state_dict().items()):
        assert v1.shape == v2.shape, (k1, v1.shape, v2.shape)
    fill_state(ref); fill_state(ora)
    x, tg = synth_inputs(plan)
    orig = torch.randperm
    torch.randperm = det_randperm
    try:
        tgr = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in tg.items()}
        picks = {}
        _rsel, _osel = ref.head.select_indices, ora.select_indices
        ref.head.select_indices = lambda *a, **k: picks.setdefault("ref", _rsel(*a, **k))
        ora.select_indices = lambda *a, **k: picks.setdefault("ora", _osel(*a, **k))
        lr, pr = ref.train_step(x, tgr, evaluation=True, batch_num=0)
        sum(lr.values()).backward()
        tgo = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in tg.items()}
        lo, po = ora.train_step(x, tgo, evaluation=True)
        sum(lo.values()).backward()
        for i in (0, 1):
            assert torch.equal(picks["ref"][i].sort()[0], picks["ora"][i].sort()[0]), "oracle and reference sampled different anchors"
    finally:
        torch.randperm = orig
    g = {} if lite else {"x": x.numpy(), "target_seg": tg["target_seg"].numpy().astype(np.uint8)}
    g["x_checksum"] = np.float64(x.double().sum().item())
    # the anchors the reference's hard-negative sampler picked (comb.py:247-276; indices into the batch-concatenated anchors, ascending):
    # the negatives sit at the boundary of a top-k pool of millions of nearly equal scores, so any change of the last bit of a logit can
    # swap one -- the full-size GPU tests replay these picks and report how many their own sampler run shares (round 4)
    g["sampled_pos"] = picks["ref"][0].sort()[0].numpy().astype(np.int64)
    g["sampled_neg"] = picks["ref"][1].sort()[0].numpy().astype(np.int64)
    for i, (b, c) in enumerate(zip(tg["target_boxes"], tg["target_classes"])):
        g[f"gt_boxes_{i}"], g[f"gt_classes_{i}"] = b.numpy(), c.numpy()
    print(f"  [{name}] reference losses:", {k: float(v) for k, v in lr.items()})
    for k in lr:
        g[f"loss_{k}"] = np.float32(lr[k].item())
        assert abs(lr[k].item() - lo[k].item()) < 1e-5, (k, lr[k].item(), lo[k].item())
    gn_ref = {k: (p.grad.norm().item() if p.grad is not None else -1.0) for k, p in ref.named_parameters()}
    gn_ora = {k: (p.grad.norm().item() if p.grad is not None else -1.0) for k, p in ora.named_parameters()}
    for k in gn_ref:
        assert abs(gn_ref[k] - gn_ora[k]) <= 1e-4 * max(1.0, abs(gn_ref[k])), (k, gn_ref[k], gn_ora[k])
    g["grad_names"] = np.asarray(list(gn_ref.keys()))
    g["grad_norms"] = np.asarray(list(gn_ref.values()), np.float32)
    # a few raw gradient slices for spot checks
    for k in ("encoder.stages.0.convs.0.0.conv.weight", "head.regressor.conv_out.conv.bias",
              "decoder.up.P1.conv.weight", "segmenter.conv_out.conv.weight"):
        g["grad::" + k] = dict(ref.named_parameters())[k].grad.numpy().reshape(-1)[:512].copy()
    for b in range(x.shape[0]):
        g[f"det_boxes_{b}"] = pr["pred_boxes"][b].numpy()
        g[f"det_scores_{b}"] = pr["pred_scores"][b].numpy()
        g[f"det_labels_{b}"] = pr["pred_labels"][b].numpy()
        assert np.allclose(po["pred_boxes"][b], g[f"det_boxes_{b}"], atol=1e-4), "detections differ"
        assert np.allclose(po["pred_scores"][b], g[f"det_scores_{b}"], atol=1e-5)
    g["pred_seg_sum"] = np.float64(pr["pred_seg"].double().sum().item())
    print(f"  [{name}] oracle == reference: losses 1e-5, all {len(gn_ref)} grad norms 1e-4, detections 1e-4 "
          f"({[len(b) for b in pr['pred_boxes']]} boxes)")
    if batch is not None:
        g["batch"] = np.int64(x.shape[0])
    np.savez_compressed(os.path.join(OUT, f"net_{tag or name}_golden.npz"), **g)


def golden_fp64(name="luna160", batch=1):
    """Arbitration of the fp32 box tolerance: the SAME network evaluated in float64 (the oracle's modules
    cast with .double(); weights, input, anchors identical). For every detection the reference (fp32, oneDNN) reported in
    net_<name>_golden.npz the anchor it came from is identified in the fp64 run (same score to 1e-5, nearest box) and the fp64
    decoded + clipped box is stored: `tests/test_parity_full_gpu.py::test_luna160_fp32_boxes_arbitrated_by_fp64` then measures
    the HIP fp32 boxes AND the reference's fp32 boxes against it."""
    plan = get_plan(name)
    plan["batch_size"] = batch
    gold = np.load(os.path.join(OUT, f"net_{name}_golden.npz"))
    ora = OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001)
    fill_state(ora)
    ora = ora.double()
    x, _ = synth_inputs(plan)
    with torch.no_grad():
        pred, anchors, npl, _ = ora(x.double())
    M = anchors.shape[0]
    deltas = pred["box_deltas"].numpy().reshape(batch, M, 6)
    probs = torch.sigmoid(pred["box_logits"]).numpy().reshape(batch, M, -1)
    P = np.asarray(plan["patch_size"], np.float64)
    g = {"batch": np.int64(batch)}
    a = anchors.astype(np.float64)
    for b in range(batch):
        rb, rs = gold[f"det_boxes_{b}"].astype(np.float64), gold[f"det_scores_{b}"].astype(np.float64)
        w, h, d = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1], a[:, 5] - a[:, 4]
        cx, cy, cz = a[:, 0] + 0.5 * w, a[:, 1] + 0.5 * h, a[:, 4] + 0.5 * d
        dl = deltas[b]
        clipv = np.log(1000.0 / 16)
        pw, ph, pd = np.exp(np.minimum(dl[:, 2], clipv)) * w, np.exp(np.minimum(dl[:, 3], clipv)) * h, np.exp(np.minimum(dl[:, 5], clipv)) * d
        pcx, pcy, pcz = dl[:, 0] * w + cx, dl[:, 1] * h + cy, dl[:, 4] * d + cz
        raw = np.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph, pcz - 0.5 * pd, pcz + 0.5 * pd], 1)
        box64 = raw.copy()
        box64[:, [0, 2]] = np.clip(box64[:, [0, 2]], 0, P[0]); box64[:, [1, 3]] = np.clip(box64[:, [1, 3]], 0, P[1])
        box64[:, [4, 5]] = np.clip(box64[:, [4, 5]], 0, P[2])
        sc64 = probs[b].max(1)
        idx, out, size, err_ref = [], [], [], []
        for i in range(len(rb)):
            cand = np.nonzero(np.abs(sc64 - rs[i]) <= 1e-5)[0]
            assert len(cand) > 0, (b, i, rs[i])
            k = cand[np.abs(box64[cand] - rb[i]).max(1).argmin()]
            idx.append(k); out.append(box64[k]); err_ref.append(np.abs(box64[k] - rb[i]).max())
            size.append(max(raw[k, 2] - raw[k, 0], raw[k, 3] - raw[k, 1], raw[k, 5] - raw[k, 4]))       # decoded size BEFORE clipping
        g[f"anchor_idx_{b}"], g[f"det_boxes64_{b}"] = np.asarray(idx, np.int64), np.asarray(out, np.float64)
        g[f"det_scores64_{b}"] = sc64[np.asarray(idx)]
        g[f"ref_err_{b}"], g[f"raw_size_{b}"] = np.asarray(err_ref, np.float64), np.asarray(size, np.float64)
        print(f"  [{name} fp64] image {b}: reference fp32 boxes vs fp64: max {max(err_ref):.3e}, median {np.median(err_ref):.3e}; "
              f"max |score| diff {np.abs(g[f'det_scores64_{b}'] - rs).max():.2e}; largest decoded size {max(size):.1f}")
    np.savez_compressed(os.path.join(OUT, f"net_{name}_fp64.npz"), **g)


def golden_grad64(name, batch, tag=None, lean=False):
    """float64 arbitration of the parameter GRADIENTS (round 4): the oracle network evaluated in float64 on the fixture's inputs with
    the anchors the fp32 run sampled (which anchors the hard-negative miner picks depends on near-ties of 4.7 M scores; the picks of
    the fp32 oracle run -- identical to the reference's, that is what net_<tag>_golden.npz pins -- are replayed). Stores the 92
    gradient norms, the four gradient slices of the fp32 fixture and the losses. A heavily cancelling sum like d(gamma) of a
    full-resolution InstanceNorm (millions of sign-alternating terms) turns the 1e-5 relative differences two fp32 evaluations of a
    50-layer network have per element into 1e-3 of the sum: tests/test_parity_full_gpu.py measures the HIP fp32 gradients AND the
    reference's against this file."""
    plan = get_plan(name)
    plan["batch_size"] = batch
    x, tg = synth_inputs(plan)
    picks = {}
    orig = torch.randperm
    torch.randperm = det_randperm
    try:
        o32 = OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001)
        fill_state(o32)
        sel = o32.select_indices
        def rec(*a, **k):
            picks["idx"] = sel(*a, **k)
            return picks["idx"]
        o32.select_indices = rec
        with torch.no_grad():
            l32, _ = o32.train_step(x, {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in tg.items()})
        del o32
        o64 = OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001)
        fill_state(o64)
        o64 = o64.double()
        o64.select_indices = lambda *a, **k: picks["idx"]
        if lean:        # recompute the full-resolution stage in the backward pass instead of keeping its four 2.5 GB (batch 4) tensors
            from torch.utils.checkpoint import checkpoint
            st0 = o64.encoder.stages[0].convs
            fwd0 = st0.forward
            st0.forward = lambda inp: checkpoint(fwd0, inp, use_reentrant=False)
            # torch's float64 conv3d on the CPU unfolds the WHOLE batch into one [Cin * 27, N * D * H * W] matrix (68 GB for the
            # full-resolution 32 -> 32 layer at batch 4): run every convolution image by image
            for m in o64.modules():
                if isinstance(m, (torch.nn.Conv3d, torch.nn.ConvTranspose3d)):
                    m.forward = (lambda f: (lambda inp: torch.cat([f(inp[i:i + 1]) for i in range(inp.shape[0])], 0)))(m.forward)
        tg64 = {"target_boxes": [t.clone() for t in tg["target_boxes"]], "target_classes": [t.clone() for t in tg["target_classes"]],
                "target_seg": tg["target_seg"].clone()}
        assigned = tuple(t.double() if t.is_floating_point() else t for t in o64.assign(x.shape, tg64))
        l64, _ = o64.train_step(x.double(), tg64, assigned=assigned)
        sum(l64.values()).backward()
    finally:
        torch.randperm = orig
    g = {"batch": np.int64(batch)}
    for k in l64:
        g[f"loss_{k}"] = np.float64(l64[k].item())
        print(f"  [{name} fp64] loss {k}: fp64 {l64[k].item():.9f} fp32 oracle {l32[k].item():.9f}")
    names = [k for k, _ in o64.named_parameters()]
    g["grad_names"] = np.asarray(names)
    g["grad_norms"] = np.asarray([(p.grad.norm().item() if p.grad is not None else -1.0) for _, p in o64.named_parameters()], np.float64)
    for k in ("encoder.stages.0.convs.0.0.conv.weight", "head.regressor.conv_out.conv.bias",
              "decoder.up.P1.conv.weight", "segmenter.conv_out.conv.weight"):
        g["grad::" + k] = dict(o64.named_parameters())[k].grad.numpy().reshape(-1)[:512].copy()
    ref = np.load(os.path.join(OUT, f"net_{tag or name}_golden.npz"))
    dev = np.abs(ref["grad_norms"].astype(np.float64) - g["grad_norms"]) / np.maximum(np.abs(g["grad_norms"]), 1e-12)
    worst = np.argsort(-dev)[:5]
    print(f"  [{name} fp64] reference fp32 gradient norms vs fp64: " + ", ".join(f"{names[i]} {dev[i]:.2e}" for i in worst))
    np.savez_compressed(os.path.join(OUT, f"net_{tag or name}_grad64.npz"), **g)


def golden_postproc():
    """Direct fixtures of three reference functions that were pinned only through end-to-end runs:
      * BaseRetinaNet.postprocess_detections_single_image (nndet/core/retina.py:332-379) on decoded boxes + probabilities,
        1 and 3 classes, with small boxes, out-of-image boxes and a score threshold;
      * HardNegativeSamplerBatched.__call__ (nndet/core/boxes/sampler.py:237-270) with torch.randperm := reversed arange;
      * BoxEnsemblerSelective.postprocess_image (nndet/inference/ensembler/detection.py:166-217) with batched_nms_model.
    Scores are distinct (ties are implementation-defined in the reference)."""
    from types import SimpleNamespace
    from nndet.core.retina import BaseRetinaNet
    from nndet.core.boxes.sampler import HardNegativeSamplerBatched
    rng = np.random.default_rng(11)
    g = {}
    shape = (96, 80, 64)
    for tag, C, M, topk, thr, dets in (("c1", 1, 6000, 2000, 0.05, 100), ("c3", 3, 3000, 1500, 0.0, 50)):
        boxes = rand_boxes(rng, M, extent=shape, smin=0.0, smax=30)
        boxes[:40] += 70                                                   # partly / fully outside the image -> clipped to (near) zero size
        boxes[40:60, 2] = boxes[40:60, 0] + 0.005                          # thinner than remove_small_boxes
        probs = ((rng.permutation(M * C).astype(np.float64) + 1) / (M * C + 1)).astype(np.float32).reshape(M, C)
        holder = SimpleNamespace(topk_candidates=topk, score_thresh=thr, num_foreground_classes=C, remove_small_boxes=0.01,
                                 nms_thresh=0.6, detections_per_img=dets)
        rbx, rp, rl = BaseRetinaNet.postprocess_detections_single_image(holder, torch.from_numpy(boxes.copy()), torch.from_numpy(probs.copy()), shape)
        ob, op, ol = bx.postprocess_single_image(boxes.copy(), probs, shape, C, topk, thr, 0.01, 0.6, dets)
        eq(ob, rbx.numpy(), f"postprocess_single_image boxes {tag}"); eq(op, rp.numpy(), f"postprocess_single_image scores {tag}")
        eq(ol, rl.numpy(), f"postprocess_single_image labels {tag}")
        g[f"pp_{tag}_boxes"], g[f"pp_{tag}_probs"] = boxes, probs
        g[f"pp_{tag}_cfg"] = np.asarray([C, topk, dets, *shape], np.int64); g[f"pp_{tag}_thr"] = np.float32(thr)
        g[f"pp_{tag}_out_boxes"], g[f"pp_{tag}_out_scores"], g[f"pp_{tag}_out_labels"] = rbx.numpy(), rp.numpy(), rl.numpy()
    # ---- sampler
    n = 40000
    labels = np.zeros(n, np.float32)
    labels[rng.choice(n, 300, replace=False)] = 1.0
    labels[rng.choice(n, 500, replace=False)] = -1.0                      # "between thresholds": ignored by the sampler
    fg = ((rng.permutation(n).astype(np.float64) + 1) / (n + 1)).astype(np.float32)
    per_img = [10000, 14000, 16000]
    sampler = HardNegativeSamplerBatched(batch_size_per_image=32, positive_fraction=0.33, min_neg=1, pool_size=20)
    orig = torch.randperm
    torch.randperm = det_randperm
    try:
        pos_m, neg_m = sampler(list(torch.from_numpy(labels).split(per_img)), torch.from_numpy(fg))
    finally:
        torch.randperm = orig
    pos = torch.where(torch.cat(pos_m))[0].numpy(); neg = torch.where(torch.cat(neg_m))[0].numpy()
    op_, on_, _ = bx.hnm_select_reversed(labels, fg, len(per_img), 32, 0.33, 1, 20)
    eq(op_, pos, "HardNegativeSamplerBatched positives"); eq(on_, neg, "HardNegativeSamplerBatched negatives")
    g["hnm_labels"], g["hnm_fg"], g["hnm_per_img"] = labels, fg, np.asarray(per_img, np.int64)
    g["hnm_pos"], g["hnm_neg"] = pos.astype(np.int64), neg.astype(np.int64)
    # ---- ensembler stage
    from oracle.refimport import install_stub_finder
    install_stub_finder()
    from nndet.inference.ensembler.detection import BoxEnsemblerSelective
    from nndet.inference.detection.model import batched_nms_model
    N = 5000
    eb = rand_boxes(rng, N, extent=(128, 128, 96), smin=0.0, smax=28)
    eb[:30] += 100
    ep = ((rng.permutation(N).astype(np.float64) + 1) / (N + 1)).astype(np.float32)
    el = rng.integers(0, 3, N).astype(np.int64)
    ew = rng.uniform(0.1, 1.0, N).astype(np.float32)
    prm = {"model_iou": 0.1, "model_nms_fn": batched_nms_model, "model_score_thresh": 0.1, "model_topk": 1000,
           "model_detections_per_image": 100, "remove_small_boxes": 1e-2}
    holder = SimpleNamespace(parameters=prm)
    tile = (128, 128, 96)
    rb_, rp_, rl_, rw_ = BoxEnsemblerSelective.postprocess_image(holder, torch.from_numpy(eb.copy()), torch.from_numpy(ep.copy()),
                                                                 torch.from_numpy(el.copy()), torch.from_numpy(ew.copy()), tile)
    g["ens_boxes"], g["ens_probs"], g["ens_labels"], g["ens_weights"] = eb, ep, el, ew
    g["ens_shape"] = np.asarray(tile, np.int64)
    g["ens_out_boxes"], g["ens_out_probs"], g["ens_out_labels"], g["ens_out_weights"] = rb_.numpy(), rp_.numpy(), rl_.numpy(), rw_.numpy()
    print(f"  ensembler postprocess_image: {len(rb_)} of {N} rows kept; postprocess_single_image c1 / c3: "
          f"{len(g['pp_c1_out_boxes'])} / {len(g['pp_c3_out_boxes'])}; sampler: {len(pos)} pos / {len(neg)} neg")
    np.savez_compressed(os.path.join(OUT, "postproc_golden.npz"), **g)


def golden_targets():
    """the three `pre_trafo` transforms of RetinaUNetModule (retinaunet/base.py:108-131) run unmodified on a small
    synthetic instance volume; the oracle restatement must reproduce them exactly (integer work)."""
    from oracle.refimport import install_stub_finder
    install_stub_finder()                                  # nndet.io imports SimpleITK / batchgenerators at module level
    from nndet.io.transforms import Compose, FindInstances, Instances2Boxes, Instances2Segmentation
    rng = np.random.default_rng(42)
    B, D, H, W = 3, 24, 20, 16
    tgt = np.zeros((B, 1, D, H, W), np.float32)
    maps = [{"1": 0, "2": 1, "5": 0, "9": 2}, {"3": 1}, {"1": 0, "4": 1, "7": 1}]      # incl. ids that are absent from the patch
    def blob(b, i, lo, hi):
        tgt[b, 0, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = i
    blob(0, 1, (2, 3, 1), (7, 9, 5)); blob(0, 2, (10, 0, 8), (24, 4, 16)); blob(0, 5, (5, 12, 2), (6, 13, 3))
    blob(0, 2, (0, 18, 0), (2, 20, 2))                     # a second, disconnected part of instance 2 (box spans both)
    blob(2, 7, (0, 0, 0), (24, 20, 16)); blob(2, 4, (11, 9, 7), (13, 11, 9)); blob(2, 1, (20, 15, 3), (23, 19, 12))
    noise = rng.integers(0, 40, (D, H, W)) == 0
    tgt[2, 0][noise] = 4                                   # scattered single voxels of instance 4
    trafo = Compose(FindInstances(instance_key="target", save_key="present_instances"),
                    Instances2Boxes(instance_key="target", map_key="instance_mapping", box_key="boxes", class_key="classes",
                                    present_instances="present_instances"),
                    Instances2Segmentation(instance_key="target", map_key="instance_mapping", present_instances="present_instances"))
    with torch.no_grad():
        out = trafo(data=torch.zeros(B, 1, D, H, W), target=torch.from_numpy(tgt.copy()), instance_mapping=maps)
    g = {"target": tgt.astype(np.uint8), "maps": np.asarray([repr(m) for m in maps])}
    for b in range(B):
        rb, rc = out["boxes"][b].numpy(), out["classes"][b].numpy()
        ob, oc, oi, osem = bx.instances_to_targets(tgt[b, 0], maps[b])
        if rb.size == 0:
            assert ob.shape[0] == 0 and rc.size == 0, "image without instances"
            rb, rc = np.zeros((0, 6), np.float32), np.zeros((0,), np.int64)
        eq(ob, rb, f"target boxes image {b}"); eq(oc, rc.astype(np.int64), f"target classes image {b}")
        eq(oi, out["present_instances"][b].numpy().astype(np.int32), f"present instances image {b}")
        eq(osem, out["target"][b, 0].numpy(), f"semantic map image {b}")
        g[f"boxes_{b}"], g[f"classes_{b}"], g[f"ids_{b}"] = rb, rc.astype(np.int64), oi
        g[f"seg_{b}"] = out["target"][b, 0].numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "targets_golden.npz"), **g)


def golden_wbc():
    """the reference's weighted box clustering (nndet/inference/detection/wbc.py, pure torch) on clustered
    synthetic predictions of 3 "models"; the oracle must agree to 1e-5 (torch's reduction order is its own)."""
    from oracle.refimport import install_stub_finder
    install_stub_finder()                                  # nndet.inference imports nndet.io (SimpleITK ...) at package level
    from nndet.inference.detection.wbc import batched_wbc as ref_batched_wbc, wbc as ref_wbc
    rng = np.random.default_rng(7)
    centers = rng.uniform(10, 150, (60, 3)); sizes = rng.uniform(6, 30, (60, 3))
    boxes, scores, labels, weights, nexp = [], [], [], [], []
    for c, sz in zip(centers, sizes):
        lab = int(rng.integers(0, 3))
        for _ in range(int(rng.integers(1, 7))):                       # 1..6 jittered predictions of the same object
            cc, ss = c + rng.normal(0, 1.0, 3), sz * rng.uniform(0.9, 1.1, 3)
            lo, hi = cc - ss / 2, cc + ss / 2
            boxes.append([lo[0], lo[1], hi[0], hi[1], lo[2], hi[2]]); scores.append(rng.uniform(0.05, 0.99))
            labels.append(lab); weights.append(rng.uniform(0.2, 1.0)); nexp.append(float(rng.integers(2, 7)))
    b = np.asarray(boxes, np.float32); s = np.asarray(scores, np.float32); l = np.asarray(labels, np.int64)
    w = np.asarray(weights, np.float32); ne = np.asarray(nexp, np.float32)
    g = {"boxes": b, "scores": s, "labels": l, "weights": w, "n_exp": ne}
    for tag, kw in (("a", dict(iou_thresh=0.3, score_thresh=0.0, use_area=False, missing_weight=1.0)),
                    ("b", dict(iou_thresh=0.1, score_thresh=0.2, use_area=True, missing_weight=0.5))):
        rb, rs, rl = ref_batched_wbc(torch.from_numpy(b), torch.from_numpy(s), torch.from_numpy(l), torch.from_numpy(w),
                                     kw["iou_thresh"], torch.from_numpy(ne), kw["score_thresh"], use_area=kw["use_area"],
                                     missing_weight=kw["missing_weight"])
        ob, os_, ol = bx.batched_wbc(b, s, l, w, kw["iou_thresh"], ne, kw["score_thresh"], kw["use_area"], kw["missing_weight"])
        assert rb.shape == ob.shape, (rb.shape, ob.shape)
        assert np.allclose(ob, rb.numpy(), rtol=1e-5, atol=1e-5) and np.allclose(os_, rs.numpy(), rtol=1e-5, atol=1e-6)
        assert np.array_equal(ol, rl.numpy())
        print(f"  [1e-5] batched_wbc {tag}: {len(b)} predictions -> {len(ob)} clusters")
        g[f"out_boxes_{tag}"], g[f"out_scores_{tag}"], g[f"out_labels_{tag}"] = rb.numpy(), rs.numpy(), rl.numpy()
    m = l == 1
    rb, rs = ref_wbc(torch.from_numpy(b[m]), torch.from_numpy(s[m]), torch.from_numpy(w[m]), torch.from_numpy(ne[m]), 0.2, 0.1)
    ob, os_ = bx.wbc(b[m], s[m], w[m], ne[m], 0.2, 0.1)
    assert np.allclose(ob, rb.numpy(), rtol=1e-5, atol=1e-5) and np.allclose(os_, rs.numpy(), rtol=1e-5, atol=1e-6)
    g["one_boxes"], g["one_scores"] = rb.numpy(), rs.numpy()
    np.savez_compressed(os.path.join(OUT, "wbc_golden.npz"), **g)


def golden_giou_grad():
    """Gradient of the reference's `generalized_box_iou` (an autograd expression, ops.py:106-128,162-185) w.r.t. BOTH box sets on the
    [37 x 501] fixture with a random cotangent, and of `GIoULoss` (losses/regression.py:118-162) for N = M in {1, 42, 300}; the torch
    restatement `oracle/boxes_torch.py` is checked against it here (fp32: same torch ops, so to round-off of the summation order)."""
    from nndet.core.boxes.ops import generalized_box_iou as ref_giou
    from nndet.losses.regression import GIoULoss
    from oracle import boxes_torch as bt
    rng = np.random.default_rng(7)
    g = {}
    b1, b2 = rand_boxes(rng, 37), rand_boxes(rng, 501)
    b2[5] = b1[3]                                   # an identical pair: every min / max of it is a tie (0.5 / 0.5 sub-gradients)
    b2[6, :] = b1[4, :]; b2[6, 2] += 1.0            # ties on five of six coordinates
    cot = rng.standard_normal((37, 501)).astype(np.float32)
    for eps, tag in ((0.0, ""), (1e-7, "_eps")):
        t1, t2 = torch.from_numpy(b1).requires_grad_(), torch.from_numpy(b2).requires_grad_()
        m = ref_giou(t1, t2, eps=eps)
        m.backward(torch.from_numpy(cot))
        o1, o2 = torch.from_numpy(b1).requires_grad_(), torch.from_numpy(b2).requires_grad_()
        mo = bt.generalized_box_iou(o1, o2, eps=eps)
        mo.backward(torch.from_numpy(cot))
        eq(mo.detach().numpy(), m.detach().numpy(), f"giou matrix (torch oracle){tag}")
        for a, b, what in ((o1.grad, t1.grad, "d boxes1"), (o2.grad, t2.grad, "d boxes2")):
            err = (a - b).abs().max().item() / b.abs().max().item()
            assert err < 1e-6, (what, err)
            print(f"  [{err:.1e}] giou pairwise gradient {what}{tag} (torch oracle vs reference autograd)")
        g[f"pw_ga{tag}"], g[f"pw_gb{tag}"] = t1.grad.numpy(), t2.grad.numpy()
    g["pw_b1"], g["pw_b2"], g["pw_cot"] = b1, b2, cot
    for n in (1, 42, 300):
        tgt = rand_boxes(rng, n)
        pred = (tgt + rng.uniform(-3, 3, tgt.shape)).astype(np.float32)
        pred[:, [2, 3, 5]] = np.maximum(pred[:, [2, 3, 5]], pred[:, [0, 1, 4]] + 0.5)
        if n > 2:
            pred[1] = tgt[1]                        # a perfect prediction
        for red in ("sum", "mean"):
            tp = torch.from_numpy(pred).requires_grad_()
            loss = GIoULoss(reduction=red, eps=1e-7, loss_weight=2.0)(tp, torch.from_numpy(tgt))
            loss.backward()
            op = torch.from_numpy(pred).requires_grad_()
            lo = bt.giou_loss(op, torch.from_numpy(tgt), eps=1e-7, reduction=red, loss_weight=2.0)
            lo.backward()
            assert abs(lo.item() - loss.item()) <= 1e-6 * max(1.0, abs(loss.item())), (n, red, lo.item(), loss.item())
            err = (op.grad - tp.grad).abs().max().item() / max(tp.grad.abs().max().item(), 1e-30)
            assert err < 1e-6, (n, red, err)
            print(f"  [{err:.1e}] GIoULoss n = {n} reduction = {red}: loss {loss.item():.6f} (torch oracle vs reference)")
            g[f"loss{n}_{red}"], g[f"loss{n}_{red}_grad"] = np.float32(loss.item()), tp.grad.numpy()
        g[f"loss{n}_pred"], g[f"loss{n}_tgt"] = pred, tgt
    np.savez_compressed(os.path.join(OUT, "giou_grad_golden.npz"), **g)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["boxes", "giou_grad", "targets", "wbc", "postproc", "tiny", "toy64", "luna160", "luna160_fp64"]
    if "wbc" in which:
        print("weighted box clustering:"); golden_wbc()
    if "targets" in which:
        print("target preparation:"); golden_targets()
    if "boxes" in which:
        print("box ops:"); golden_boxes()
    if "giou_grad" in which:
        print("GIoU gradients:"); golden_giou_grad()
    print("network:")
    if "tiny" in which:
        golden_net("tiny")
    if "toy64" in which:
        golden_net("toy64", lite=True)        # BASELINE.json configs[0]: the reference's own CPU-runnable case
    if "luna160" in which:
        golden_net("luna160", lite=True, batch=1)   # BASELINE.json configs[1] (the benchmarked plan), one 160x160x96 patch, fp32
    if "lidc192" in which:
        golden_net("lidc192", lite=True, batch=1)   # BASELINE.json configs[3] (192x192x128), one patch, fp32 (round 4)
    if "luna160_b4" in which:
        # BASELINE.json configs[1] at the BENCHMARKED batch (c002.py:52): the batch-level hard-negative mining (sampler.py:237-270) and
        # batch_dice couple the four patches (round 4; ~40 GB peak: the two models run one after the other)
        golden_net("luna160", lite=True, batch=4, tag="luna160_b4")
    if "lidc192_b2" in which:
        # BASELINE.json configs[3] beyond one patch (round 5): two 192x192x128 patches couple through the batch-level hard-negative mining
        # and batch_dice, and every kernel walks 2 x 1.9 x the tiles of luna160 (~40 GB peak on the CPU)
        golden_net("lidc192", lite=True, batch=2, tag="lidc192_b2")
    if "lidc192_grad64" in which:
        golden_grad64("lidc192", 1)
    if "lidc192_b2_grad64" in which:
        golden_grad64("lidc192", 2, tag="lidc192_b2", lean=True)
    if "luna160_grad64" in which:
        golden_grad64("luna160", 1)
    if "luna160_b4_grad64" in which:
        golden_grad64("luna160", 4, tag="luna160_b4", lean=True)
    if "postproc" in which:
        golden_postproc()
    if "luna160_fp64" in which:
        golden_fp64("luna160", 1)
