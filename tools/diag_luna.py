import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.detweights import fill_state
from oracle.retina_torch import OracleRetinaUNet
from nndetection_amd.plans import get_plan, MODEL_CFG_V001
from nndetection_amd.ptmodule import build_model
from tests.gpu_util import det_randperm, synth_inputs
name = sys.argv[1] if len(sys.argv) > 1 else "luna160"
gn = np.load(f"tests/golden/net_{name}_golden.npz")
plan = get_plan(name)
if "batch" in gn: plan["batch_size"] = int(gn["batch"])
x, tg = synth_inputs(plan)
ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
net = build_model(plan); net.load_state_dict(ora.state_dict()); net.cuda()
torch.randperm = det_randperm
tgc = {"target_boxes": [b.cuda() for b in tg["target_boxes"]], "target_classes": [c.cuda() for c in tg["target_classes"]], "target_seg": tg["target_seg"].cuda()}
losses, pred = net.train_step(x.cuda(), tgc, evaluation=True)
print({k: float(v) for k, v in losses.items()})
for b in range(plan["batch_size"]):
    pb, ps = pred["pred_boxes"][b].cpu().numpy(), pred["pred_scores"][b].cpu().numpy()
    rb, rs = gn[f"det_boxes_{b}"], gn[f"det_scores_{b}"]
    d = np.abs(pb - rb).max(1)
    print("image", b, "rows", len(pb), len(rb), "rows with err>1e-3:", np.nonzero(d > 1e-3)[0][:20], "max err on good rows", d[d <= 1e-3].max())
    print(" score max diff", np.abs(ps - rs).max())
    bad = np.nonzero(d > 1e-3)[0]
    for i in bad[:6]:
        print("  row", i, "got", pb[i], ps[i], "ref", rb[i], rs[i])
    # relative error on good rows
    good = d <= 1e-3
    rel = (np.abs(pb - rb)[good] / np.maximum(1.0, np.abs(rb[good]))).max()
    print(" max rel err good rows", rel, "coord at max abs", rb[good].reshape(-1)[np.abs(pb - rb)[good].reshape(-1).argmax()])
    # set match
    from oracle import boxes_np as bx
    iou = bx.box_iou(rb, pb)
    print(" best-match IoU min", iou.max(1).min(), "n ref rows without a >0.999 match", int((iou.max(1) < 0.999).sum()))
