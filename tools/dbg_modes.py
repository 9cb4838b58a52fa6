import sys, torch
sys.path.insert(0, "/root/repo")
from tests.golden_util import load_golden, inputs_of
from tests.test_net_parity import build_net
from tunevlseg_amd import ops, hip
name = sys.argv[1]
fx = load_golden(name)
pix, ids, am, mask = (t.cuda() for t in inputs_of(fx))
def run(mode, tp3):
    hip.set_gemm_mode(mode); hip.TP3_MIN_ROWS = 1 if tp3 else 10**9
    net = build_net(fx)
    logits = net(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
    loss, _ = ops.DiceCELossFn.apply(logits, mask, 1.0, 0.2, 0.5)
    loss.backward()
    return logits.detach().cpu().double(), net.context_learner.context_vectors.grad.cpu().double()
res = {"old_bf16x6": run("bf16x6", False), "old_f32": run("f32", False), "tp3": run("bf16x6", True)}
hip.SPLITK = False
res["old_bf16x6_nosplitk"] = run("bf16x6", False)
hip.SPLITK = True
k = "context_learner.context_vectors"
res["ref32"] = (torch.from_numpy(fx["out.logits"]).double(), torch.from_numpy(fx["grad." + k]).double())
res["ref64"] = (None, torch.from_numpy(fx["grad64." + k]).double())
names = list(res)
for i, a in enumerate(names):
    for b in names[i + 1:]:
        ga, gb = res[a][1], res[b][1]
        s = f"{a:18s} vs {b:18s} grad relL2 {((ga-gb).norm()/gb.norm()).item():.3e}"
        if res[a][0] is not None and res[b][0] is not None:
            s += f"  logits maxdiff {(res[a][0]-res[b][0]).abs().max().item():.3e}"
        print(s)
