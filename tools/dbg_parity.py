import sys, torch
sys.path.insert(0, "/root/repo")
from tests.golden_util import load_golden, inputs_of
from tests.test_net_parity import build_net
from tunevlseg_amd import ops
name = sys.argv[1]
fx = load_golden(name)
net = build_net(fx)
pix, ids, am, mask = (t.cuda() for t in inputs_of(fx))
logits = net(text_input={"input_ids": ids, "attention_mask": am}, image_input=pix)
loss, isum = ops.DiceCELossFn.apply(logits, mask, 1.0, 0.2, 0.5)
loss.backward()
for k, p in net.named_parameters():
    if not p.requires_grad or k in fx["meta"]["grads_none"]:
        continue
    g_ref = torch.from_numpy(fx["grad." + k])
    scale = g_ref.abs().max().item() + 1e-12
    d = (p.grad.cpu() - g_ref)
    gerr = d.abs().max().item()
    # relative L2 too
    if k.endswith("context_vectors") and "grad64." + k in fx:
        t64 = torch.from_numpy(fx["grad64." + k])
        for dpt in range(g_ref.shape[0]):
            e = (p.grad.cpu()[dpt] - t64[dpt])
            print(f"   depth {dpt}: |g|max {t64[dpt].abs().max():.3e} HIP-vs-f64 maxerr/|g|max {e.abs().max()/t64[dpt].abs().max():.2e}; worst row {int(e.abs().max(1).values.argmax())}; ref32-vs-f64 {(g_ref[dpt]-t64[dpt]).abs().max()/t64[dpt].abs().max():.2e}")
    print(f"{k:60s} shape {tuple(g_ref.shape)} max|g| {scale:.3e} maxerr/scale {gerr/scale:.2e} relL2 {(d.norm()/g_ref.norm()).item():.2e}")
