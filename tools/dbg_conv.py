import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch, torch.nn.functional as F
import hiputil as hu
from noisediff_amd import synth
ctx = hu.Ctx()
def U(name, shape, lo=-1.0, hi=1.0): return synth.uniform(11, name, shape, lo, hi)
for (B,H,W,cin,cout) in [(16,56,72,16,64),(16,64,64,16,64),(16,64,64,32,64),(16,64,64,64,64),(8,64,64,32,128),(8,64,64,32,256),(32,64,64,8,64),(16,64,64,16,128)]:
    x = U("x", (B, cin, H, W), -1.5, 1.5); w = U("w", (cout, cin, 3, 3), -0.2, 0.2); b = U("b", (cout,))
    ref = F.conv2d(x, w, b, padding=1)
    out, st, sc, slots = hu.conv3x3(ctx, hu.src(hu.nhwc(x)), hu.pack_conv3(ctx, w), hu.dev(b), B, H, W, cin, cout, stats=True)
    got = hu.nchw(out)
    nan = torch.isnan(got)
    err = (torch.nan_to_num(got) - ref).abs().max().item()
    print((B,H,W,cin,cout), "slots", slots, "nan frac", nan.float().mean().item(), "err(non-nan)", err, "stats nan", torch.isnan(st).float().mean().item(), flush=True)
