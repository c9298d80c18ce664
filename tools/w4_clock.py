"""Diagnostic: effective shader clock and cycle split of conv3x3_wino4 (needs a -DW4_STAMP side build, ND_LIB).  W4_MODE=aff: the GroupNorm-affine + SiLU prologue; map / gen: + the per-pixel maps, read or formed in the kernel."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.zeros(1, device="cuda")
from noisediff_amd import _lib as L
L.load(os.environ["ND_LIB"])
import hiputil as hu
ctx = hu.Ctx()
ENTRY = os.environ.get("W4_ENTRY", "nd_conv3x3_wino4_nhwc_f32")     # nd_conv3x3_wino4_16_nhwc_f32: 16 x 16-pixel regions, two workgroups per CU
MFMA_CHUNK = 4608 if "_16_" in ENTRY else 9216      # MFMA issue cycles per wave and 16-channel chunk
PACK = "nd_pack_conv3x3_wino4_weight"
SHAPES = [(16, 256, 256, 64, 64), (16, 32, 32, 512, 512)]
if os.environ.get("W4_SHAPES"):          # e.g. W4_SHAPES="1,256,256,64,64;2,256,256,64,64"
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["W4_SHAPES"].split(";")]
for (B, H, W, cin, cout) in SHAPES:
    x = torch.randn(B, H, W, cin, device=hu.DEV); w = torch.randn(cout, cin, 3, 3) * 0.05
    wd = hu.dev(w); wp = torch.empty(ctx.lib.nd_pack_conv3x3_wino4_weight_floats(cin, cout), device=hu.DEV)
    L.call(PACK, wd.data_ptr(), wp.data_ptr(), cin, cout, ctx.stream); ctx.sync()
    out = torch.empty(B, H, W, cout, device=hu.DEV)
    dbg = torch.zeros(16 * 1024, dtype=torch.int64, device=hu.DEV)
    st = torch.zeros(B * ctx.lib.nd_conv3x3_wino4_stat_slots(H, W) * cout * 2, device=hu.DEV)
    torch.cuda.synchronize()
    if os.environ.get("W4_MODE") == "aff":
        mad = torch.randn(B, 3, cin, device=hu.DEV); mad[:, 1] = mad[:, 1].abs() + 0.5
        src = hu.src(x, None, L.PRO_AFFINE_SILU, mad=mad)
    elif os.environ.get("W4_MODE") in ("map", "gen"):           # ResnetBlock2's per-pixel scale / shift: maps read (blocked layout) or formed in the kernel
        mad = torch.randn(B, 3, cin, device=hu.DEV); mad[:, 1] = mad[:, 1].abs() + 0.5
        if os.environ["W4_MODE"] == "map":
            src = hu.src(x, None, L.PRO_AFFINE_MAP_SILU, mad=mad, map=torch.randn(B, H, W, 2 * cin, device=hu.DEV) * 0.1, map_blocked=1)
        else:
            src = hu.src(x, None, L.PRO_AFFINE_GENMAP_SILU, mad=mad, map=torch.randn(B, H, W, 8, device=hu.DEV), gamma=torch.randn(2 * cin, 8, device=hu.DEV) * 0.1,
                         beta=torch.randn(2 * cin, device=hu.DEV) * 0.1)
    else:
        src = hu.src(x)
    d = L.Conv3x3(); d.src, d.weight, d.out = src, wp.data_ptr(), out.data_ptr()
    d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
    d.slot_count = dbg.data_ptr(); d.stats = st.data_ptr()
    for _ in range(3):
        L.call(ENTRY, C.byref(d), ctx.stream); ctx.sync()
    v = dbg.cpu().view(1024, 16).double()
    v = v[v[:, 2] > 0]                                  # workgroups that had work (small problems start fewer than 256)
    print(f"{os.environ.get('W4_MODE', 'plain')}: {v.shape[0]} workgroups;", end=" ")
    cyc, real, chunks, epi, xf, second, first, last, third, wait, pro, top, stile = (v[:, i] for i in range(13))
    n_chunks = (cin + 15) // 16
    mhz = cyc / (real / 100.0)
    per = lambda t: float((t / chunks).mean())
    tiles = chunks / n_chunks
    pt = lambda t: float((t / tiles).mean())
    print((B, H, W, cin, cout), f"clock {mhz.mean():.0f} MHz (min {mhz.min():.0f} max {mhz.max():.0f}); wall/WG {float(real.mean()) / 100:.1f} us, {float(tiles.mean()):.0f} tiles x {n_chunks} chunks;",
          f"per tile: {pt(cyc):.0f} cycles (MFMA {MFMA_CHUNK * n_chunks}) = stages of chunk 0 {pt(first):.0f}, chunk 1 {pt(second):.0f}, chunk 2 {pt(third):.0f}, last {pt(last):.0f},",
          f"all chunks {pt(cyc - epi - xf - wait - pro):.0f}; barrier waits {pt(wait):.0f}, transforms {pt(xf):.0f}, epilogue {pt(epi):.0f}; loop top {pt(top):.0f}, stage_tile {pt(stile):.0f}; per WG: stagger + prologue {float(pro.mean()):.0f}", flush=True)
