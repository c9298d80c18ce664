#!/usr/bin/env python3
"""Bitwise comparison of nd_pack_conv3x3_wino4_weight between two builds of the library (a rewrite of the pack kernel must not change a bit of the
packed weights): python tools/pack_compare.py tools/_build/lib_head_pack.so noisediff_amd/libnoisediff_hip.so.  Each library in its own process."""
import os, subprocess, sys, hashlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(64, 64), (128, 64), (72, 40), (24, 8), (512, 512), (1536, 512), (20, 2048)]


def worker(lib):
    sys.path.insert(0, REPO)
    import ctypes as C, time
    import torch
    handle = C.CDLL(lib)                     # bare ctypes: an older build lacks symbols _lib.load() insists on
    handle.nd_pack_conv3x3_wino4_weight_floats.restype = C.c_int64
    handle.nd_pack_conv3x3_wino4_weight.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    torch.zeros(1, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    class L:
        @staticmethod
        def call(name, *a):
            assert getattr(handle, name)(*a) == 0
    for cin, cout in SHAPES:
        g = torch.Generator().manual_seed(cin * 7919 + cout)
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).cuda()
        n = int(handle.nd_pack_conv3x3_wino4_weight_floats(cin, cout))
        out = torch.full((n,), float("nan"), device="cuda")
        L.call("nd_pack_conv3x3_wino4_weight", w.data_ptr(), out.data_ptr(), cin, cout, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            L.call("nd_pack_conv3x3_wino4_weight", w.data_ptr(), out.data_ptr(), cin, cout, st)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 10 * 1e6
        print(f"PACK {cin} {cout} {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]} {us:.1f}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(sys.argv[2])
    else:
        res = []
        for lib in sys.argv[1:3]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", os.path.abspath(lib)], capture_output=True, text=True, timeout=300)
            res.append([ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith("PACK ")])
            if not res[-1]:
                print(r.stdout[-2000:], r.stderr[-2000:])
        same = True
        for a, b in zip(*res):
            ok = a[2] == b[2]
            same &= ok
            print(f"cin {a[0]:>5s} cout {a[1]:>5s}: {a[2]} / {b[2]} {'same bits' if ok else 'DIFFERENT'}   {a[3]} us -> {b[3]} us")
        print("all packings bit-identical" if same and res[0] else "MISMATCH")
        sys.exit(0 if same and res[0] else 1)
