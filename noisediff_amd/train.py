"""Training on MI355X, first slice: the 3x3 convolutions of a differentiable network -- 85 % of the FLOPs of
``GaussianDiffusion.p_losses`` (models/denoising_diffusion_pytorch.py:481-531) -- forward AND backward on the HIP library.

    model = <the reference NoiseDiffNet, or any nn.Module>            # differentiable PyTorch
    noisediff_amd.train.accelerate(model)                             # every nn.Conv2d(3x3, padding 1, stride 1) -> HIP
    loss = noisediff_amd.GaussianDiffusion(model, ...)(img, condition); loss.backward()

``Conv3x3Function`` is a ``torch.autograd.Function`` over the C ABI:
  * forward      nd_conv3x3_{wino4, wino2, direct}_nhwc_f32 (the sampling path's kernels, same selection rule as the engine)
  * grad input   the SAME forward kernels on weights packed from ``w.flip(2, 3).transpose(0, 1)`` -- the data gradient of a
                 stride-1 "same" convolution is that convolution with the taps flipped and the channel roles swapped
                 (nd_pack_conv3x3_*_weight_dgrad read the forward weight that way in place)
  * grad weight  nd_conv3x3_wgrad_nhwc_f32 (conv3x3_wgrad.hip: nine tap GEMMs over the pixels on the fp32 MFMA, fixed
                 summation order, no atomics)
  * grad bias    falls out of the weight-gradient kernel's staged dY tiles
Tensors stay what PyTorch hands over: NCHW-shaped, ``channels_last`` in memory (= the library's NHWC; other layouts are
converted once per call), fp32, on the caller's CUDA stream -- so autograd's stream ordering holds without synchronisation.
``GroupNormFunction`` (second slice) does the same for nn.GroupNorm: nd_groupnorm_train_forward_f32 / _backward_f32 (norm_train.hip) stream
the channels_last tensors once per pass, where PyTorch's NCHW group norm first copies them to NCHW and back -- on the full-resolution
blocks the norms cost more than the convolutions before this.  ``LinearFunction`` (third slice) keeps the output and the data
gradient of token Linears / 1x1 convolutions on the library GEMM and takes their weight and bias gradient -- a reduction over up to
10^6 tokens that library GEMMs run at a tenth of HBM speed -- to nd_linear_wgrad_f32 (linear_wgrad.hip); ``LayerNormFunction``:
nn.LayerNorm over token channels forward and backward (norm_train.hip).  Activations and attention stay on PyTorch's own ROCm kernels for now.

There is no fallback: a CPU tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch.autograd.function import once_differentiable
from torch import nn

from . import _lib as L


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    """NCHW-shaped tensor whose memory is NHWC (channels_last), fp32, contiguous in that format."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous(memory_format=torch.channels_last)


class _Nop:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOP = _Nop()


def _on(device: torch.device):
    """The library launches on the CURRENT device: switch only when the tensors live elsewhere (the context manager costs ~10 us of
    host time per call, and a training step makes several hundred calls)."""
    return _NOP if device.index == torch.cuda.current_device() else torch.cuda.device(device)


def _stream(device: Optional[torch.device] = None) -> C.c_void_p:
    """torch's current stream OF THE TENSORS' DEVICE (not of the current device: a caller may sit on another GPU)."""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))         # (torch.cuda.current_stream(...).cuda_stream builds a Stream object: 5 us, x 1800 per step)


_SPLIT_K = __import__("os").environ.get("ND_TRAIN_SPLITK", "1") != "0"      # A/B knob (tools/): 0 = never the split-K form of conv3x3_wino4


def _conv3x3_nhwc(x: torch.Tensor, w_oihw: torch.Tensor, bias: Optional[torch.Tensor], dgrad: bool = False, stats: bool = False,
                  x1: Optional[torch.Tensor] = None):
    """y = conv2d(x, w, bias, padding=1) on the HIP library; x (B, cin, H, W) channels_last, returns (B, cout, H, W) channels_last.
    ``dgrad``: ``w_oihw`` is the FORWARD layer's weight and the operator is its data gradient, conv2d(x, w.flip(2, 3).transpose(0, 1),
    padding=1) -- the pack kernels read the flipped / role-swapped weight in place (nd_pack_conv3x3_*_weight_dgrad).
    ``stats``: returns (y, st, sc) with the kernel's GroupNorm statistics epilogue -- per-(sample, slot, channel) {sum, M2} partials ``st`` and the
    pixel counts ``sc`` of the slots, what nd_groupnorm_finalize*_f32 pools."""
    lib = L.load()
    B, cin, H, W = x.shape
    c0 = cin
    if x1 is not None:                                                   # the virtual concatenation cat((x, x1), 1): the kernels read both sources (nd_src.p0 / p1)
        if x1.shape[0] != B or tuple(x1.shape[2:]) != (H, W) or x1.device != x.device or dgrad:
            raise ValueError(f"conv3x3: second source {tuple(x1.shape)} does not extend {tuple(x.shape)}")
        cin = c0 + x1.shape[1]
    cout = w_oihw.shape[1 if dgrad else 0]
    if w_oihw.shape[0 if dgrad else 1] != cin:
        raise ValueError(f"conv3x3: x {tuple(x.shape)} does not match weight {tuple(w_oihw.shape)}" + (" (data gradient)" if dgrad else ""))
    if x.device.type != "cuda":
        raise L.HipError(f"noisediff_amd.train runs on the HIP library only; tensor is on {x.device} and there is no CPU path")
    if cin % 4 or cout % 4:
        raise L.HipError(f"conv3x3 on the HIP library needs channel counts that are multiples of 4 (cin={cin}, cout={cout})")
    st = _stream(x.device)
    wbase = _PackCache._base(w_oihw) if _PackCache.cacheable(w_oihw) else None     # a parameter: the packing lives until the optimizer changes it, all stale ones repacked in one launch
    cached = wbase is not None
    w_oihw = w_oihw.detach().to(torch.float32).contiguous()
    with _on(x.device):
        out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        src_bytes = B * H * W * cin * 4
        wino = H >= 16 and W >= 16 and cin % 8 == 0
        wino4 = (wino and W >= 32 and (W % 32 == 0 or W >= 96) and W <= 2048 and cin > 16 and cout <= 2048
                 and (B * H * W + W + 2) * cin * 4 < (1 << 30) - (1 << 16) and B * H * W + W + 2 < (1 << 24))
        if x1 is not None and not (wino4 and c0 % 16 == 0 and x1.shape[1] % 16 == 0):
            raise L.HipError(f"conv3x3 over two sources needs the F(4x4) kernel and sources of whole 16-channel chunks ({c0} + {x1.shape[1]} channels, {H}x{W})")
        wino2 = wino and B * H * W < (1 << 24) and src_bytes < (1 << 31)
        if wino4:
            pack, entry = "nd_pack_conv3x3_wino4_weight", "nd_conv3x3_wino4_nhwc_f32"
        elif wino2:
            pack, entry = "nd_pack_conv3x3_wino_weight", "nd_conv3x3_wino2_nhwc_f32"
        else:
            pack, entry = "nd_pack_conv3x3_weight", "nd_conv3x3_nhwc_f32"
            if cin % 8:
                raise L.HipError(f"conv3x3 on the HIP library needs cin % 8 == 0 (cin={cin})")
        if wino4 and cached:
            wptr = _pack_cache(x.device).get(w_oihw, cin, cout, dgrad, st, kind="w4", base=wbase)
        else:
            wp = torch.empty(int(getattr(lib, pack + "_floats")(cin, cout)), dtype=torch.float32, device=x.device)
            L.call(pack + ("_dgrad" if dgrad else ""), w_oihw.data_ptr(), wp.data_ptr(), cin, cout, st)
            wptr = wp.data_ptr()
        d = L.Conv3x3()
        d.src.p0, d.src.c0, d.src.ld0, d.src.mode = x.data_ptr(), c0, c0, L.PRO_NONE
        if x1 is not None:
            d.src.p1, d.src.c1, d.src.ld1 = x1.data_ptr(), x1.shape[1], x1.shape[1]
        d.weight, d.out = wptr, out.data_ptr()
        if bias is not None:
            b = bias.detach().to(torch.float32).contiguous()
            d.bias = b.data_ptr()
        d.B, d.H, d.W, d.cin, d.cout, d.ldo = B, H, W, cin, cout, cout
        splits = int(lib.nd_conv3x3_wino4_splitk_plan(B, H, W, cin, cout)) if wino4 and _SPLIT_K else 1
        st_t = sc_t = None
        if stats:                                                 # (the split-K form leaves the same slots from its reduction kernel)
            slots = int(lib.nd_conv3x3_wino4_stat_slots(H, W) if wino4 else lib.nd_conv3x3_wino_stat_slots(H, W) if wino2
                        else lib.nd_conv3x3_stat_slots(H, W, cout, B))
            st_t = torch.empty((B, slots, cout, 2), dtype=torch.float32, device=x.device)
            sc_t = torch.empty(slots, dtype=torch.float32, device=x.device)
            d.stats, d.slot_count = st_t.data_ptr(), sc_t.data_ptr()
        if splits > 1:      # few (sample, region, cout tile) items for 256 CUs (the deep layers at training batch sizes): cut along cin
            ws = torch.empty(int(lib.nd_conv3x3_wino4_splitk_workspace_floats(B, H, W, cout, splits)), dtype=torch.float32, device=x.device)
            L.call("nd_conv3x3_wino4_splitk_nhwc_f32", C.byref(d), ws.data_ptr(), splits, st)
        else:
            L.call(entry, C.byref(d), st)
        # wp / b are dropped on return: safe, because the kernels were enqueued on torch's CURRENT stream and the caching
        # allocator reuses a block only for work that is enqueued later on that same stream
    return (out, st_t, sc_t) if stats else out


class Conv3x3Function(torch.autograd.Function):
    """nn.Conv2d(cin, cout, 3, padding=1) forward and backward on libnoisediff_hip."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_stats=False):
        xn = _nhwc(x)
        ctx.save_for_backward(xn, weight)
        ctx.has_bias = bias is not None
        ctx.n_out = 1
        if not want_stats:
            return _conv3x3_nhwc(xn, weight, bias)
        y, st, sc = _conv3x3_nhwc(xn, weight, bias, stats=True)
        ctx.mark_non_differentiable(st, sc)
        ctx.set_materialize_grads(False)                         # no zero tensors for the statistics outputs' (never used) gradients
        ctx.n_out = 3
        return y, st, sc

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, *unused):
        xn, weight = ctx.saved_tensors
        if grad_out is None:                                     # (materialize_grads is off for the statistics variant)
            return None, None, None, None
        lib = L.load()
        g = _nhwc(grad_out)
        B, cin, H, W = xn.shape
        cout = weight.shape[0]
        grad_x = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            # dL/dx = conv(dL/dy, flip(w)^T): the forward operator itself, its weight packed straight from the forward layer's
            grad_x = _conv3x3_nhwc(g, weight, None, dgrad=True)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            with _on(xn.device):
                grad_w = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=xn.device)
                grad_b = torch.empty(cout, dtype=torch.float32, device=xn.device) if want_b else None     # falls out of the staged dY tiles
                ws = torch.empty(int(lib.nd_conv3x3_wgrad_workspace_floats(B, H, W, cin, cout)), dtype=torch.float32, device=xn.device)
                L.call("nd_conv3x3_wgrad_nhwc_f32", xn.data_ptr(), cin, g.data_ptr(), cout, grad_w.data_ptr(),
                       grad_b.data_ptr() if want_b else None, ws.data_ptr(), B, H, W, cin, cout, _stream(xn.device))
        elif want_b:
            grad_b = g.sum(dim=(0, 2, 3))
        return grad_x, grad_w, grad_b, None


def conv3x3(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable F.conv2d(x, weight, bias, padding=1) for 3x3 kernels on the HIP library."""
    if tuple(weight.shape[2:]) != (3, 3) or x.dim() != 4 or x.shape[1] != weight.shape[1]:
        raise ValueError(f"conv3x3: x {tuple(x.shape)} / weight {tuple(weight.shape)} is not a 3x3 convolution")
    return Conv3x3Function.apply(x, weight, bias)


def conv3x3_with_stats(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """conv3x3 plus the kernel's GroupNorm statistics epilogue: (y, (st, sc)) for ``group_norm_silu(y, ..., conv_stats=)`` -- the norm then needs no
    pass of its own over y for its moments, as in the sampling engine (Block: conv -> statistics -> finalize -> apply); the split-K form of the deep
    layers leaves the same slots from its reduction kernel."""
    if tuple(weight.shape[2:]) != (3, 3) or x.dim() != 4 or x.shape[1] != weight.shape[1]:
        raise ValueError(f"conv3x3: x {tuple(x.shape)} / weight {tuple(weight.shape)} is not a 3x3 convolution")
    y, st, sc = Conv3x3Function.apply(x, weight, bias, True)
    return y, (st, sc)


def cat_sources_ok(x0: torch.Tensor, x1: torch.Tensor, cout: Optional[int] = None) -> bool:
    """Can conv3x3_cat / conv1x1_cat take this pair (else the caller concatenates)?  The F(4x4) kernel's geometry and whole 16-channel chunks per source;
    ``cout`` (when given): a multiple of 8 within the kernel's bias table (2048)."""
    B, c0, H, W = x0.shape
    c1 = x1.shape[1]
    if cout is not None and (cout % 8 or cout > 2048):
        return False
    return (x0.is_cuda and x1.is_cuda and x0.dim() == 4 and tuple(x1.shape[2:]) == (H, W) and x1.shape[0] == B and c0 % 16 == 0 and c1 % 16 == 0
            and H >= 16 and W >= 32 and (W % 32 == 0 or W >= 96) and W <= 2048
            and (B * H * W + W + 2) * (c0 + c1) * 4 < (1 << 30) - (1 << 16) and B * H * W + W + 2 < (1 << 24))


class Conv3x3CatFunction(torch.autograd.Function):
    """conv3x3(torch.cat((x0, x1), 1), weight, bias) without the concatenated tensor: the up path's ``cat((x, skip))`` in front of a ResnetBlock
    (Diffusion_arch.py:620-630).  Forward: the kernel reads both sources (nd_src.p0 / p1).  Backward: one data gradient of c0 + c1 channels whose channel
    slices are the two inputs' gradients; the weight gradient per source, joined along cin."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, want_stats=False):
        a, b = _nhwc(x0), _nhwc(x1)
        ctx.save_for_backward(a, b, weight)
        ctx.has_bias = bias is not None
        if not want_stats:
            return _conv3x3_nhwc(a, weight, bias, x1=b)
        y, st, sc = _conv3x3_nhwc(a, weight, bias, stats=True, x1=b)
        ctx.mark_non_differentiable(st, sc)
        ctx.set_materialize_grads(False)
        return y, st, sc

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, *unused):
        a, b, weight = ctx.saved_tensors
        if grad_out is None:
            return None, None, None, None, None
        lib = L.load()
        g = _nhwc(grad_out)
        B, c0, H, W = a.shape
        c1, cout = b.shape[1], weight.shape[0]
        gx0 = gx1 = grad_w = grad_b = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            full = _conv3x3_nhwc(g, weight, None, dgrad=True)            # (B, c0 + c1, H, W) channels_last: the slices are views
            gx0, gx1 = (full[:, :c0] if ctx.needs_input_grad[0] else None), (full[:, c0:] if ctx.needs_input_grad[1] else None)
        want_b = ctx.has_bias and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[2] and H % 4 == 0 and W % 16 == 0 and c0 % 32 == 0 and c1 % 32 == 0 and cout % 16 == 0 and lib.nd_conv3x3_wgrad_form(-1) != 1:
            with _on(a.device):                                          # the Winograd-domain forms read a cin block from either source: ONE weight gradient
                grad_w = torch.empty((cout, c0 + c1, 3, 3), dtype=torch.float32, device=a.device)
                grad_b = torch.empty(cout, dtype=torch.float32, device=a.device) if want_b else None
                ws = torch.empty(int(lib.nd_conv3x3_wgrad_workspace_floats(B, H, W, c0 + c1, cout)), dtype=torch.float32, device=a.device)
                L.call("nd_conv3x3_wgrad_cat_nhwc_f32", a.data_ptr(), c0, c0, b.data_ptr(), c1, c1, g.data_ptr(), cout, grad_w.data_ptr(),
                       grad_b.data_ptr() if want_b else None, ws.data_ptr(), B, H, W, cout, _stream(a.device))
        elif ctx.needs_input_grad[2]:
            with _on(a.device):
                parts = []
                for i, (src, c) in enumerate(((a, c0), (b, c1))):
                    dw = torch.empty((cout, c, 3, 3), dtype=torch.float32, device=a.device)
                    if i == 0 and want_b:
                        grad_b = torch.empty(cout, dtype=torch.float32, device=a.device)
                    ws = torch.empty(int(lib.nd_conv3x3_wgrad_workspace_floats(B, H, W, c, cout)), dtype=torch.float32, device=a.device)
                    L.call("nd_conv3x3_wgrad_nhwc_f32", src.data_ptr(), c, g.data_ptr(), cout, dw.data_ptr(),
                           grad_b.data_ptr() if (i == 0 and want_b) else None, ws.data_ptr(), B, H, W, c, cout, _stream(a.device))
                    parts.append(dw)
                grad_w = torch.cat(parts, dim=1)
        elif want_b:
            grad_b = g.sum(dim=(0, 2, 3))
        return gx0, gx1, grad_w, grad_b, None


def conv3x3_cat(x0: torch.Tensor, x1: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, with_stats: bool = False):
    """Differentiable F.conv2d(torch.cat((x0, x1), 1), weight, bias, padding=1) that never builds the concatenation (``cat_sources_ok`` says when);
    ``with_stats``: (y, (st, sc)) as conv3x3_with_stats."""
    if tuple(weight.shape[2:]) != (3, 3) or x0.shape[1] + x1.shape[1] != weight.shape[1] or not cat_sources_ok(x0, x1, weight.shape[0]):
        raise ValueError(f"conv3x3_cat: x0 {tuple(x0.shape)} + x1 {tuple(x1.shape)} / weight {tuple(weight.shape)}")
    if not with_stats:
        return Conv3x3CatFunction.apply(x0, x1, weight, bias)
    y, st, sc = Conv3x3CatFunction.apply(x0, x1, weight, bias, True)
    return y, (st, sc)


class GroupNormFunction(torch.autograd.Function):
    """nn.GroupNorm(groups, C) forward and backward on libnoisediff_hip (norm_train.hip), channels_last in and out."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        lib = L.load()
        xn = _nhwc(x)
        if xn.device.type != "cuda":
            raise L.HipError(f"noisediff_amd.train runs on the HIP library only; tensor is on {xn.device} and there is no CPU path")
        B, C_, H, W = xn.shape
        st = _stream(xn.device)
        with _on(xn.device):
            y = torch.empty_like(xn, memory_format=torch.channels_last)
            mean_rstd = torch.empty((B, groups, 2), dtype=torch.float32, device=xn.device)
            ws = torch.empty(int(lib.nd_groupnorm_train_workspace_floats(B, H * W, C_)), dtype=torch.float32, device=xn.device)
            w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
            L.call("nd_groupnorm_train_forward_f32", xn.data_ptr(), C_, w32.data_ptr(), b32.data_ptr(), y.data_ptr(), C_, mean_rstd.data_ptr(),
                   ws.data_ptr(), B, H * W, C_, groups, float(eps), st)
        ctx.save_for_backward(xn, weight, mean_rstd)
        ctx.groups = groups
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        xn, weight, mean_rstd = ctx.saved_tensors
        lib = L.load()
        g = _nhwc(grad_out)
        B, C_, H, W = xn.shape
        with _on(xn.device):
            dx = torch.empty_like(xn, memory_format=torch.channels_last)
            dgamma = torch.empty(C_, dtype=torch.float32, device=xn.device)
            dbeta = torch.empty(C_, dtype=torch.float32, device=xn.device)
            ws = torch.empty(int(lib.nd_groupnorm_train_workspace_floats(B, H * W, C_)), dtype=torch.float32, device=xn.device)
            w32 = weight.detach().float().contiguous()
            L.call("nd_groupnorm_train_backward_f32", g.data_ptr(), C_, xn.data_ptr(), C_, w32.data_ptr(), mean_rstd.data_ptr(), dx.data_ptr(), C_,
                   dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), B, H * W, C_, ctx.groups, _stream(xn.device))
        return dx, dgamma, dbeta, None, None


def group_norm(x: torch.Tensor, groups: int, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """Differentiable F.group_norm(x, groups, weight, bias, eps) for 4-D inputs on the HIP library."""
    if x.dim() != 4 or x.shape[1] % groups or x.shape[1] % 4 or x.shape[1] > 1024 or weight is None or bias is None:
        raise ValueError(f"group_norm: x {tuple(x.shape)}, groups {groups}: needs a 4-D input, C a multiple of 4 and of groups (<= 1024), affine parameters")
    return GroupNormFunction.apply(x, weight, bias, groups, eps)


class GroupNormSiLUFunction(torch.autograd.Function):
    """y = silu(GroupNorm(x) * (scale + 1) + shift) -- Block's tail (Diffusion_arch.py:137-143) -- as one operator on libnoisediff_hip
    (norm_train.hip); ``scale_shift`` is the (B, 2C) output of ResnetBlock.mlp (scale | shift) or None."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale_shift, res, groups, eps, st=None, sc=None):
        lib = L.load()
        xn = _nhwc(x)
        rn = None if res is None else _nhwc(res)
        if xn.device.type != "cuda":
            raise L.HipError(f"noisediff_amd.train runs on the HIP library only; tensor is on {xn.device} and there is no CPU path")
        B, C_, H, W = xn.shape
        ss = None if scale_shift is None else scale_shift.detach().reshape(B, 2 * C_).float().contiguous()
        with _on(xn.device):
            y = torch.empty_like(xn, memory_format=torch.channels_last)
            mean_rstd = torch.empty((B, groups, 2), dtype=torch.float32, device=xn.device)
            mad = torch.empty((B, 3, C_), dtype=torch.float32, device=xn.device)
            ws = torch.empty(int(lib.nd_groupnorm_silu_train_workspace_floats(B, H * W, C_)), dtype=torch.float32, device=xn.device)
            w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
            if st is not None:       # the convolution's statistics epilogue already holds the partial moments: finalize (saving mean / rstd) + one apply pass
                stp = _stream(xn.device)
                L.call("nd_groupnorm_finalize_train_f32", st.data_ptr(), sc.data_ptr(), st.shape[1], w32.data_ptr(), b32.data_ptr(),
                       None if ss is None else ss.data_ptr(), 2 * C_, mad.data_ptr(), mean_rstd.data_ptr(), B, C_, groups, float(eps), stp)
                L.call("nd_affine_silu_add_f32", xn.data_ptr(), C_, mad.data_ptr(), None if rn is None else rn.data_ptr(), C_, None, 0, y.data_ptr(), C_,
                       B, H * W, C_, stp)
            else:
                L.call("nd_groupnorm_silu_train_forward_f32", xn.data_ptr(), C_, w32.data_ptr(), b32.data_ptr(), None if ss is None else ss.data_ptr(),
                       None if rn is None else rn.data_ptr(), C_, y.data_ptr(), C_, mean_rstd.data_ptr(), mad.data_ptr(), ws.data_ptr(), B, H * W, C_, groups, float(eps), _stream(xn.device))
        ctx.save_for_backward(xn, weight, bias, mean_rstd, mad, ss if ss is not None else mean_rstd.new_empty(0))
        ctx.groups, ctx.has_ss, ctx.ss_shape = groups, ss is not None, None if scale_shift is None else scale_shift.shape
        ctx.has_res = res is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        xn, weight, bias, mean_rstd, mad, ss = ctx.saved_tensors
        lib = L.load()
        g = _nhwc(grad_out)
        B, C_, H, W = xn.shape
        with _on(xn.device):
            dx = torch.empty_like(xn, memory_format=torch.channels_last)
            dgamma = torch.empty(C_, dtype=torch.float32, device=xn.device)
            dbeta = torch.empty(C_, dtype=torch.float32, device=xn.device)
            dss = torch.empty((B, 2 * C_), dtype=torch.float32, device=xn.device) if ctx.has_ss else None
            ws = torch.empty(int(lib.nd_groupnorm_silu_train_workspace_floats(B, H * W, C_)), dtype=torch.float32, device=xn.device)
            w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
            L.call("nd_groupnorm_silu_train_backward_f32", g.data_ptr(), C_, xn.data_ptr(), C_, w32.data_ptr(), b32.data_ptr(),
                   ss.data_ptr() if ctx.has_ss else None, mean_rstd.data_ptr(), mad.data_ptr(), dx.data_ptr(), C_, dgamma.data_ptr(), dbeta.data_ptr(),
                   dss.data_ptr() if ctx.has_ss else None, ws.data_ptr(), B, H * W, C_, ctx.groups, _stream(xn.device))
        return dx, dgamma, dbeta, (dss.view(ctx.ss_shape) if ctx.has_ss else None), (grad_out if ctx.has_res else None), None, None, None, None


def group_norm_silu(x: torch.Tensor, groups: int, weight: torch.Tensor, bias: torch.Tensor, scale_shift: Optional[torch.Tensor] = None,
                    eps: float = 1e-5, res: Optional[torch.Tensor] = None, conv_stats=None) -> torch.Tensor:
    """Differentiable silu(F.group_norm(x, groups, weight, bias, eps) * (scale + 1) + shift) (+ res), scale | shift = the halves of ``scale_shift``
    ((B, 2C) or (B, 2C, 1, 1)), on the HIP library; ``res`` (x's shape): the ResnetBlock's shortcut added in the same pass (Diffusion_arch.py:170)."""
    if x.dim() != 4 or x.shape[1] % groups or x.shape[1] % 4 or x.shape[1] > 1024 or weight is None or bias is None:
        raise ValueError(f"group_norm_silu: x {tuple(x.shape)}, groups {groups}: needs a 4-D input, C a multiple of 4 and of groups (<= 1024), affine parameters")
    if scale_shift is not None and scale_shift.numel() != x.shape[0] * 2 * x.shape[1]:
        raise ValueError(f"group_norm_silu: scale_shift {tuple(scale_shift.shape)} is not (B, 2C) for x {tuple(x.shape)}")
    if res is not None and res.shape != x.shape:
        raise ValueError(f"group_norm_silu: res {tuple(res.shape)} does not match x {tuple(x.shape)}")
    if conv_stats is not None:       # (st, sc) of conv3x3_with_stats for THIS x: the partial moments of the producing convolution's epilogue
        return GroupNormSiLUFunction.apply(x, weight, bias, scale_shift, res, groups, eps, conv_stats[0], conv_stats[1])
    return GroupNormSiLUFunction.apply(x, weight, bias, scale_shift, res, groups, eps)


class BroadcastAddFunction(torch.autograd.Function):
    """tokens (B, N, C) + vec (B, 1, C): AttnBlock's one-token cross attention adds the same vector to every token (Diffusion_arch.py:435-437).  The
    gradient of ``vec`` is a sum over the tokens: nd_token_sum_f32 (fixed order) instead of ATen's reduction over a middle dimension."""

    @staticmethod
    def forward(ctx, tokens, vec):
        ctx.vec_shape = vec.shape
        return tokens + vec

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        lib = L.load()
        B, N, C_ = grad_out.shape
        g = grad_out if grad_out.is_contiguous() and grad_out.dtype == torch.float32 else grad_out.float().contiguous()
        grad_vec = None
        if ctx.needs_input_grad[1]:
            with _on(g.device):
                grad_vec = torch.empty((B, C_), dtype=torch.float32, device=g.device)
                ws = torch.empty(int(lib.nd_token_sum_workspace_floats(B, N, C_)), dtype=torch.float32, device=g.device)
                L.call("nd_token_sum_f32", g.data_ptr(), C_, grad_vec.data_ptr(), ws.data_ptr(), B, N, C_, _stream(g.device))
            grad_vec = grad_vec.view(ctx.vec_shape)
        return grad_out, grad_vec


def broadcast_add(tokens: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    """tokens (B, N, C) + vec (B, 1, C) with the token sum of the backward on the HIP library (C a multiple of 4, <= 1024; CUDA tensors)."""
    if tokens.dim() != 3 or vec.shape != (tokens.shape[0], 1, tokens.shape[2]) or tokens.shape[2] % 4 or tokens.shape[2] > 1024 or not tokens.is_cuda:
        return tokens + vec
    return BroadcastAddFunction.apply(tokens, vec)


class ModulateSiLUFunction(torch.autograd.Function):
    """y = silu(n * (scale + 1) + shift) with per-pixel maps (ResnetBlock2, Diffusion_arch.py:188-192): ``n`` (B, C, H, W) the GroupNorm's output,
    ``ss`` (B, 2C, H, W) = scale | shift as ResnetBlock2.mlp emits it; one pass forward, one backward (nd_modulate_silu_*_f32)."""

    @staticmethod
    def forward(ctx, n, ss):
        nn_, sn = _nhwc(n), _nhwc(ss)
        if nn_.device.type != "cuda":
            raise L.HipError(f"noisediff_amd.train runs on the HIP library only; tensor is on {nn_.device} and there is no CPU path")
        B, C_, H, W = nn_.shape
        with _on(nn_.device):
            y = torch.empty_like(nn_, memory_format=torch.channels_last)
            L.call("nd_modulate_silu_forward_f32", nn_.data_ptr(), C_, sn.data_ptr(), 2 * C_, y.data_ptr(), C_, B * H * W, C_, _stream(nn_.device))
        ctx.save_for_backward(nn_, sn)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        nn_, sn = ctx.saved_tensors
        g = _nhwc(grad_out)
        B, C_, H, W = nn_.shape
        with _on(nn_.device):
            dn = torch.empty_like(nn_, memory_format=torch.channels_last)
            dss = torch.empty_like(sn, memory_format=torch.channels_last)
            L.call("nd_modulate_silu_backward_f32", g.data_ptr(), C_, nn_.data_ptr(), C_, sn.data_ptr(), 2 * C_, dn.data_ptr(), C_, dss.data_ptr(), 2 * C_,
                   B * H * W, C_, _stream(nn_.device))
        return dn, dss


def modulate_silu(n: torch.Tensor, scale_shift: torch.Tensor) -> torch.Tensor:
    """Differentiable silu(n * (scale + 1) + shift), scale | shift = the channel halves of the per-pixel map ``scale_shift`` (B, 2C, H, W)."""
    if n.dim() != 4 or scale_shift.dim() != 4 or scale_shift.shape != (n.shape[0], 2 * n.shape[1], n.shape[2], n.shape[3]) or n.shape[1] % 4:
        raise ValueError(f"modulate_silu: n {tuple(n.shape)} / scale_shift {tuple(scale_shift.shape)}: needs (B, C, H, W) and (B, 2C, H, W), C a multiple of 4")
    return ModulateSiLUFunction.apply(n, scale_shift)


def _group_norm_ok(channels: int, groups: int) -> bool:
    """Channel counts the norm_train.hip kernels take (the C side rejects the rest): C % 4 == 0, C <= 1024, C / groups <= 512."""
    return channels % 4 == 0 and channels % groups == 0 and channels <= 1024 and channels // groups <= 512


def _eligible_norm(m: nn.Module) -> bool:
    return isinstance(m, nn.GroupNorm) and m.affine and _group_norm_ok(m.num_channels, m.num_groups)


def _hip_norm_forward(self: nn.GroupNorm, x: torch.Tensor) -> torch.Tensor:
    if x.dim() != 4:                                                     # (the reference applies GroupNorm to images only)
        return torch.nn.functional.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
    return group_norm(x, self.num_groups, self.weight, self.bias, self.eps)


# Output and data gradient of Linears / 1x1 convolutions: the sampling path's pointwise kernels (nd_pointwise_gemm_nhwc_f32), since r5 by default --
# no library GEMM (rocBLAS / hipBLASLt) is left in a training step.  The packed operands are cached per weight and refreshed when the parameter's
# version changes: ONE launch per optimizer step packs every stale weight (nd_pack_pointwise_weights_batch), where r3 packed per call (178 GEMMs,
# 0.77 ms of 4-microsecond launches per step -- that, not the GEMMs, was what lost to the library then).  ND_TRAIN_PW=0 (or train._PW_GEMM = False):
# the library GEMM, as an A/B knob.
_PW_GEMM = __import__("os").environ.get("ND_TRAIN_PW", "1") != "0"


class _PackCache:
    """Packed operands of one device's weights: the Linear / 1x1 weights (kind "pw": nd_pack_pointwise_weight, ``flag`` = the transposed data-gradient
    packing) and the 3x3 weights on the F(4x4) kernels (kind "w4": nd_pack_conv3x3_wino4_weight, ``flag`` = the data-gradient form).  An entry belongs
    to an ``nn.Parameter`` (or a view of one: ``weight.flatten(1)`` of a 1x1 convolution): key (kind, data_ptr, shape, flag), valid while the parameter's
    version counter stands.  The first stale entry of a kind that a step meets repacks EVERY stale entry of that kind in one launch over a descriptor
    table kept in device memory.  Anything that is not a parameter (the per-step torch.cat of the stacked time projections, nn.DataParallel's
    broadcast copies, a padded temporary under no_grad) is packed per call and never cached.

    What the version counter does NOT see -- call ``train.invalidate_packs()`` after any of these, before the next eager forward:
      * writes through ``.data`` (``p.data.copy_ / lerp_ / normal_``: ema_pytorch's update of its shadow model, SID_arch's re-initialisation);
      * replays of a captured whole-step graph: the optimizer inside it moves the parameters without bumping a version.  Entries that were packed
        while a stream was capturing are flagged ``volatile`` and repacked on EVERY eager use from then on, so the sequence replay / eager forward /
        replay never computes with a packing from before the last update even without the call.
    An entry holds its parameter (as in r5): a deleted model's weights and packings stay in GPU memory until ``train.release_packs()`` is called (or 4096
    entries accumulate).  r6 tried weak references here: the full GPU suite then ended in a GPU memory access fault in a later test's backward
    pass on two of three runs -- freeing those models changes which pages of the caching allocator stay mapped, and some access of the training path (not
    found in the time of the round) reaches into them; with the r5 lifetime the suite is green as it was.  Entries whose parameter has MOVED (module.to(),
    p.data = ...) are dropped: their address is no longer the parameter's."""
    _FLOATS = {"pw": "nd_pack_pointwise_weight_floats", "w4": "nd_pack_conv3x3_wino4_weight_floats"}
    _BATCH = {"pw": "nd_pack_pointwise_weights_batch", "w4": "nd_pack_conv3x3_wino4_weights_batch"}
    # entry fields
    _REF, _BUF, _VER, _CIN, _COUT, _FLAG, _VOLATILE = range(7)

    def __init__(self, device: torch.device):
        self.device = device
        self.entries: Dict[tuple, list] = {}        # key -> [ref() -> the base parameter (held), packed buffer, version packed, cin, cout, flag, volatile]
        self.tables: Dict[str, Optional[torch.Tensor]] = {"pw": None, "w4": None}     # device copies of the nd_pack_item records, in the order of `order`
        self.order: Dict[str, list] = {"pw": [], "w4": []}

    @staticmethod
    def _base(w: torch.Tensor):
        """The nn.Parameter behind ``w``: itself, or the root of a view of it (``p.detach()``, ``p.flatten(1)``, ``p.detach().flatten(1)``); else None."""
        if isinstance(w, torch.nn.Parameter):
            return w
        b = w._base if w._is_view() else None
        return b if isinstance(b, torch.nn.Parameter) else None

    @staticmethod
    def cacheable(w: torch.Tensor) -> bool:
        return w.dtype == torch.float32 and w.is_contiguous() and w.is_cuda and _PackCache._base(w) is not None

    def _drop(self, key: tuple) -> None:
        self.entries.pop(key, None)
        kind = key[0]
        if key in self.order[kind]:
            self.order[kind].remove(key)
        self.tables[kind] = None

    def invalidate(self) -> None:
        for e in self.entries.values():
            e[self._VER] = -1

    def get(self, w: torch.Tensor, cin: int, cout: int, flag: bool, st, kind: str = "pw", base=None) -> int:
        """Device pointer of the packing of ``w`` (``base``: the parameter behind it where the caller has already detached ``w``)."""
        lib = L.load()
        key = (kind, w.data_ptr(), tuple(w.shape), flag)
        base = base if base is not None else self._base(w)
        if base is None:
            raise L.HipError("_PackCache.get: not a parameter (check cacheable() first)")
        e = self.entries.get(key)
        if e is not None and e[self._REF]() is not base:     # the parameter that owned this address is gone (or replaced): the entry goes with it
            self._drop(key)
            e = None
        if e is None:
            if len(self.entries) >= 4096:            # (a model is a few hundred weights: anything beyond is a leak -- start over)
                self.entries.clear(); self.order = {"pw": [], "w4": []}; self.tables = {"pw": None, "w4": None}
            buf = torch.empty(int(getattr(lib, self._FLOATS[kind])(cin, cout)), dtype=torch.float32, device=self.device)
            e = self.entries[key] = [(lambda b=base: b), buf, -1, cin, cout, flag, False]
            self.order[kind].append(key)
            self.tables[kind] = None
        capturing = torch.cuda.is_current_stream_capturing()
        if e[self._VER] != base._version or (e[self._VOLATILE] and not capturing):
            if e[self._VOLATILE] and not capturing:
                e[self._VER] = -1                    # packed inside a captured graph: replays may have moved the weight since, unseen by the version counter
            self._repack_stale(kind, st)
        return e[self._BUF].data_ptr()

    def _pack_one(self, kind: str, key, e, st) -> None:
        ptr, buf, cin, cout, flag = key[1], e[self._BUF], e[self._CIN], e[self._COUT], e[self._FLAG]
        if kind == "pw":
            L.call("nd_pack_pointwise_weight_t" if flag else "nd_pack_pointwise_weight", ptr, buf.data_ptr(), cin, cout, *(() if flag else (0,)), st)
        else:
            L.call("nd_pack_conv3x3_wino4_weight" + ("_dgrad" if flag else ""), ptr, buf.data_ptr(), cin, cout, st)

    def _repack_stale(self, kind: str, st) -> None:
        def gone(k):
            # the parameter has been freed, or its storage has moved (module.to(), p.data = ...) and the address this entry was keyed on is no longer inside it:
            # the entry must not be repacked from that address (ADVICE r5 asked for weak references; a strong one used to keep the old storage alive)
            base = self.entries[k][self._REF]()
            if base is None:
                return True
            lo = base.data_ptr()
            return not (lo <= k[1] < lo + max(base.numel(), 1) * base.element_size())
        for k in [k for k in self.order[kind] if gone(k)]:
            self._drop(k)
        order = self.order[kind]
        live = {k: self.entries[k][self._REF]() for k in order}
        stale = [k for k in order if self.entries[k][self._VER] != live[k]._version]
        capturing = torch.cuda.is_current_stream_capturing()
        if (self.tables[kind] is None or len(stale) != len(order)) and capturing:
            # a new descriptor table would need a host-to-device copy, which a capturing stream refuses: one capturable launch per stale weight instead
            for k in stale:
                e = self.entries[k]
                self._pack_one(kind, k, e, st)
                e[self._VER], e[self._VOLATILE] = live[k]._version, True
            return
        if self.tables[kind] is None or len(stale) != len(order):
            # first use, or only part of the table is stale (a partially frozen model): a table of the stale entries only
            items = (L.PackItem * len(stale))()
            for i, k in enumerate(stale):
                e = self.entries[k]
                items[i].w, items[i].packed, items[i].cin, items[i].cout, items[i].transposed = k[1], e[self._BUF].data_ptr(), e[self._CIN], e[self._COUT], int(e[self._FLAG])
            host = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
            table = host.to(self.device)                     # (synchronous copy of a few KB)
            if len(stale) == len(order):
                self.tables[kind] = table
        else:
            table = self.tables[kind]
        L.call(self._BATCH[kind], table.data_ptr(), len(stale), st)
        self._last_table = getattr(self, "_last_table", {})
        self._last_table[kind] = table                       # alive until the next repack (the launch is asynchronous)
        for k in stale:
            self.entries[k][self._VER] = live[k]._version
            if capturing:
                self.entries[k][self._VOLATILE] = True


_PACK_CACHES: Dict[int, _PackCache] = {}


def _pack_cache(device: torch.device) -> _PackCache:
    cache = _PACK_CACHES.get(device.index)
    if cache is None:
        cache = _PACK_CACHES[device.index] = _PackCache(device)
    return cache


def release_packs(device: Optional[torch.device] = None) -> None:
    """Drop every cached packing (and the references to their parameters) of one device, or of all: call it when models are deleted, so that their weights
    and packings leave GPU memory; live models repack on their next forward."""
    for idx in list(_PACK_CACHES):
        if device is None or device.index == idx:
            del _PACK_CACHES[idx]


def invalidate_packs(device: Optional[torch.device] = None) -> None:
    """Mark every cached weight packing (of one device, or of all) stale: the next forward repacks them in one launch per kind.  Needed after writes the
    parameters' version counters do not see -- ``p.data.copy_ / lerp_ / normal_`` (ema_pytorch's shadow-model update, SID_arch's initialisation) -- and after
    replays of a captured whole-step graph (see _PackCache)."""
    for idx, cache in _PACK_CACHES.items():
        if device is None or device.index == idx:
            cache.invalidate()


def _tokens(t: torch.Tensor, c: int) -> torch.Tensor:
    t = t.reshape(-1, c)
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


def _pointwise_gemm(x2: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], transposed: bool, x2b: Optional[torch.Tensor] = None,
                    res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[N, cout] = x2[N, cin] @ W^T (+ bias) (+ res[N, cout]) on nd_pointwise_gemm_nhwc_f32 (pointwise.hip: the sampling path's 1x1 / Linear kernels, exact-fp32
    MFMA), W = ``w`` (cout, cin); ``transposed``: W^T = ``w`` -- the forward weight of the layer whose data gradient dx = dy @ w this is (packed
    in place by nd_pack_pointwise_weight_t).  ``res``: a residual added in the kernel's epilogue (the sampling engine's res0)."""
    lib = L.load()
    N, cin = x2.shape
    c0 = cin
    if x2b is not None:                                                  # tokens of a second source: the virtual concatenation cat((x2, x2b), -1)
        cin = c0 + x2b.shape[1]
    cout = w.shape[1] if transposed else w.shape[0]
    st = _stream(x2.device)
    with _on(x2.device):
        y = torch.empty((N, cout), dtype=torch.float32, device=x2.device)
        if _PackCache.cacheable(w):                   # a parameter (or a view of one): its packing is cached until the optimizer changes it
            wptr = _pack_cache(x2.device).get(w, cin, cout, transposed, st)
        else:
            wp = torch.empty(int(lib.nd_pack_pointwise_weight_floats(cin, cout)), dtype=torch.float32, device=x2.device)
            w32 = w.detach().float().contiguous()
            if transposed:
                L.call("nd_pack_pointwise_weight_t", w32.data_ptr(), wp.data_ptr(), cin, cout, st)
            else:
                L.call("nd_pack_pointwise_weight", w32.data_ptr(), wp.data_ptr(), cin, cout, 0, st)
            wptr = wp.data_ptr()
        d = L.Pointwise()
        d.src.p0, d.src.c0, d.src.ld0, d.src.mode = x2.data_ptr(), c0, c0, L.PRO_NONE
        if x2b is not None:
            d.src.p1, d.src.c1, d.src.ld1 = x2b.data_ptr(), x2b.shape[1], x2b.shape[1]
        d.weight, d.out = wptr, y.data_ptr()
        if bias is not None:
            b32 = bias.detach().float().contiguous()
            d.bias = b32.data_ptr()
        if res is not None:
            d.res0, d.ldr0 = res.data_ptr(), cout
        d.B, d.HW, d.W, d.cin, d.cout, d.ldo, d.act = 1, N, 1, cin, cout, cout, L.ACT_NONE
        L.call("nd_pointwise_gemm_nhwc_f32", C.byref(d), st)
    return y


def _pw_takes(N: int, cin: int, cout: int) -> bool:
    return _PW_GEMM and cin % 4 == 0 and cout % 4 == 0 and 0 < N < (1 << 31) and N * max(cin, cout) * 4 < (1 << 40)


class LinearFunction(torch.autograd.Function):
    """y = x @ W^T + b over tokens (x: (..., cin) with the tokens contiguous).  Output and data gradient are plain GEMMs (the library's; with
    ND_TRAIN_PW=1 the sampling path's pointwise kernels, nd_pointwise_gemm_nhwc_f32, the data gradient's weight packed straight from the forward
    weight); the weight and bias gradients -- a reduction over 10^5..10^6 tokens into a 64..1024-wide matrix, where library GEMMs run at a tenth of HBM speed --
    come from nd_linear_wgrad_f32 (linear_wgrad.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res=None):
        """``res`` (optional, the shape of the output): y = x @ W^T + b + res with the addition in the GEMM's epilogue (the residual connections around the
        AttnBlock's feed-forward and ``proj_out``, Diffusion_arch.py:440-443); its gradient is the output's."""
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        cout, cin = weight.shape
        if x.is_cuda and _pw_takes(x.numel() // cin, cin, cout):
            r2 = _tokens(res, cout) if res is not None else None
            return _pointwise_gemm(_tokens(x, cin), weight, bias, False, res=r2).view(*x.shape[:-1], cout)
        y = torch.nn.functional.linear(x, weight, bias)
        return y if res is None else y + res

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        lib = L.load()
        cout, cin = weight.shape
        grad_x = grad_w = grad_b = None
        g2 = _tokens(grad_out, cout)
        if ctx.needs_input_grad[0]:
            if g2.is_cuda and _pw_takes(g2.shape[0], cout, cin):
                grad_x = _pointwise_gemm(g2, weight, None, True).view(x.shape)
            else:
                grad_x = grad_out @ weight
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            x2 = _tokens(x, cin)
            if x2.device.type != "cuda":
                raise L.HipError(f"noisediff_amd.train runs on the HIP library only; tensor is on {x2.device} and there is no CPU path")
            N = x2.shape[0]
            with _on(x2.device):
                grad_w = torch.empty((cout, cin), dtype=torch.float32, device=x2.device)
                grad_b = torch.empty(cout, dtype=torch.float32, device=x2.device) if ctx.has_bias else None
                ws = torch.empty(int(lib.nd_linear_wgrad_workspace_floats(N, cin, cout)), dtype=torch.float32, device=x2.device)
                L.call("nd_linear_wgrad_f32", x2.data_ptr(), cin, g2.data_ptr(), cout, grad_w.data_ptr(),
                       grad_b.data_ptr() if grad_b is not None else None, ws.data_ptr(), N, cin, cout, _stream(x2.device))
        return grad_x, grad_w, grad_b, (grad_out if len(ctx.needs_input_grad) > 3 and ctx.needs_input_grad[3] else None)


class LinearCatFunction(torch.autograd.Function):
    """linear(torch.cat((t0, t1), -1), weight, bias) over tokens without the concatenated tensor (the 1x1 ``res_conv`` of the up path's ResnetBlocks on
    ``cat((x, skip))``): forward with two sources, one data gradient whose channel slices are the inputs' gradients, the weight gradient per source."""

    @staticmethod
    def forward(ctx, t0, t1, weight, bias):
        ctx.save_for_backward(t0, t1, weight)
        ctx.has_bias = bias is not None
        cout = weight.shape[0]
        c0, c1 = t0.shape[-1], t1.shape[-1]
        return _pointwise_gemm(_tokens(t0, c0), weight, bias, False, _tokens(t1, c1)).view(*t0.shape[:-1], cout)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        t0, t1, weight = ctx.saved_tensors
        lib = L.load()
        cout, cin = weight.shape
        c0, c1 = t0.shape[-1], t1.shape[-1]
        g0 = g1 = grad_w = grad_b = None
        g2 = _tokens(grad_out, cout)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            full = _pointwise_gemm(g2, weight, None, True).view(*t0.shape[:-1], cin)
            g0, g1 = (full[..., :c0] if ctx.needs_input_grad[0] else None), (full[..., c0:] if ctx.needs_input_grad[1] else None)
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            with _on(g2.device):
                parts = []
                for i, (t, c) in enumerate(((t0, c0), (t1, c1))):
                    x2 = _tokens(t, c)
                    N = x2.shape[0]
                    dw = torch.empty((cout, c), dtype=torch.float32, device=x2.device)
                    if i == 0 and ctx.has_bias:
                        grad_b = torch.empty(cout, dtype=torch.float32, device=x2.device)
                    ws = torch.empty(int(lib.nd_linear_wgrad_workspace_floats(N, c, cout)), dtype=torch.float32, device=x2.device)
                    L.call("nd_linear_wgrad_f32", x2.data_ptr(), c, g2.data_ptr(), cout, dw.data_ptr(),
                           grad_b.data_ptr() if (i == 0 and grad_b is not None) else None, ws.data_ptr(), N, c, cout, _stream(x2.device))
                    parts.append(dw)
                grad_w = torch.cat(parts, dim=1)
        return g0, g1, grad_w, grad_b


def conv1x1_cat(x0: torch.Tensor, x1: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable F.conv2d(torch.cat((x0, x1), 1), weight, bias) for 1x1 kernels that never builds the concatenation (channels_last in and out; both
    channel counts multiples of 4)."""
    if x0.dim() != 4 or tuple(weight.shape[2:]) != (1, 1) or x0.shape[1] + x1.shape[1] != weight.shape[1] or x0.shape[1] % 4 or x1.shape[1] % 4 \
            or weight.shape[0] % 4 or not (x0.is_cuda and x1.is_cuda):
        raise ValueError(f"conv1x1_cat: x0 {tuple(x0.shape)} + x1 {tuple(x1.shape)} / weight {tuple(weight.shape)}")
    t0, t1 = x0.permute(0, 2, 3, 1), x1.permute(0, 2, 3, 1)
    y = LinearCatFunction.apply(t0 if t0.is_contiguous() else t0.contiguous(), t1 if t1.is_contiguous() else t1.contiguous(), weight.flatten(1), bias)
    return y.permute(0, 3, 1, 2)


class ShortcutCatFunction(torch.autograd.Function):
    """ResnetBlock.res_conv on ``cat((t0, t1), -1)`` (tokens) AND the hand-over of the two inputs to the block's first convolution: returns (t0, t1, r =
    linear(cat(t0, t1), weight, bias)).  The block reads the returned aliases, so the convolution's data gradient arrives HERE instead of being added to the
    shortcut's by autograd (one elementwise pass over a block input per source: 18 of a step's additions, 6 of them full resolution): the shortcut's own data
    gradient dr @ W takes it as the residual of its GEMM epilogue.  Diffusion_arch.py:163-170 (h + res_conv(x)) on the up path's concatenations (:620-630)."""

    @staticmethod
    def forward(ctx, t0, t1, weight, bias):
        ctx.save_for_backward(t0, t1, weight)
        ctx.has_bias = bias is not None
        ctx.set_materialize_grads(False)
        cout = weight.shape[0]
        c0, c1 = t0.shape[-1], t1.shape[-1]
        r = _pointwise_gemm(_tokens(t0, c0), weight, bias, False, _tokens(t1, c1)).view(*t0.shape[:-1], cout)
        return t0.view_as(t0), t1.view_as(t1), r

    @staticmethod
    @once_differentiable
    def backward(ctx, d0, d1, dr):
        t0, t1, weight = ctx.saved_tensors
        cout, cin = weight.shape
        c0, c1 = t0.shape[-1], t1.shape[-1]
        if dr is None:
            return d0, d1, None, None
        g2 = _tokens(dr, cout)
        g0 = g1 = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # the convolution's gradients are the two channel slices of ONE (tokens, c0 + c1) tensor (Conv3x3CatFunction.backward): the residual of the epilogue
            want, acc = [], 1                                            # strides of a contiguous (..., cin) tensor: what both slices must sit in
            for n in (cin,) + tuple(reversed(t0.shape[:-1])):
                want.insert(0, acc)
                acc *= n
            joined = (d0 is not None and d1 is not None and d0.dtype == torch.float32 and d1.dtype == torch.float32 and d0.shape == t0.shape and d1.shape == t1.shape
                      and list(d0.stride()) == want and list(d1.stride()) == want and d0.data_ptr() + 4 * c0 == d1.data_ptr() and d0.data_ptr() % 16 == 0)
            if joined:
                res = torch.as_strided(d0, (g2.shape[0], cin), (cin, 1))
                full = _pointwise_gemm(g2, weight, None, True, res=res).view(*t0.shape[:-1], cin)
                g0, g1 = full[..., :c0], full[..., c0:]
            else:
                full = _pointwise_gemm(g2, weight, None, True).view(*t0.shape[:-1], cin)
                g0 = full[..., :c0] if d0 is None else full[..., :c0] + d0
                g1 = full[..., c0:] if d1 is None else full[..., c0:] + d1
        grad_w = grad_b = None
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            lib = L.load()
            with _on(g2.device):
                parts = []
                for i, (t, c) in enumerate(((t0, c0), (t1, c1))):
                    x2 = _tokens(t, c)
                    N = x2.shape[0]
                    dw = torch.empty((cout, c), dtype=torch.float32, device=x2.device)
                    if i == 0 and ctx.has_bias:
                        grad_b = torch.empty(cout, dtype=torch.float32, device=x2.device)
                    ws = torch.empty(int(lib.nd_linear_wgrad_workspace_floats(N, c, cout)), dtype=torch.float32, device=x2.device)
                    L.call("nd_linear_wgrad_f32", x2.data_ptr(), c, g2.data_ptr(), cout, dw.data_ptr(),
                           grad_b.data_ptr() if (i == 0 and grad_b is not None) else None, ws.data_ptr(), N, c, cout, _stream(x2.device))
                    parts.append(dw)
                grad_w = torch.cat(parts, dim=1)
        return g0, g1, grad_w, grad_b


def conv1x1_shortcut_cat(x0: torch.Tensor, x1: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """(x0', x1', F.conv2d(torch.cat((x0, x1), 1), weight, bias)) for a ResnetBlock whose shortcut is a 1x1 convolution of a concatenation: x0' / x1' are the inputs
    again, to be read by the block's first convolution, whose data gradient then joins the shortcut's inside one GEMM (ShortcutCatFunction)."""
    if x0.dim() != 4 or tuple(weight.shape[2:]) != (1, 1) or x0.shape[1] + x1.shape[1] != weight.shape[1] or x0.shape[1] % 4 or x1.shape[1] % 4 \
            or weight.shape[0] % 4 or not (x0.is_cuda and x1.is_cuda):
        raise ValueError(f"conv1x1_shortcut_cat: x0 {tuple(x0.shape)} + x1 {tuple(x1.shape)} / weight {tuple(weight.shape)}")
    t0, t1 = x0.permute(0, 2, 3, 1), x1.permute(0, 2, 3, 1)
    a0, a1, r = ShortcutCatFunction.apply(t0 if t0.is_contiguous() else t0.contiguous(), t1 if t1.is_contiguous() else t1.contiguous(), weight.flatten(1), bias)
    return a0.permute(0, 3, 1, 2), a1.permute(0, 3, 1, 2), r.permute(0, 3, 1, 2)


def _linear_ok(cin: int, cout: int) -> bool:
    return cin % 4 == 0 and cout % 4 == 0


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable F.linear(x, weight, bias) (+ res, added in the GEMM's epilogue) with the weight / bias gradient on the HIP library (channel counts multiples of 4)."""
    if weight.dim() != 2 or x.shape[-1] != weight.shape[1] or not _linear_ok(weight.shape[1], weight.shape[0]):
        raise ValueError(f"linear: x {tuple(x.shape)} / weight {tuple(weight.shape)}: needs a 2-D weight with channel counts that are multiples of 4")
    if res is None:
        return LinearFunction.apply(x, weight, bias)
    if tuple(res.shape) != tuple(x.shape[:-1]) + (weight.shape[0],):
        raise ValueError(f"linear: residual {tuple(res.shape)} is not the output's shape {tuple(x.shape[:-1]) + (weight.shape[0],)}")
    return LinearFunction.apply(x, weight, bias, res)


def conv1x1(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable F.conv2d(x, weight, bias) (+ res, NCHW-shaped like the output) for 1x1 kernels as a Linear over the pixels (NHWC tokens; channels_last in and out)."""
    if x.dim() != 4 or tuple(weight.shape[2:]) != (1, 1) or x.shape[1] != weight.shape[1]:
        raise ValueError(f"conv1x1: x {tuple(x.shape)} / weight {tuple(weight.shape)} is not a 1x1 convolution")
    t = x.permute(0, 2, 3, 1)                                              # a view; contiguous when x is channels_last
    r = res.permute(0, 2, 3, 1) if res is not None else None
    y = linear(t if t.is_contiguous() else t.contiguous(), weight.flatten(1), bias, r)
    return y.permute(0, 3, 1, 2)                                           # NCHW-shaped, channels_last in memory


class Conv7x7Function(torch.autograd.Function):
    """The 7x7 stem over a 4-channel image (init_conv / cond_init_conv: Diffusion_arch.py:478, others_arch.py:394-398): forward on nd_conv7x7_c4_f32 (the sampling
    path's kernel), weight and bias gradient on nd_conv7x7_c4_wgrad_f32 (H % 4 == 0, W % 32 == 0, cout in {32, 48, 64, 96, 128}; r5) or, for other
    shapes, as the Linear weight gradient of the unfolded image -- dW[co, (ci, ky, kx)] = sum over the pixels of dy[pixel, co] * patch[pixel, (ci, ky, kx)] on
    nd_linear_wgrad_f32 (both: fixed summation order).  The input is an image, not an activation: its gradient is asked for
    only by callers outside the reference's training loop and then comes from PyTorch's transposed convolution."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = L.load()
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        xn = x.detach().permute(0, 2, 3, 1).contiguous().float()          # NHWC, 4 channels
        st = _stream(x.device)
        with _on(x.device):
            wp = torch.empty(196 * cout, dtype=torch.float32, device=x.device)
            w32 = weight.detach().float().contiguous()
            L.call("nd_pack_conv7x7_weight", w32.data_ptr(), wp.data_ptr(), cout, st)
            b32 = bias.detach().float().contiguous() if bias is not None else torch.zeros(cout, dtype=torch.float32, device=x.device)
            y = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
            L.call("nd_conv7x7_c4_f32", xn.data_ptr(), wp.data_ptr(), b32.data_ptr(), y.data_ptr(), cout, B, H, W, cout, st)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y.permute(0, 3, 1, 2)                                       # NCHW-shaped, channels_last in memory

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        lib = L.load()
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        grad_x = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            grad_x = torch.nn.grad.conv2d_input(x.shape, weight, grad_out, padding=3)
        if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and cin == 4 and \
                int(lib.nd_conv7x7_c4_wgrad_workspace_floats(B, H, W, cout)) >= 0:
            # nd_conv7x7_c4_wgrad_f32: the GEMM over the pixels straight from halo tiles (no unfolded image: 205 MB and two copies at B = 4, 256 x 256)
            g = _tokens(grad_out.permute(0, 2, 3, 1), cout)                 # (pixels, cout)
            xn = x.detach().permute(0, 2, 3, 1).contiguous().float()
            with _on(xn.device):
                grad_w = torch.empty(weight.shape, dtype=torch.float32, device=xn.device)
                grad_b = torch.empty(cout, dtype=torch.float32, device=xn.device) if ctx.has_bias else None
                ws = torch.empty(int(lib.nd_conv7x7_c4_wgrad_workspace_floats(B, H, W, cout)), dtype=torch.float32, device=xn.device)
                L.call("nd_conv7x7_c4_wgrad_f32", xn.data_ptr(), g.data_ptr(), cout, grad_w.data_ptr(), grad_b.data_ptr() if grad_b is not None else None,
                       ws.data_ptr(), B, H, W, cout, _stream(xn.device))
        elif ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            g2 = _tokens(grad_out.permute(0, 2, 3, 1), cout)                # (pixels, cout)
            cols = torch.nn.functional.unfold(x.detach().float(), kernel_size=7, padding=3)      # (B, cin * 49, H W): index ci * 49 + ky * 7 + kx == weight.flatten(1)'s
            x2 = cols.transpose(1, 2).reshape(B * H * W, cin * 49).contiguous()
            N, K = x2.shape
            with _on(x2.device):
                gw = torch.empty((cout, K), dtype=torch.float32, device=x2.device)
                grad_b = torch.empty(cout, dtype=torch.float32, device=x2.device) if ctx.has_bias else None
                ws = torch.empty(int(lib.nd_linear_wgrad_workspace_floats(N, K, cout)), dtype=torch.float32, device=x2.device)
                L.call("nd_linear_wgrad_f32", x2.data_ptr(), K, g2.data_ptr(), cout, gw.data_ptr(),
                       grad_b.data_ptr() if grad_b is not None else None, ws.data_ptr(), N, K, cout, _stream(x2.device))
            grad_w = gw.view(weight.shape)
        return grad_x, grad_w, grad_b


def conv7x7_c4(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable F.conv2d(x, weight, bias, padding=3) for the 7x7 stem of a 4-channel image (cout a multiple of 4) on the HIP library."""
    if x.dim() != 4 or tuple(weight.shape[1:]) != (4, 7, 7) or x.shape[1] != 4 or weight.shape[0] % 4:
        raise ValueError(f"conv7x7_c4: x {tuple(x.shape)} / weight {tuple(weight.shape)}: a 7x7 convolution of a 4-channel image, cout a multiple of 4")
    return Conv7x7Function.apply(x, weight, bias)


def _eligible_linear(m: nn.Module) -> bool:
    if isinstance(m, nn.Linear):
        return _linear_ok(m.in_features, m.out_features)
    return (isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0) and m.groups == 1
            and _linear_ok(m.in_channels, m.out_channels))


def _hip_linear_forward(self, x: torch.Tensor) -> torch.Tensor:
    if isinstance(self, nn.Linear):
        return linear(x, self.weight, self.bias)
    return conv1x1(x, self.weight, self.bias)


def _layer_norm_ok(C_: int) -> bool:
    return C_ % 4 == 0 and 16 <= C_ <= 1024          # (r5: any multiple of 4 -- the d = 48 network's 48 / 96 / 192 / 384 included)


class LayerNormFunction(torch.autograd.Function):
    """nn.LayerNorm(C) over the last dimension of contiguous tokens, forward and backward on libnoisediff_hip (norm_train.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        lib = L.load()
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != torch.float32 or not x2.is_contiguous():
            x2 = x2.float().contiguous()
        if x2.device.type != "cuda":
            raise L.HipError(f"noisediff_amd.train runs on the HIP library only; tensor is on {x2.device} and there is no CPU path")
        N, C_ = x2.shape
        with _on(x2.device):
            y = torch.empty_like(x2)
            stats = torch.empty((N, 2), dtype=torch.float32, device=x2.device)
            w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
            L.call("nd_layernorm_train_forward_f32", x2.data_ptr(), C_, w32.data_ptr(), b32.data_ptr(), y.data_ptr(), C_, stats.data_ptr(),
                   N, C_, float(eps), _stream(x2.device))
        ctx.save_for_backward(x2, weight, stats)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x2, weight, stats = ctx.saved_tensors
        lib = L.load()
        N, C_ = x2.shape
        g2 = grad_out.reshape(N, C_)
        if g2.dtype != torch.float32 or not g2.is_contiguous():
            g2 = g2.float().contiguous()
        with _on(x2.device):
            dx = torch.empty_like(x2)
            dgamma = torch.empty(C_, dtype=torch.float32, device=x2.device)
            dbeta = torch.empty(C_, dtype=torch.float32, device=x2.device)
            ws = torch.empty(int(lib.nd_layernorm_train_workspace_floats(N, C_)), dtype=torch.float32, device=x2.device)
            w32 = weight.detach().float().contiguous()
            L.call("nd_layernorm_train_backward_f32", g2.data_ptr(), C_, x2.data_ptr(), C_, w32.data_ptr(), stats.data_ptr(), dx.data_ptr(), C_,
                   dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), N, C_, _stream(x2.device))
        return dx.view(ctx.shape), dgamma, dbeta, None


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """Differentiable F.layer_norm(x, (C,), weight, bias, eps) over the last dimension on the HIP library (C = 64, 128 or a multiple of 256 <= 1024)."""
    if weight is None or bias is None or not _layer_norm_ok(x.shape[-1]) or weight.shape != (x.shape[-1],):
        raise ValueError(f"layer_norm: x {tuple(x.shape)}: needs affine parameters and C = 64, 128 or a multiple of 256 up to 1024")
    return LayerNormFunction.apply(x, weight, bias, eps)


def _eligible_layer_norm(m: nn.Module) -> bool:
    return (isinstance(m, nn.LayerNorm) and len(m.normalized_shape) == 1 and m.elementwise_affine and m.bias is not None
            and _layer_norm_ok(m.normalized_shape[0]))


def _hip_layer_norm_forward(self: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
    return layer_norm(x, self.weight, self.bias, self.eps)


def _eligible_block(m: nn.Module) -> bool:
    """A module shaped like the reference's ``Block`` (Diffusion_arch.py:128-144): children proj (conv), norm (affine GroupNorm), act (SiLU)
    applied in that order with an optional (scale, shift) modulation in between -- recognised by structure, not by import."""
    proj, norm, act = getattr(m, "proj", None), getattr(m, "norm", None), getattr(m, "act", None)
    return (isinstance(proj, nn.Conv2d) and isinstance(norm, nn.GroupNorm) and isinstance(act, nn.SiLU) and _eligible_norm(norm)
            and len(list(m.children())) == 3 and type(m).__name__ == "Block")


def _hip_block_forward(self, x: torch.Tensor, scale_shift=None) -> torch.Tensor:
    """Block.forward with the norm, the per-sample modulation and the activation as one operator (train.group_norm_silu)."""
    x = self.proj(x)
    n = self.norm
    if scale_shift is None:
        return group_norm_silu(x, n.num_groups, n.weight, n.bias, None, n.eps)
    scale, shift = scale_shift
    if scale.numel() == x.shape[0] * x.shape[1] and shift.numel() == scale.numel():              # (B, C, 1, 1): the time embedding's
        return group_norm_silu(x, n.num_groups, n.weight, n.bias, torch.cat((scale.reshape(x.shape[0], -1), shift.reshape(x.shape[0], -1)), 1), n.eps)
    return self.act(self.norm(x) * (scale + 1) + shift)                                           # per-pixel maps (ResnetBlock2): unfused here; trainable.py fuses them (modulate_silu)


def _eligible(m: nn.Module) -> bool:
    return (isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.dilation == (1, 1)
            and m.groups == 1 and m.padding_mode == "zeros" and m.in_channels % 8 == 0 and m.out_channels % 8 == 0)


def _hip_conv_forward(self: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    return conv3x3(x, self.weight, self.bias)


def accelerate(model: nn.Module, norms: bool = True, linears: bool = True) -> int:
    """Route every eligible 3x3 convolution of ``model`` (stride 1, padding 1, channel counts multiples of 8) and -- unless
    ``norms=False`` / ``linears=False`` -- every affine nn.GroupNorm (C a multiple of 4) and nn.LayerNorm (C = 64, 128, 256 k) and the weight / bias gradient of every
    nn.Linear and 1x1 nn.Conv2d (channel counts multiples of 4) through the HIP library, forward and backward; modules shaped like the
    reference's ``Block`` (proj / norm / act) get their norm + modulation + SiLU tail as one operator.  Parameters,
    module tree and state dict are untouched.  The replacement is made on the module's CLASS (a cached subclass of its own type
    whose ``forward`` is the HIP version), not as an instance attribute: ``copy.deepcopy`` (the trainer's EMA) and
    ``nn.DataParallel`` replicas (``define_G`` wraps the net, models/modules.py:81; a replica copies ``__dict__``, so an instance-bound
    method would keep running on the ORIGINAL module's cuda:0 parameters) resolve ``forward`` through their own ``self``.
    Returns the number of convolutions taken."""
    n = 0
    for m in model.modules():
        if _eligible(m) and getattr(m.forward, "__func__", None) is not _hip_conv_forward:
            _retarget(m, _hip_conv_forward)
            n += 1
        elif norms and _eligible_norm(m) and getattr(m.forward, "__func__", None) is not _hip_norm_forward:
            _retarget(m, _hip_norm_forward)
        elif norms and _eligible_layer_norm(m) and getattr(m.forward, "__func__", None) is not _hip_layer_norm_forward:
            _retarget(m, _hip_layer_norm_forward)
        elif linears and _eligible_linear(m) and getattr(m.forward, "__func__", None) is not _hip_linear_forward:
            _retarget(m, _hip_linear_forward)
        elif norms and _eligible_block(m) and getattr(m.forward, "__func__", None) is not _hip_block_forward:
            _retarget(m, _hip_block_forward)
    return n


_RETARGETED: Dict[tuple, type] = {}


def _rebuild_retargeted(base: type, fn_name: str, state):
    """pickle support (torch.save(model), mp.spawn arguments): the dynamic subclass is not importable by name, so an accelerated
    module is pickled as (its base class, the name of the HIP forward) and re-targeted when it is loaded."""
    m = base.__new__(base)
    m.__dict__.update(state)
    _retarget(m, globals()[fn_name])
    return m


def _reduce_retargeted(self):
    return _rebuild_retargeted, (type(self)._nd_accelerated_base, type(self).forward.__name__, self.__dict__.copy())


def _retarget(m: nn.Module, fn) -> None:
    """m.__class__ <- the cached subclass of type(m) whose forward is ``fn`` (same name / module / qualname, so structural
    checks by class name and ``isinstance`` keep working)."""
    base = getattr(type(m), "_nd_accelerated_base", type(m))
    cls = _RETARGETED.get((base, fn))
    if cls is None:
        cls = type(base.__name__, (base,), {"forward": fn, "__module__": base.__module__, "__qualname__": base.__qualname__,
                                             "__doc__": base.__doc__, "_nd_accelerated_base": base, "__reduce__": _reduce_retargeted})
        _RETARGETED[(base, fn)] = cls
    m.__dict__.pop("forward", None)          # an instance attribute (older accelerate(), user patches) would shadow the class
    m.__class__ = cls


class Adam(torch.optim.Adam):
    """``torch.optim.Adam`` (the reference's optimizer: models/trainer_diffusion.py:94, models/denoising_diffusion_pytorch.py:22) whose ``step`` updates
    all parameters of a group in ONE launch of nd_adam_step_f32 (adam.hip) instead of PyTorch's ~10 foreach passes: 1.14 -> 0.3 ms of the d = 64
    network's 28 ms step.  Same constructor, same state (``step`` -- a CPU tensor per parameter --, ``exp_avg``, ``exp_avg_sq``), so state dicts move
    between the two classes (``--resume_optim``); the same update in fp32 up to the rounding of one fused pass.  What the kernel does not do is refused:
    amsgrad, maximize, differentiable, sparse gradients, parameters that are not fp32 on a GPU.  ``capturable=True`` (PyTorch's flag): the step counters live on
    the device and nothing is computed on the host per step (nd_adam_step_capturable_f32) -- the form a training step captured as ONE CUDA graph needs
    (tools/train_graph_bench.py); run one step before the capture."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **kw):
        for k in ("amsgrad", "maximize", "differentiable", "fused"):
            if kw.get(k):
                raise NotImplementedError(f"noisediff_amd.train.Adam: {k}=True is not built (use torch.optim.Adam)")
        kw.pop("foreach", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, **{k: v for k, v in kw.items() if k != "fused"})
        self._nd_tables: Dict[tuple, tuple] = {}     # (group index, parameter data_ptrs) -> (chunk table on the device, number of chunks)

    def check_captured_lr(self) -> None:
        """Call before replaying a captured training step: a captured ``capturable=True`` step carries the learning rate it was captured with (a launch
        argument).  Raises if a scheduler or the caller has changed ``group['lr']`` since -- re-capture the step then."""
        cap = getattr(self, "_nd_captured_lr", {})
        for group in self.param_groups:
            lr0 = cap.get(id(group))
            if lr0 is not None and float(group["lr"]) != lr0:
                raise RuntimeError(f"noisediff_amd.train.Adam: the captured step was recorded with lr={lr0:g}, the group now has lr={float(group['lr']):g}: "
                                   "a replay would ignore the change; capture the step again")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if capturing and not all(g.get("capturable") for g in self.param_groups):
            raise RuntimeError("noisediff_amd.train.Adam computes its step sizes on the host unless built with capturable=True: this step cannot be captured "
                               "into a CUDA graph")
        lib = L.load()
        import numpy as np
        per = int(lib.nd_adam_chunk_elements())
        item_t = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("step_size", "<f4"), ("bias2_sqrt", "<f4"), ("vec4", "<i4"),
                           ("reserved", "<i4"), ("step", "<u8")])                    # nd_adam_item (L.AdamItem)
        # ---- what does not change from step to step is checked and laid out once per set of parameters (the host side of a step is ~1 ms this way, 4.4
        #      per parameter); EVERY group's table is validated before ANY group is launched (a rebuild in the middle used to re-run the whole
        #      step and move the groups already updated twice)
        work = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            cap = bool(group.get("capturable"))                          # step counters on the device, nothing computed on the host (nd_adam_step_capturable_f32)
            key = (gi, cap, tuple(p.data_ptr() for p in ps))
            ent = self._nd_tables.get(key)
            if ent is not None:
                # load_state_dict (or anything else) replaced the state tensors: lay the table out again (compared by object, not by id() of a dict that
                # may have been freed and its id reused)
                _, _, _, ms0, vs0, steps0 = ent
                for p, m0, v0, t0 in zip(ps, ms0, vs0, steps0):
                    st = self.state[p]
                    if st.get("exp_avg") is not m0 or st.get("exp_avg_sq") is not v0 or st.get("step") is not t0:
                        ent = None
                        break
            if ent is None:
                if capturing:
                    raise RuntimeError("noisediff_amd.train.Adam: run a step with the same parameters before the capture (state and tables are built on the first step)")
                if len(self._nd_tables) > 64:
                    self._nd_tables.clear()
                for p in ps:
                    if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda or p.device != dev or not p.is_contiguous():
                        raise NotImplementedError("noisediff_amd.train.Adam updates dense, contiguous fp32 parameters of one GPU per group")
                    st = self.state[p]
                    if len(st) == 0:                                     # torch.optim.Adam._init_group's state
                        st["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if cap else torch.tensor(0.0, dtype=torch.float32)
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    m, v = st["exp_avg"], st["exp_avg_sq"]
                    if not (m.is_contiguous() and v.is_contiguous()) or m.device != dev or v.device != dev:
                        raise NotImplementedError("noisediff_amd.train.Adam: optimizer state must be contiguous and on the parameter's device")
                    if cap and not (st["step"].is_cuda and st["step"].dtype == torch.float32):
                        raise NotImplementedError("noisediff_amd.train.Adam(capturable=True): the step counters must be fp32 tensors on the parameter's device")
                    if not cap and st["step"].is_cuda:
                        raise NotImplementedError("noisediff_amd.train.Adam: step counters on the device need capturable=True")
                items = np.zeros(len(ps), dtype=item_t)
                ms, vs, steps = [self.state[p]["exp_avg"] for p in ps], [self.state[p]["exp_avg_sq"] for p in ps], [self.state[p]["step"] for p in ps]
                items["p"] = [p.data_ptr() for p in ps]
                items["m"] = [t.data_ptr() for t in ms]
                items["v"] = [t.data_ptr() for t in vs]
                items["n"] = [p.numel() for p in ps]
                if cap:
                    items["step"] = [t.data_ptr() for t in steps]
                pairs = [(i, c) for i, p in enumerate(ps) for c in range((p.numel() + per - 1) // per)]
                chunks = torch.tensor(pairs, dtype=torch.int32).reshape(-1, 2).to(dev)
                ent = self._nd_tables[key] = (items, chunks, len(pairs), ms, vs, steps)
            work.append((group, ps, dev, cap, ent))
        for group, ps, dev, cap, ent in work:
            beta1, beta2 = group["betas"]
            lr = float(group["lr"])
            if cap:
                # the learning rate is a launch argument: a captured step replays with the value it was captured with (the reference's CosineAnnealingLR
                # and its hand-written group['lr'] = ..., trainer_diffusion.py:95,104-105, do not reach a replayed graph) -- refuse a silent mismatch
                if capturing:
                    self._nd_captured_lr = getattr(self, "_nd_captured_lr", {})
                    self._nd_captured_lr[id(group)] = lr
            items, chunks, n_chunks, ms, vs, steps = ent
            # ---- this step: gradient pointers, step sizes
            gs = [p.grad for p in ps]
            if not all(g.is_contiguous() and g.dtype == torch.float32 and not g.is_sparse for g in gs):
                gs = [g.contiguous() if not g.is_sparse and g.dtype == torch.float32 else None for g in gs]
                if any(g is None for g in gs):
                    raise NotImplementedError("noisediff_amd.train.Adam updates from dense fp32 gradients")
            items = items.copy()
            items["g"] = [g.data_ptr() for g in gs]
            items["vec4"] = ((items["p"] | items["g"] | items["m"] | items["v"]) & 15) == 0
            if not cap:
                torch._foreach_add_(steps, 1.0)                          # (CPU scalars: one C++ loop)
                # every parameter's OWN step count (a cheap loop over CPU scalars): parameters that were stepped under another set -- grad None on some
                # steps -- carry different counts, and each gets its own bias correction
                ts = np.array([float(t) for t in steps], dtype=np.float64)
                items["step_size"] = lr / (1.0 - beta1 ** ts)
                items["bias2_sqrt"] = np.sqrt(1.0 - beta2 ** ts)
            # pointers and step sizes of this step (a few KB): pinned + asynchronous, so the host keeps running ahead of the device
            host_table = torch.from_numpy(items.view(np.uint8)).pin_memory()
            table = host_table.to(dev, non_blocking=True)
            with _on(dev):
                if cap:
                    L.call("nd_adam_step_capturable_f32", table.data_ptr(), len(ps), chunks.data_ptr(), n_chunks, lr, float(beta1), float(beta2),
                           float(group["eps"]), float(group["weight_decay"]), _stream(dev))
                else:
                    L.call("nd_adam_step_f32", table.data_ptr(), len(ps), chunks.data_ptr(), n_chunks, float(beta1), float(beta2), float(group["eps"]),
                           float(group["weight_decay"]), _stream(dev))
            if capturing:                                                # a captured step replays with these tables: they live as long as the optimizer
                self._nd_captured = getattr(self, "_nd_captured", []) + [(table, host_table, gs)]
            self._nd_keep = (table, gs)                                  # alive until the next step (the launch is asynchronous)
            # the kernel wrote through raw pointers: tell autograd (and the packing caches, which compare version counters) that these tensors changed
            torch.autograd.graph.increment_version(ps)
            torch.autograd.graph.increment_version(ms)
            torch.autograd.graph.increment_version(vs)
        return loss
