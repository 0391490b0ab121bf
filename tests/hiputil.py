"""Thin helpers for calling the stand-alone C-ABI ops with torch CUDA tensors (tests only)."""
import ctypes as C

import torch

from mcvd_pytorch_amd import _lib


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Ctx:
    def __init__(self):
        self.h = C.c_void_p()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib.mcvd_ctx_create(torch.cuda.current_device(), C.c_void_p(stream), C.byref(self.h)), "ctx_create")

    def opt(self, key, val):
        _lib.check(_lib.lib.mcvd_ctx_set_option(self.h, key.encode(), int(val)), "set_option")

    def conv2d(self, x0, w, bias, x1=None, coef=None, act=0, res=None, scale=1.0):
        B, C0, H, W = x0.shape
        C1 = x1.shape[1] if x1 is not None else 0
        Cout, ks = w.shape[0], w.shape[-1]
        y = torch.empty(B, Cout, H, W, device=x0.device)
        _lib.check(_lib.lib.mcvd_op_conv2d(self.h, P(x0), C0, P(x1), C1, P(w.contiguous()), P(bias), Cout, ks, P(coef), act,
                                           P(res), scale, P(y), B, H, W), "op_conv2d")
        return y

    def conv2d_stats(self, x0, w, bias, **kw):
        """conv2d that also returns the GroupNorm partials its epilogue wrote: (y, stats [B, Cout, np, 2], np); np == 0 means the
        kernel that ran does not emit."""
        B, Cout, H, W = x0.shape[0], w.shape[0], x0.shape[2], x0.shape[3]
        buf = torch.full((B * Cout * (H * W // 32) * 2,), float("nan"), device=x0.device)
        _lib.check(_lib.lib.mcvd_ctx_set_stats_buffer(self.h, P(buf)))
        try:
            y = self.conv2d(x0, w, bias, **kw)
            np_ = _lib.lib.mcvd_last_conv_stats_np()
        finally:
            _lib.check(_lib.lib.mcvd_ctx_set_stats_buffer(self.h, None))
        st = buf[:B * Cout * np_ * 2].view(B, Cout, np_, 2) if np_ > 0 else None
        return y, st, np_

    def gn_finalize(self, st0, np0, groups, eps, mode, HW, st1=None, np1=1, p0=None, p1=None, emb_stride=0, emb_off=0):
        B, C0 = st0.shape[:2]
        C1 = st1.shape[1] if st1 is not None else 0
        coef = torch.empty(B, C0 + C1, 2, device=st0.device)
        _lib.check(_lib.lib.mcvd_op_gn_finalize(self.h, P(st0.contiguous()), C0, np0, P(st1.contiguous()) if st1 is not None else None, C1, np1,
                                                groups, eps, mode, P(p0), P(p1), emb_stride, emb_off, P(coef), B, HW), "op_gn_finalize")
        return coef

    def gn_coef(self, x0, groups, eps, mode, x1=None, p0=None, p1=None, emb_stride=0, emb_off=0):
        B, C0 = x0.shape[:2]
        C1 = x1.shape[1] if x1 is not None else 0
        HW = x0.shape[2] * x0.shape[3]
        coef = torch.empty(B, C0 + C1, 2, device=x0.device)
        _lib.check(_lib.lib.mcvd_op_gn_coef(self.h, P(x0), C0, P(x1), C1, groups, eps, mode, P(p0), P(p1), emb_stride, emb_off,
                                            P(coef), B, HW), "op_gn_coef")
        return coef

    def attention(self, qkv, heads):
        B, C3, HW = qkv.shape
        out = torch.empty(B, C3 // 3, HW, device=qkv.device)
        _lib.check(_lib.lib.mcvd_op_attention(self.h, P(qkv), P(out), B, C3 // 3, heads, HW), "op_attention")
        return out

    def fir2(self, x, up, coef=None, act=0):
        B, Cc, H, W = x.shape
        y = torch.empty(B, Cc, H * 2 if up else H // 2, W * 2 if up else W // 2, device=x.device)
        _lib.check(_lib.lib.mcvd_op_fir2(self.h, P(x), P(coef), act, 1 if up else 0, P(y), B, Cc, H, W), "op_fir2")
        return y

    def upfirdn2d(self, x, kernel_cpu, up, down, pad0, pad1):
        N, Cc, H, W = x.shape
        kh, kw = kernel_cpu.shape
        oh = (H * up + pad0 + pad1 - kh) // down + 1
        ow = (W * up + pad0 + pad1 - kw) // down + 1
        y = torch.empty(N, Cc, oh, ow, device=x.device)
        k = kernel_cpu.float().contiguous()
        _lib.check(_lib.lib.mcvd_upfirdn2d(self.h, P(x), P(k), kh, kw, up, down, pad0, pad1, P(y), N, Cc, H, W), "upfirdn2d")
        return y

    def randn(self, B, per, seed, offset, draw):
        out = torch.empty(B, per, device="cuda")
        _lib.check(_lib.lib.mcvd_randn(self.h, P(out), seed, offset, draw, B, per), "randn")
        return out

    def __del__(self):
        try:
            _lib.lib.mcvd_ctx_destroy(self.h)
        except Exception:
            pass


def module_output(net, module, B):
    """Output of reference module index `module` from the last forward at batch B (debug API)."""
    cap = 1 << 28
    buf = torch.empty(cap, device=net.device)
    c, h = C.c_int(), C.c_int()
    _lib.check(_lib.lib.mcvd_model_module_output(net._model, module, B, P(buf), cap, C.byref(c), C.byref(h)), "module_output")
    if h.value == 0:
        return buf[:B * c.value].view(B, c.value).clone()
    return buf[:B * c.value * h.value * h.value].view(B, c.value, h.value, h.value).clone()
