"""``upfirdn2d`` backed by the MI355X kernel ``sis_upfirdn2d``.

Public surface = the reference module networks/stylegan2/op/upfirdn2d.py: the function
``upfirdn2d(input[N,C,H,W], kernel[kh,kw], up=1, down=1, pad=(p0, p1))`` (:144-149) and the autograd
classes ``UpFirDn2d`` (:87-141) / ``UpFirDn2dBackward`` (:18-84).

The op is linear: y = D_down . F_k . P_pad . U_up x.  Its adjoint is again an upfirdn2d with the
roles of up and down exchanged, the taps flipped, and pads chosen so that the result has the input's
size (the closed form the reference uses, upfirdn2d.py:110-113):

    g_pad0 = k - pad0 - 1,        g_pad1 = in*up - out*down + pad0 - up + 1

That geometry lives in ``_Geometry``; both Functions are thin shells around it, and because the
adjoint of the adjoint is the forward op, ``UpFirDn2dBackward.backward`` just runs the forward
geometry again (needed only for second-order terms of GAN training, out of the synthesis path).
"""
from dataclasses import dataclass
from typing import Tuple

import torch
from torch.autograd import Function

import sis_hip


@dataclass(frozen=True)
class _Geometry:
    up: Tuple[int, int]        # (x, y)
    down: Tuple[int, int]      # (x, y)
    pad: Tuple[int, int, int, int]  # x0, x1, y0, y1
    taps: Tuple[int, int]      # (kh, kw)
    in_hw: Tuple[int, int]

    @property
    def out_hw(self):
        (ux, uy), (dx, dy), (px0, px1, py0, py1) = self.up, self.down, self.pad
        kh, kw = self.taps
        h, w = self.in_hw
        return (h * uy + py0 + py1 - kh) // dy + 1, (w * ux + px0 + px1 - kw) // dx + 1

    def adjoint(self):
        (ux, uy), (dx, dy), (px0, _, py0, _) = self.up, self.down, self.pad
        kh, kw = self.taps
        h, w = self.in_hw
        oh, ow = self.out_hw
        gpad = (kw - px0 - 1, w * ux - ow * dx + px0 - ux + 1, kh - py0 - 1, h * uy - oh * dy + py0 - uy + 1)
        return _Geometry(up=self.down, down=self.up, pad=gpad, taps=self.taps, in_hw=(oh, ow))

    def run(self, planes, taps):
        """planes: [P, h, w] -> [P, oh, ow] through the native [major, h, w, minor] entry point."""
        out = sis_hip.upfirdn2d(planes.reshape(-1, self.in_hw[0], self.in_hw[1], 1), taps,
                                self.up[0], self.up[1], self.down[0], self.down[1], *self.pad)
        return out.squeeze(-1)


class UpFirDn2dBackward(Function):
    """grad_output -> grad_input for a given forward geometry (argument list as the reference)."""

    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        fwd = _Geometry(tuple(up), tuple(down), tuple(pad), tuple(kernel.shape), (in_size[2], in_size[3]))
        adj = _Geometry(tuple(down), tuple(up), tuple(g_pad), tuple(kernel.shape), tuple(out_size))
        ctx.fwd, ctx.in_size = fwd, tuple(in_size)
        ctx.save_for_backward(kernel)
        grad_input = adj.run(grad_output.reshape(-1, out_size[0], out_size[1]), grad_kernel)
        return grad_input.reshape(in_size)

    @staticmethod
    def backward(ctx, gradgrad_input):
        (kernel,) = ctx.saved_tensors
        n, c = ctx.in_size[:2]
        oh, ow = ctx.fwd.out_hw
        gg = ctx.fwd.run(gradgrad_input.reshape(-1, *ctx.fwd.in_hw), kernel)
        return (gg.reshape(n, c, oh, ow),) + (None,) * 8


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        n, c, h, w = input.shape
        geo = _Geometry(tuple(up), tuple(down), tuple(pad), tuple(kernel.shape), (h, w))
        oh, ow = geo.out_hw
        ctx.geo, ctx.in_size = geo, (n, c, h, w)
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]))
        return geo.run(input.reshape(n * c, h, w), kernel).reshape(n, c, oh, ow)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, flipped = ctx.saved_tensors
        geo = ctx.geo
        grad_input = UpFirDn2dBackward.apply(grad_output, kernel, flipped, geo.up, geo.down, geo.pad,
                                             geo.adjoint().pad, ctx.in_size, geo.out_hw)
        return grad_input, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """Same factor and the same (pad0, pad1) on both axes, as the reference's public wrapper."""
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
