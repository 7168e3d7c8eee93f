"""TEST INFRASTRUCTURE: bf16-STORAGE emulation of a torch module graph on the CPU.

Question this answers: is what the bf16 HIP path shows against the fp32 reference -- loss spikes in a few
K-step runs, first-block gradient norms a few per cent low -- a property of storing activations / gradients in bf16 between fp32
accumulating kernels, or a defect of the HIP path?  `attach(model)` makes the IMPORTED REFERENCE (or any nn.Module graph) behave like
the arithmetic contract of the HIP performance mode (DESIGN.md §2 "Numerics"):

  * every tensor that the HIP plan keeps in HBM as bf16 is rounded (RNE) to bf16 where it is produced: the outputs of the 3x3
    convolutions (pre-BatchNorm `i` / `z`), of ConvTranspose2d (`up`), of every ReLU (the BatchNorm-apply + ReLU operand and the block
    output) -- and the GRADIENT arriving at the same place is rounded too (the backward pass stores d(out), dz, di, d(up) in bf16);
  * convolution weights are rounded to bf16 as MFMA operands (straight-through: the fp32 master weight receives the gradient);
  * everything else (BatchNorm statistics, the ECAM head, the loss, parameter gradients, Adam) stays fp32, accumulation is fp32.

It does NOT reproduce the summation order of the HIP kernels: it is one more rounding realisation of the same contract, which is what
the comparison needs.  Nothing under kurosiwo_amd/ imports this file.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _RoundBoth(torch.autograd.Function):
    """y = bf16(x); dx = bf16(dy)"""

    @staticmethod
    def forward(ctx, x):
        return bf16_round(x)

    @staticmethod
    def backward(ctx, g):
        return bf16_round(g)


class _RoundSTE(torch.autograd.Function):
    """w_op = bf16(w); dw passes through in fp32 (master weights)"""

    @staticmethod
    def forward(ctx, w):
        return bf16_round(w)

    @staticmethod
    def backward(ctx, g):
        return g


def round_both(t):
    return _RoundBoth.apply(t)


def attach(model, round_inputs=True, round_grads=True, skip=("ca.", "ca1.", "conv_final"), what=("w", "conv", "relu", "up"), only=None,
           fp32_operands=("conv0_0.conv1",)):
    """Instance-level forward overrides; returns the list of patched module names.  `skip`: name prefixes that stay fp32 (the ECAM
    head of SNUNet runs in fp32 registers in the HIP path).  `what` / `only(name)`: bisection switches (which roundings, which modules).
    `fp32_operands`: convolutions whose image and weights are NOT rounded (the HIP first-layer kernel multiplies the fp32 image by
    the fp32 weights, csrc/elementwise.hip conv_first_fwd_kernel; its OUTPUT is stored in bf16 like every other)."""
    rnd = round_both if round_grads else (lambda t: t + (bf16_round(t) - t).detach())
    ident = lambda t: t
    rw = _RoundSTE.apply if "w" in what else ident
    rc = rnd if "conv" in what else ident
    rr = rnd if "relu" in what else ident
    ru = rnd if "up" in what else ident
    patched = []
    for name, m in model.named_modules():
        if any(name.startswith(s) or ("." + s) in name for s in skip):
            continue
        if only is not None and not only(name):
            continue
        if isinstance(m, nn.Conv2d):
            def fwd(x, m=m, keep=name in fp32_operands):
                if round_inputs and not keep:
                    x = rnd(x)          # MFMA A operand (a no-op for tensors that are already stored in bf16)
                return rc(F.conv2d(x, m.weight if keep else rw(m.weight), m.bias, m.stride, m.padding, m.dilation, m.groups))
            m.forward = fwd
            patched.append(name)
        elif isinstance(m, nn.ConvTranspose2d):
            def fwd(x, m=m):
                return ru(F.conv_transpose2d(rnd(x) if round_inputs else x, rw(m.weight), m.bias, m.stride, m.padding,
                                             m.output_padding, m.groups, m.dilation))
            m.forward = fwd
            patched.append(name)
        elif isinstance(m, nn.ReLU):
            def fwd(x, m=m):
                return rr(F.relu(x))    # not in place: the pre-activation is a different stored tensor
            m.forward = fwd
            patched.append(name)
    return patched
