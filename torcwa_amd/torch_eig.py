"""`Eig`: the reference's single-op plugin seam (torcwa/torch_eig.py:8-44), served by libtrx.

forward  : trx_eig (batched HIP eigensolver) instead of torch.linalg.eig            (torch_eig.py:12-17)
backward : torcwa's own broadened adjoint, F = conj(s)/(|s|^2 + eps): trx_eig_backward (two GEMMs, one fused
           elementwise kernel, one LU solve with V^H)                                (torch_eig.py:20-44)
Accepts [n,n] (like the reference) or batched [B,n,n] input.
"""
import torch

from .engine import default_engine


class Eig(torch.autograd.Function):
    broadening_parameter = 1e-10          # process-global, settable (mutated in example/Example4.ipynb)
    engine = None

    @staticmethod
    def _eng():
        return Eig.engine if Eig.engine is not None else default_engine()

    UNBROADENED = "unbroadened"

    @staticmethod
    def forward(ctx, x, mode=None):
        """mode None (the reference's signature, `Eig.apply(x)`): the backward reads `Eig.broadening_parameter` when it runs,
        like torch_eig.py:27.  mode Eig.UNBROADENED: this call's backward never broadens, whatever the global says -- the
        `stable_eig_grad=False` branch of the reference (rcwa.py:1238, plain torch.linalg.eig); the choice is bound to the
        graph node at forward time, so no global is touched and concurrent solvers cannot disturb each other."""
        eng = Eig._eng()
        ctx.eng = eng                         # the backward runs on the engine (device, stream) of its forward
        ctx.mode = mode
        ctx.nargs = 1 if mode is None else 2
        xb = x if x.dim() == 3 else x[None]
        was_real = not torch.is_complex(xb)
        if was_real:
            xb = xb.to(torch.complex128 if xb.dtype == torch.float64 else torch.complex64)
        w, V = eng.eig(xb.contiguous(), refine_steps=3)      # differentiable path: always the three-step (all-fp64 class) refinement
        ctx.batched = x.dim() == 3
        ctx.was_real = was_real
        ctx.save_for_backward(w, V)
        return (w, V) if ctx.batched else (w[0], V[0])

    @staticmethod
    def backward(ctx, grad_eigval, grad_eigvec):
        eng = ctx.eng
        w, V = ctx.saved_tensors
        gw = grad_eigval if ctx.batched else grad_eigval[None]
        gV = grad_eigvec if ctx.batched else grad_eigvec[None]
        gw, gV = gw.to(w.dtype), gV.to(V.dtype)
        # F = conj(s)/(|s|^2 + eps), s_ij = w_j - w_i; eps = broadening, or the smallest positive number of the dtype when the
        # broadening is switched off (torch_eig.py:27-31); the whole adjoint is one libtrx call (include/trx.h: trx_eig_backward)
        if ctx.mode != Eig.UNBROADENED and Eig.broadening_parameter is not None:
            eps = float(Eig.broadening_parameter)
        else:
            eps = 1.4e-45 if w.dtype == torch.complex64 else 4.9e-324
        grad = eng.eig_backward(w, V, gw.contiguous(), gV.contiguous(), eps)
        if ctx.was_real:
            grad = torch.real(grad)
        grad = grad if ctx.batched else grad[0]
        return grad if ctx.nargs == 1 else (grad, None)
