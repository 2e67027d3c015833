"""`Eig`: the reference's single-op plugin seam (torcwa/torch_eig.py:8-44), served by libtrx.

forward  : trx_eig (batched HIP eigensolver) instead of torch.linalg.eig            (torch_eig.py:12-17)
backward : torcwa's own broadened adjoint, F = conj(s)/(|s|^2 + eps), with the dense products and the
           (V^H)^-1 solve done by libtrx GEMM / LU kernels                           (torch_eig.py:20-44)
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

    @staticmethod
    def forward(ctx, x):
        eng = Eig._eng()
        xb = x if x.dim() == 3 else x[None]
        was_real = not torch.is_complex(xb)
        if was_real:
            xb = xb.to(torch.complex128 if xb.dtype == torch.float64 else torch.complex64)
        w, V = eng.eig(xb.contiguous())
        ctx.batched = x.dim() == 3
        ctx.was_real = was_real
        ctx.save_for_backward(w, V)
        return (w, V) if ctx.batched else (w[0], V[0])

    @staticmethod
    def backward(ctx, grad_eigval, grad_eigvec):
        eng = Eig._eng()
        w, V = ctx.saved_tensors
        gw = grad_eigval if ctx.batched else grad_eigval[None]
        gV = grad_eigvec if ctx.batched else grad_eigvec[None]
        gw, gV = gw.to(w.dtype), gV.to(V.dtype)
        s = w.unsqueeze(-2) - w.unsqueeze(-1)                       # s_ij = w_j - w_i
        if Eig.broadening_parameter is not None:
            F = torch.conj(s) / (torch.abs(s) ** 2 + Eig.broadening_parameter)
        elif s.dtype == torch.complex64:
            F = torch.conj(s) / (torch.abs(s) ** 2 + 1.4e-45)
        else:
            F = torch.conj(s) / (torch.abs(s) ** 2 + 4.9e-324)
        idx = torch.arange(F.shape[-1], device=F.device)
        F[:, idx, idx] = 0.
        XH = torch.conj(V).transpose(-2, -1).contiguous()
        inner = torch.diag_embed(gw) + torch.conj(F) * eng.gemm(XH, gV.contiguous())
        rhs = eng.gemm(inner, XH)
        grad = eng.solve(XH, rhs)                                   # (V^H)^-1 (...) V^H
        if ctx.was_real:
            grad = torch.real(grad)
        return grad if ctx.batched else grad[0]
