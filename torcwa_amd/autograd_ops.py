"""torch.autograd.Function wrappers around the libtrx primitives (forward AND backward run on the HIP kernels).

Used by the differentiable path of `BatchedRCWA` (Examples 4-6 of the reference: gradients w.r.t. permittivity grids,
thickness, ...).  PyTorch's complex-autograd convention applies: for C = f(A) holomorphic, grad_A = conj(f') applied to
grad_C, i.e. for C = A B: grad_A = grad_C B^H, grad_B = A^H grad_C.
"""
import math

import torch


class ConvMatFn(torch.autograd.Function):
    """E = convmat(grid) (torcwa/rcwa.py:1183-1204).  Linear in the grid; the backward is its adjoint: Toeplitz-diagonal
    sums of grad_E followed by the conjugate (inverse-direction) pruned DFT."""

    @staticmethod
    def forward(ctx, grid, ox, oy, cdtype, engine):
        ctx.meta = (ox, oy, grid.shape, grid.is_complex(), grid.dtype)
        return engine.convmat(grid, ox, oy, cdtype)

    @staticmethod
    def backward(ctx, gE):
        ox, oy, (B, nx, ny), cplx, gdt = ctx.meta
        dev = gE.device
        wy = 2 * oy + 1
        N = (2 * ox + 1) * wy
        idx = torch.arange(N, device=dev)
        m, n_ = idx // wy, idx % wy
        dm = (m[:, None] - m[None, :] + 2 * ox).reshape(-1)
        dn = (n_[:, None] - n_[None, :] + 2 * oy).reshape(-1)
        nq = 4 * oy + 1
        G = torch.zeros((B, (4 * ox + 1) * nq), dtype=gE.dtype, device=dev)
        G.index_add_(1, dm * nq + dn, gE.reshape(B, -1))                    # G[p,q] = sum_{(i,j): diff = (p,q)} gE[i,j]
        G = G.reshape(B, 4 * ox + 1, nq)
        rdt = torch.float64 if gE.dtype == torch.complex128 else torch.float32
        x = torch.arange(nx, device=dev, dtype=rdt)[:, None]
        p = torch.arange(-2 * ox, 2 * ox + 1, device=dev, dtype=rdt)[None, :]
        y = torch.arange(ny, device=dev, dtype=rdt)[None, :]
        q = torch.arange(-2 * oy, 2 * oy + 1, device=dev, dtype=rdt)[:, None]
        Fx = torch.exp(2j * math.pi * (x * p) / nx).to(gE.dtype)             # conj of the forward twiddle
        Fy = torch.exp(2j * math.pi * (q * y) / ny).to(gE.dtype)
        g = (Fx[None] @ G @ Fy[None]) / (nx * ny)                            # [B, nx, ny]  (tiny: 4ox+1 inner dims)
        if not cplx:
            g = torch.real(g)
        return g.to(gdt), None, None, None, None


class GemmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B, engine):
        ctx.engine = engine
        ctx.save_for_backward(A, B)
        return engine.gemm(A, B)

    @staticmethod
    def backward(ctx, G):
        A, B = ctx.saved_tensors
        eng = ctx.engine
        G = G.contiguous()
        gA = eng.gemm(G, B, opB=2) if ctx.needs_input_grad[0] else None      # G B^H
        gB = eng.gemm(A, G, opA=2) if ctx.needs_input_grad[1] else None      # A^H G
        return gA, gB, None


class InverseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, engine):
        Y = engine.inverse(A)
        ctx.engine = engine
        ctx.save_for_backward(Y)
        return Y

    @staticmethod
    def backward(ctx, G):
        (Y,) = ctx.saved_tensors
        eng = ctx.engine
        T = eng.gemm(Y, G.contiguous(), opA=2)                               # Y^H G
        return -eng.gemm(T, Y, opB=2), None                                   # - Y^H G Y^H


class SolveFn(torch.autograd.Function):
    """X = A^-1 B."""

    @staticmethod
    def forward(ctx, A, B, engine):
        X = engine.solve(A, B)
        ctx.engine = engine
        ctx.save_for_backward(A, X)
        return X

    @staticmethod
    def backward(ctx, G):
        A, X = ctx.saved_tensors
        eng = ctx.engine
        AH = torch.conj(A).transpose(-2, -1).contiguous()
        gB = eng.solve(AH, G.contiguous())                                    # A^-H G
        gA = -eng.gemm(gB, X, opB=2) if ctx.needs_input_grad[0] else None     # - gB X^H
        return gA, gB, None
